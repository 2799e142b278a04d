cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-cfg5 --no-traffic --steps 5 --warmup 1 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_e2e 2>&1 | head -18 | cut -c1-150
