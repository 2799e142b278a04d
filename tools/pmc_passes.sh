R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r1e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c > $O/r1e_pmc_$(echo $c | tr A-Z a-z)_np.json 2>> $O/pmc_$c.err
done
grep -A3 stage1_np $O/r1e_pmc_fetch_size_np.json | head -5; grep -A3 stage1_np $O/r1e_pmc_write_size_np.json | head -5
