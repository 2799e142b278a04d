#!/bin/bash
# Evidence for the latency-bound configurations: per-workgroup timelines, rocprofv3 kernel
# stats and a FETCH_SIZE pass for each workload.  Run as `gpurun -- bash tools/r2_profile_small.sh <tag> [workloads]`;
# writes gpurun_out/<tag>/; the files judged are copied from there into profiles/.
TAG=${1:-r2a}; shift
WLS=${@:-tot50_4k medusa64_node gqa_4kx32 forest_8kx8 northstar_4kx32}
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  WL=$wl timeout 300 python $R/tools/np_timeline.py > $O/${TAG}_timeline_$wl.txt 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -- python $R/bench.py --workload $wl --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $O/${TAG}_bench_$wl.json 2> $O/rocprof_$wl.err
  python $R/tools/prof_summary.py /tmp/prof_$wl 2>&1 | head -8 > $O/${TAG}_kernel_stats_$wl.txt
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$wl -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2> $O/pmc_$wl.err
  python $R/tools/pmc_summary.py /tmp/pmc_$wl > $O/${TAG}_pmc_fetch_size_$wl.json 2>> $O/pmc_$wl.err
  echo "== $wl"; head -4 $O/${TAG}_kernel_stats_$wl.txt | cut -c1-150; grep -m1 -A4 stage1_np $O/${TAG}_pmc_fetch_size_$wl.json | grep hbm_read
done
