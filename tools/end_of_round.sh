#!/bin/bash
# End-of-round evidence on ONE MI355X box (run as `gpurun -- bash tools/end_of_round.sh [tag]`): replay tables, the default
# bench line, the rocprofv3 kernel-trace summary of the same command and the two PMC passes (own runs, --kernel-trace
# only).  Writes gpurun_out/<tag>/<tag>_*; the files judged are copied from there into profiles/ (see profiles/README.md).
TAG=${1:-r2z}
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --out $O/${TAG}_replay_few_shot_4kx32.json > $O/replay_fs.log 2>&1
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --pipelined --out $O/${TAG}_replay_few_shot_4kx32_pipelined.json > $O/replay_fsp.log 2>&1
timeout 300 python tools/replay.py --task reasoning --out $O/${TAG}_replay_reasoning_tot50.json > $O/replay_tot.log 2>&1
timeout 300 python tools/replay.py --task reasoning --model llama3-8b --out $O/${TAG}_replay_reasoning_tot50_llama3.json > $O/replay_tot3.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten seq --tree-size 64 --out $O/${TAG}_replay_speculative_64.json > $O/replay_sd.log 2>&1
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python $R/bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-e2e --no-cfg5 > $O/${TAG}_bench_under_rocprof_stats.json 2> $O/rocprof.err
python $R/tools/prof_summary.py /tmp/prof_$TAG > $O/${TAG}_kernel_stats.txt 2>&1
tail -c 600 $O/${TAG}_bench_default.json | head -c 300; echo; head -8 $O/${TAG}_kernel_stats.txt | cut -c1-160
# HBM traffic of the same command: PMC counters, one pass each, with --kernel-trace only (MI355X_MICROARCH.md "HBM")
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --no-e2e --no-cfg5 > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c > $O/${TAG}_pmc_$(echo $c | tr A-Z a-z).json 2>> $O/pmc_$c.err
done
grep -A3 stage1_np $O/${TAG}_pmc_fetch_size.json | head -5; grep -A3 stage1_np $O/${TAG}_pmc_write_size.json | head -5
# the whole decode step with the tree advancing, under the kernel trace: what a step costs kernel by kernel
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rp_$TAG -- python $R/tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --no-warmup > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/prof_rp_$TAG 2>&1 | head -18 | cut -c1-160 > $O/${TAG}_replay_kernel_stats.txt
