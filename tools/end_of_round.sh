#!/bin/bash
# End-of-round evidence on ONE MI355X box (run as `gpurun -- bash tools/end_of_round.sh`): replay tables, the default
# bench line, the rocprofv3 kernel-trace summary of the same command and the two PMC passes (own runs, --kernel-trace
# only).  Writes gpurun_out/r1e/r1e_*; the files judged are copied from there into profiles/ (see profiles/README.md).
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r1e; mkdir -p $O
cd $R
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --out $O/r1e_replay_few_shot_4kx32.json > $O/replay_fs.log 2>&1
timeout 300 python tools/replay.py --task reasoning --out $O/r1e_replay_reasoning_tot50.json > $O/replay_tot.log 2>&1
timeout 300 python tools/replay.py --task reasoning --model llama3-8b --out $O/r1e_replay_reasoning_tot50_llama3.json > $O/replay_tot3.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten seq --tree-size 64 --out $O/r1e_replay_speculative_64.json > $O/replay_sd.log 2>&1
timeout 600 python bench.py > $O/r1e_bench_default.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r1e -- python $R/bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $O/r1e_bench_under_rocprof_stats.json 2> $O/rocprof.err
python $R/tools/prof_summary.py /tmp/prof_r1e > $O/r1e_kernel_stats_np.txt 2>&1
tail -c 600 $O/r1e_bench_default.json | head -c 300; echo; head -8 $O/r1e_kernel_stats_np.txt | cut -c1-160
# HBM traffic of the same command: PMC counters, one pass each, with --kernel-trace only (MI355X_MICROARCH.md "HBM")
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c > $O/r1e_pmc_$(echo $c | tr A-Z a-z)_np.json 2>> $O/pmc_$c.err
done
grep -A3 stage1_np $O/r1e_pmc_fetch_size_np.json | head -5; grep -A3 stage1_np $O/r1e_pmc_write_size_np.json | head -5
