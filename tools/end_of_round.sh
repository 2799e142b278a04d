#!/bin/bash
# End-of-round evidence on ONE MI355X box (run as `gpurun -- bash tools/end_of_round.sh [tag]`): replay tables, the default
# bench line, the rocprofv3 kernel-trace summary of the STEP graph alone and the two PMC passes (own runs, --kernel-trace
# only), the two-rank rehearsal of the N > 1 path over gloo, the fuzzers.  Writes gpurun_out/<tag>/<tag>_*; the files judged are
# copied from there into profiles/ (see profiles/README.md).
TAG=${1:-r6z}
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
# replays: through the captured session (default) -- per-step synchronised (per-step attention times) and pipelined (the loop as a
# caller sees it) -- and the eager calls for comparison
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --out $O/${TAG}_replay_few_shot_4kx32.json > $O/replay_fs.log 2>&1
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --pipelined --out $O/${TAG}_replay_few_shot_4kx32_pipelined.json > $O/replay_fsp.log 2>&1
timeout 300 python tools/replay.py --task reasoning --out $O/${TAG}_replay_reasoning_tot50.json > $O/replay_tot.log 2>&1
timeout 300 python tools/replay.py --task reasoning --modes flatten node --pipelined --out $O/${TAG}_replay_reasoning_tot50_pipelined.json > $O/replay_totp.log 2>&1
timeout 300 python tools/replay.py --task reasoning --model llama3-8b --out $O/${TAG}_replay_reasoning_tot50_llama3.json > $O/replay_tot3.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten seq --tree-size 64 --out $O/${TAG}_replay_speculative_64.json > $O/replay_sd.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten --tree-size 64 --pipelined --reps 3 --out $O/${TAG}_replay_speculative_64_pipelined.json > $O/replay_sdp.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten --tree-size 64 --eager --out $O/${TAG}_replay_speculative_64_eager.json > $O/replay_sde.log 2>&1
# the shipped reasoning templates' shape: width 10 per level, a branch (and nine prunes) every 8 steps -- synchronised per step and pipelined
timeout 300 python tools/replay.py --task reasoning --beam 10,12,8 --prompt-len 4096 --modes flatten --out $O/${TAG}_replay_reasoning_beam10x8.json > $O/replay_beam.log 2>&1
timeout 300 python tools/replay.py --task reasoning --beam 10,12,8 --prompt-len 4096 --modes flatten --pipelined --out $O/${TAG}_replay_reasoning_beam10x8_pipelined.json > $O/replay_beamp.log 2>&1
# the reference's OWN reasoning template (first complete tree of docmergeToT.json: 31 nodes, a 1073-token prompt, 2375 decode steps), whole
timeout 600 python tools/replay.py --task reasoning --golden-template docmergeToT --max-gen-len 100000 --modes flatten node --pipelined --out $O/${TAG}_replay_reasoning_docmergeToT_pipelined.json > $O/replay_docmerge.log 2>&1
# round 6: the same replays through sessions that rebuild metadata and plan on every step (DecodeSession(incremental=False), the round-5 loop)
timeout 300 python tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --pipelined --legacy --out $O/${TAG}_replay_few_shot_4kx32_pipelined_rebuild.json > $O/replay_fspl.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten --tree-size 64 --pipelined --legacy --reps 3 --out $O/${TAG}_replay_speculative_64_pipelined_rebuild.json > $O/replay_sdpl.log 2>&1
timeout 300 python tools/replay.py --task speculative_decoding --modes node flatten --tree-size 64 --legacy --out $O/${TAG}_replay_speculative_64_rebuild.json > $O/replay_sdl.log 2>&1
timeout 300 python tools/replay.py --task reasoning --modes flatten node --pipelined --legacy --out $O/${TAG}_replay_reasoning_tot50_pipelined_rebuild.json > $O/replay_totpl.log 2>&1
timeout 300 python tools/replay.py --task reasoning --model llama3-8b --modes flatten --pipelined --out $O/${TAG}_replay_reasoning_tot50_llama3_pipelined.json > $O/replay_tot3p.log 2>&1
timeout 300 python tools/replay.py --task reasoning --model llama3-8b --modes flatten --pipelined --legacy --out $O/${TAG}_replay_reasoning_tot50_llama3_pipelined_rebuild.json > $O/replay_tot3pl.log 2>&1
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench.err
# the N > 1 path on one GPU: two ranks over gloo (RCCL needs a GPU per rank; the driver's 8-GPU run uses it)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --dist-backend gloo --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-traffic 2> $O/bench_2rank.err | grep '^{' > $O/${TAG}_bench_2rank_gloo_one_gpu.json
# ... BASELINE configs[4] at its full rank count: eight ranks, 64 trees, all sharing this GPU over gloo; and RCCL itself at world size 1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --dist-backend gloo --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --no-e2e 2> $O/bench_8rank.err | grep '^{' > $O/${TAG}_bench_8rank_gloo_one_gpu.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-traffic --no-e2e 2> $O/bench_rccl1.err | grep '^{' > $O/${TAG}_bench_rccl_world_size_1.json
# what one cold launch of each BASELINE size can read (stand-alone probe; bench.py measures the same per workload as `ceiling_us`), and
# the per-CU L2 -> LDS rate
(cd tools/probes && make -s launch_ceiling l2_to_lds_bw) > /dev/null 2>&1
timeout 200 tools/probes/launch_ceiling > $O/${TAG}_launch_ceiling.txt 2>&1
timeout 120 tools/probes/l2_to_lds_bw > $O/${TAG}_l2_to_lds_bw.txt 2>&1
# where a stage-1 work item's tile loop spends its time (experiments build)
for wl in northstar_4kx32 gqa_4kx32 medusa64_node; do echo "== $wl"; WL=$wl DEFT_AMD_LIB=$R/deft_amd/lib/libdeft_amd_exp.so timeout 100 python tools/np_phases.py 2>&1 | grep "^rep 2\|^  n="; done > $O/${TAG}_stage1_phases.txt
cd /tmp && export TMPDIR=/tmp
# per-kernel averages of the STEP graph alone (no stage-1-only sweeps, no eager percentiles): sum of the two averages <= step time / layers
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python $R/bench.py --steps 100 --warmup 10 --step-only > $O/${TAG}_bench_under_rocprof_stats.json 2> $O/rocprof.err
python $R/tools/prof_summary.py /tmp/prof_$TAG > $O/${TAG}_kernel_stats.txt 2>&1
# the same command's kernel trace as CSV: durations AND the gaps between consecutive kernels, next to the step time the traced run printed
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py --steps 100 --warmup 10 --step-only > $O/${TAG}_bench_under_rocprof_trace.json 2>/dev/null
python $R/tools/kernel_gaps.py /tmp/kt_$TAG 0.4 > $O/${TAG}_kernel_gaps.txt 2>&1
tail -c 600 $O/${TAG}_bench_default.json | head -c 300; echo; head -8 $O/${TAG}_kernel_stats.txt | cut -c1-160
# the small BASELINE configurations under the kernel trace (configs[2], configs[3]) and head_dim 64
for wl in medusa64_node medusa64_tree_node tot50_4k gqa_4kx32 northstar_4kx32_d64 northstar_4kx32_node_chunk; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$wl -- python $R/bench.py --workload $wl --steps 100 --warmup 10 --step-only > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/prof_${TAG}_$wl 2>&1 | head -6 | cut -c1-160 > $O/${TAG}_kernel_stats_$wl.txt
done
# HBM traffic of the same command: PMC counters, one pass each, with --kernel-trace only (MI355X_MICROARCH.md "HBM")
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --step-only > /dev/null 2> $O/pmc_$c.err
  python $R/tools/pmc_summary.py /tmp/pmc_$c > $O/${TAG}_pmc_$(echo $c | tr A-Z a-z).json 2>> $O/pmc_$c.err
done
grep -A3 stage1_np $O/${TAG}_pmc_fetch_size.json | head -5; grep -A3 stage1_np $O/${TAG}_pmc_write_size.json | head -5
# the whole decode step with the tree advancing, under the kernel trace: what a step costs kernel by kernel
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rp_$TAG -- python $R/tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --no-warmup > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/prof_rp_$TAG 2>&1 | head -18 | cut -c1-160 > $O/${TAG}_replay_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sd_$TAG -- python $R/tools/replay.py --task speculative_decoding --modes flatten --tree-size 64 --pipelined --no-warmup > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/prof_sd_$TAG 2>&1 | head -18 | cut -c1-160 > $O/${TAG}_replay_speculative_kernel_stats.txt
# round 6: what stands between two steps' layers -- window plans against the rebuild-every-step loop (kernel trace of a pipelined replay)
for leg in "" "--legacy"; do
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_b_$TAG$leg -- python $R/tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --pipelined --no-warmup $leg > /dev/null 2>&1
  (echo "== tools/replay.py --task few_shot --width 32 --prompt-len 4096 --max-gen-len 200 --modes flatten --pipelined --no-warmup $leg"; python $R/tools/step_boundary.py /tmp/kt_b_$TAG$leg 1) >> $O/${TAG}_step_boundary.txt 2>&1
done
# ... and attention IN SITU: the model-shaped step (dense layer kernels between the attention calls) under the kernel trace
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_situ_$TAG -- python $R/bench.py --in-situ-only > $O/${TAG}_in_situ_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/prof_situ_$TAG 2>&1 | head -8 | cut -c1-170 > $O/${TAG}_in_situ_kernel_stats.txt
cd $R
# rounds 3, 4 and 5 -- each round's own tree (prev/, see .gitignore) on its own library -- in turn on THIS box: the table of DESIGN 4b
PREV=$(ls -d prev/r* 2>/dev/null | sort | tr '\n' ' ')
if [ -n "$PREV" ]; then
  timeout 1500 tools/ab_rounds.sh "northstar_4kx32 fewshot_1kx32 medusa64_node tot50_4k forest_8kx8 forest_8kx8_single gqa_4kx32 northstar_4kx32_d64 northstar_4kx32_node" $PREV . > $O/${TAG}_ab_rounds.txt 2>&1
  timeout 600 tools/ab_e2e.sh $(ls -d prev/r* | sort | tail -1) . > $O/${TAG}_ab_e2e.txt 2>&1
fi
# the whole GPU suite on this box
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/${TAG}_gputest.txt
# fuzzers: the captured loop against the eager path bit for bit (with speculative-decoding merge / reset steps); every step of random
# replays against fp64 attention
(timeout 400 python tools/fuzz_session.py 240 ${FUZZ_SEED:-31} mix 2>&1 | tail -2; timeout 300 python tools/fuzz_replay.py 180 ${FUZZ_SEED:-31} 2>&1 | tail -2) > $O/${TAG}_fuzz.txt
cat $O/${TAG}_fuzz.txt
