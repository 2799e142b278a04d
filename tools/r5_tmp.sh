cd $GRAFT_REPO_ROOT; O=gpurun_out/r5k; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_list or golden or flatten" 2>&1 | tail -2 > $O/merge_small2.txt
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
for wl in medusa64_node northstar_4kx32 tot50_4k gqa_4kx32 medusa64_tree_flatten; do echo "== $wl"; python tools/ab_step.py --workload $wl --steps 100 --rounds 2 DEFT_MERGE_SMALL=0,1 2>&1 | tail -2; done >> $O/merge_small2.txt
(bash tools/kernel_stats_knob.sh medusa64_node DEFT_MERGE_SMALL=0; bash tools/kernel_stats_knob.sh northstar_4kx32 DEFT_MERGE_SMALL=0) 2>&1 | grep "==\|merge_kernel" >> $O/merge_small2.txt
cat $O/merge_small2.txt
