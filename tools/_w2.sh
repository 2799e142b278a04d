cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
SH="llama2-7b flatten few_shot:4096:32:1 few_shot:4096:32:25 few_shot:4096:32:50 few_shot:4096:32:100 few_shot:4096:32:125 few_shot:4096:32:150 few_shot:5120:32:10 few_shot:5120:32:50 few_shot:5120:32:125 few_shot:6144:32:10 few_shot:6144:32:25 few_shot:6144:32:60 few_shot:6144:32:100 few_shot:7168:32:25 few_shot:7168:32:100 few_shot:4096:24:30 few_shot:4096:48:20"
bash tools/shape_ab.sh "$SH" deft_amd/lib/libdeft_amd.so deft_amd/lib/libdeft_amd_rules_c5.so deft_amd/lib/libdeft_amd_rules_c6.so deft_amd/lib/libdeft_amd_rules_c7.so deft_amd/lib/libdeft_amd_rules_c8.so
