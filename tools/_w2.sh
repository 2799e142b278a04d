cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for L in 128 200 256 300 384 400; do
  echo "== branch-len $L"
  BENCH_EXTRA="--branch-len $L" bash tools/ab_rules.sh "northstar_4kx32" deft_amd/lib/libdeft_amd_rules_nosolo.so deft_amd/lib/libdeft_amd.so
done
echo "== other workloads"
bash tools/ab_rules.sh "fewshot_1kx32 medusa64_tree_flatten northstar_4kx32_d64 tot50_4k" deft_amd/lib/libdeft_amd_rules_nosolo.so deft_amd/lib/libdeft_amd.so
