export DEFT_AMD_LIB=$GRAFT_REPO_ROOT/deft_amd/lib/libdeft_amd_exp.so
for wl in northstar_4kx32 fewshot_1kx32 tot50_4k medusa64_node forest_8kx8; do
  echo "== $wl"
  timeout 400 python tools/ab_step.py --workload $wl --steps 40 --rounds 2 DEFT_NP_FAST=0,512,768,100000 2>&1 | grep "\->"
done
