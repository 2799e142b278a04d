cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for wl in medusa64_node medusa64_tree_node medusa64_tree_flatten; do
  for c in 1 2 3 4; do
    export DEFT_AMD_LIB=$GRAFT_REPO_ROOT/deft_amd/lib/libdeft_amd_rules_c$c.so
    python bench.py --workload $wl --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl C=$c  layer %7.2f  stage1 %s' % (d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us')))"
  done
  unset DEFT_AMD_LIB
  python bench.py --workload $wl --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl shipped  layer %7.2f  stage1 %s' % (d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us')))"
done
for L in 75 150; do
  for c in 2 3 4; do
    export DEFT_AMD_LIB=$GRAFT_REPO_ROOT/deft_amd/lib/libdeft_amd_rules_c$c.so
    python bench.py --workload fewshot_1kx32 --branch-len $L --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1k L=$L C=$c  layer %7.2f  stage1 %s' % (d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us')))"
  done
done
