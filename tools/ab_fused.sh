python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for wl in northstar_4kx32 medusa64_node tot50_4k fewshot_1kx32 forest_8kx8 gqa_4kx32; do
  python bench.py --workload $wl --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['name'], 'us/layer', d['attention_latency_us_per_layer'], 'stage1', (d['roofline'] or {}).get('avg_launch_us'), 'plan us', d['plan_build_us_per_step'])"
done
