export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
run() { wl=$1; shift; env "$@" python bench.py --workload $wl --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['name'], '$*', d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us'))"; }
for rep in 1 2; do
run forest_8kx8_single A=1; run forest_8kx8_single DEFT_NP_CHUNK=3
run gqa_4kx32 A=1; run gqa_4kx32 DEFT_NP_UNION=2
run northstar_4kx32_d64 A=1; run northstar_4kx32_d64 DEFT_NP_UNION=4
run tot50_4k A=1; run tot50_4k DEFT_NP_UNION=2
done
