# Whole-layer effect of stage-1 plan knobs (experiments build): tools/knob_layer.sh <workload> VAR=v [VAR=v ...]   (one bench.py run per setting)
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
wl=$1; shift
for rep in 1 2; do
for kv in A=0 "$@"; do
  env $kv python bench.py --workload $wl --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['name'], '$kv', d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us'))"
done; done
