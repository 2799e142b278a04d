#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <libA.so> <libB.so> [workload ...]
# (process-to-process and box-to-box noise is ~1 us on the stage-1 kernel; build the other variant with
#  `hipcc ... -o deft_amd/lib/libdeft_amd_prev.so` from a checkout of the other revision's csrc/)
A=$1; B=$2; shift 2
WLS=${@:-northstar_4kx32 fewshot_1kx32 tot50_4k forest_8kx8 northstar_4kx32_seq medusa64_node}
for rep in 1 2; do
for lib in $A $B; do
  export DEFT_AMD_LIB=$(realpath $lib)
  echo "== $lib (rep $rep)"
  for wl in $WLS; do
    python bench.py --workload $wl --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['name'], d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us'))"
  done
done
done
