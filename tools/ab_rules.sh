#!/bin/bash
# Same-box comparison of rule variants of the SHIPPED code (make -C deft_amd/csrc rules KNOBS=... NAME=...):
#   [BENCH_EXTRA="--branch-len 400"] tools/ab_rules.sh "<workload> ..." <lib.so> [<lib.so> ...]   (us per layer, stage-1 us; two rounds)
WLS=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  export DEFT_AMD_LIB=$(realpath $lib)
  for wl in $WLS; do
    python bench.py --workload $wl $BENCH_EXTRA --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s rep $rep  %-22s %7.2f  %s' % ('$(basename $lib)', d['config']['name'], d['attention_latency_us_per_layer'], (d['roofline'] or {}).get('avg_launch_us')))"
  done
done
done
