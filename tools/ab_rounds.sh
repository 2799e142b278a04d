#!/bin/bash
# Same-box comparison of ROUNDS: each tree (a checkout of an earlier round under prev/, built in place -- see .gitignore -- or `.`)
#   (to make one:  mkdir -p prev/r4 && git archive <the round's last commit> | tar -x -C prev/r4 && make -C prev/r4/deft_amd/csrc ../lib/libdeft_amd.so
#    -- round 3 ended at aecaf97~1, round 4 at 9cddd5b~1; prev/ is git-ignored and travels to the GPU box with gpurun)
# runs ITS OWN bench.py on ITS OWN library, in turn, twice:
#   tools/ab_rounds.sh "<workload> ..." <tree> [<tree> ...]        e.g.  tools/ab_rounds.sh "northstar_4kx32 medusa64_node" prev/r3 prev/r4 .
# prints us per layer of the captured 32-layer step and the stage-1 average (HIP events) per tree, workload and repetition.
# (Round tables that compare the driver's box of one round with the builder's box of the next are off by the box-to-box spread,
#  up to 5 % on the small trees: VERDICT r4 weak #9.)
WLS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
for tree in "$@"; do
  T=$(cd "$ROOT/$tree" 2>/dev/null && pwd || (cd "$tree" && pwd))
  for wl in $WLS; do
    (cd $T && unset DEFT_AMD_LIB && PYTHONPATH=$T python bench.py --workload $wl $BENCH_EXTRA --no-cpu-baseline --no-extras --no-traffic --no-e2e --no-cfg5 --steps 100 2>/dev/null) |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s rep $rep  %-26s %7.2f  %s' % ('$tree', d['config']['name'], d['attention_latency_us_per_layer'], (d.get('roofline') or {}).get('avg_launch_us')))" 2>/dev/null || echo "$tree rep $rep $wl: failed"
  done
done
done
