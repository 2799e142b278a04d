#!/bin/bash
# Same-box comparison of the ADVANCING decode loop (bench.py `end_to_end`: host slot allocation, nq slot numbers over PCIe, device tree
# advance, TreeMetadata + plan on the GPU, 32 layers) between trees (checkouts under prev/ or `.`), each on its own library:
#   tools/ab_e2e.sh <tree> [<tree> ...]     prints graphed / eager ms per step, the frozen step at the loop's mean length and their ratio
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
for tree in "$@"; do
  T=$(cd "$ROOT/$tree" 2>/dev/null && pwd || (cd "$tree" && pwd))
  (cd $T && unset DEFT_AMD_LIB && PYTHONPATH=$T python bench.py --no-cpu-baseline --no-extras --no-traffic --no-cfg5 --steps 100 2>/dev/null) |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print('%-10s rep $rep  graphed %.4f  eager %.4f  frozen %.4f  ratio %s' % ('$tree', e['graphed']['ms_per_step'], e['eager']['ms_per_step'], e['frozen_step_at_mean_len']['ms_per_step'], e['graphed'].get('over_frozen_step_at_mean_len')))" 2>/dev/null || echo "$tree rep $rep: failed"
done
done
