#!/bin/bash
# Hardware counters of stage1_np_kernel for a workload, one rocprofv3 pass per group (each with --kernel-trace only), averaged per launch:
#   tools/pmc_stage1.sh <workload> > profiles/<name>.txt          (on the GPU box: gpurun -- bash tools/pmc_stage1.sh gqa_4kx32)
# What they answer: how busy are a launch's waves (SQ_BUSY / WAVE cycles, waiting vs issuing), how many VALU / MFMA / LDS / VMEM
# instructions they issue, how many read requests go TCP -> L2 and how many of those hit, how busy the texture addresser is.
WL=${1:-gqa_4kx32}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE")
echo "# $WL: rocprofv3 --pmc <group> --kernel-trace on bench.py --workload $WL --step-only; per launch of stage1_np_kernel (mean)"
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmc_s1_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_s1_$i -- python $R/bench.py --workload $WL --steps 6 --warmup 2 --step-only > /dev/null 2>/tmp/pmc_s1_$i.err
  python - "$i" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(f"/tmp/pmc_s1_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "stage1_np_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in acc.items():
    print(f"{c:36s} {sum(v) / len(v):16.1f}   ({len(v)} launches)")
PY
done
