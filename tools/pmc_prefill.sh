#!/bin/bash
# Hardware counters of the prefill kernel (one rocprofv3 pass per group, --kernel-trace only), per launch (mean):
#   gpurun -- bash tools/pmc_prefill.sh [S] > profiles/<name>.txt
# (round 4 ran it on two kernels -- the shipped one and a one-wave-per-SIMD form, profiles/r4_pmc_prefill_16k.txt; per MFMA both
#  issue ~9 instructions and SQ_VALU_MFMA_COEXEC_CYCLES is 34-39 % of the MFMA-busy cycles)
S=${1:-16384}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU2" "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT")
echo "# prefill, llama2-7b, $S tokens: rocprofv3 --pmc <group> --kernel-trace; per launch (mean)"
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmc_pf_$i
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_pf_$i -- python $R/tools/prefill_one.py $S llama2-7b 3 > /dev/null 2>/tmp/pmc_pf_$i.err
  python - "$i" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(f"/tmp/pmc_pf_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "prefill" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in acc.items():
    print(f"{c:36s} {sum(v) / len(v):18.1f}   ({len(v)} launches)")
PY
done
