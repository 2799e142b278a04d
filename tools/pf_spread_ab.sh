set -e
mkdir -p gpurun_out/r4p
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
for v in 1 2; do DEFT_PREFILL_SPREAD=$v python -m pytest tests/test_prefill.py -m gpu -x -q 2>&1 | tail -2; done
for rep in 1 2; do for v in 0 1 2; do echo "== DEFT_PREFILL_SPREAD=$v"; DEFT_PREFILL_SPREAD=$v python tools/prefill_bench.py | python -c "
import sys,json
print('  '.join('%s:%d:%.1f'%(r['model'][5:7],r['S'],r['TFLOPs']) for r in map(json.loads,sys.stdin)))"; done; done
