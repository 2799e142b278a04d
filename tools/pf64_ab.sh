# A/B of the one-wave-per-SIMD prefill kernel (experiments build, DEFT_PREFILL_64=1) against the shipped 8-wave kernel
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
DEFT_PREFILL_64=1 timeout 600 python -m pytest tests/test_prefill.py -m gpu -x -q 2>&1 | tail -15
for rep in 1 2; do for v in 0 1; do echo "== DEFT_PREFILL_64=$v"; DEFT_PREFILL_64=$v timeout 300 python tools/prefill_bench.py 2>/dev/null | python -c "
import sys,json
print('  '.join('%s:%d:%.1f'%(r['model'][5:7],r['S'],r['TFLOPs']) for r in map(json.loads,sys.stdin)))"; done; done
