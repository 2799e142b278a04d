# quick look at the one-wave-per-SIMD prefill kernel (experiments build): parity tests, cycles per step, throughput
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
DEFT_PREFILL_64=1 timeout 600 python -m pytest tests/test_prefill.py -m gpu -x -q 2>&1 | tail -3
DEFT_PREFILL_64=1 python tools/prefill64_cycles.py 16384 2>&1 | grep "wave [03]"
for v in 1 0; do echo "== DEFT_PREFILL_64=$v"; DEFT_PREFILL_64=$v timeout 300 python tools/prefill_bench.py 2>/dev/null | python -c "
import sys,json
print('  '.join('%s:%d:%.1f'%(r['model'][5:7],r['S'],r['TFLOPs']) for r in map(json.loads,sys.stdin)))"; done
