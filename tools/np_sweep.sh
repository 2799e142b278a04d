export DEFT_STAGE1_KERNEL=np
echo "== len1"; python tools/ab.py --branch-len 1 --reps 3 --rounds 1 DEFT_NP_CHUNK=1,2,4,8 2>&1 | grep "\->"
echo "== tot50"; python tools/ab.py --workload tot50_4k --reps 3 --rounds 1 DEFT_NP_CHUNK=1,2,4,8 DEFT_NP_UNION=1,2,4 2>&1 | grep "\->"
echo "== forest"; python tools/ab.py --workload forest_8kx8 --reps 3 --rounds 1 DEFT_NP_CHUNK=2,4,8 DEFT_NP_UNION=1,2,4 2>&1 | grep "\->"
echo "== forest single"; python tools/ab.py --workload forest_8kx8_single --reps 3 --rounds 1 DEFT_NP_CHUNK=1,2,4,8 2>&1 | grep "\->"
echo "== 1kx32"; python tools/ab.py --workload fewshot_1kx32 --reps 3 --rounds 1 DEFT_NP_CHUNK=2,4,8 DEFT_NP_UNION=2,4 2>&1 | grep "\->"
