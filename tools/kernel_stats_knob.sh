# Per-kernel durations (rocprofv3 --kernel-trace --stats) of one workload's step under plan knobs of the experiments build:
#   tools/kernel_stats_knob.sh <workload> VAR=v [VAR=v ...]      (one traced bench.py --step-only run per setting)
export DEFT_AMD_LIB=$PWD/deft_amd/lib/libdeft_amd_exp.so
R=$PWD; cd /tmp; export TMPDIR=/tmp
wl=$1; shift
for kv in A=0 "$@"; do
  d=/tmp/ksk_$$_$(echo $kv | tr -c 'A-Za-z0-9' _)
  env $kv timeout 300 rocprofv3 --kernel-trace --stats -d $d -- python $R/bench.py --workload $wl --steps 100 --warmup 10 --step-only > /dev/null 2>&1
  echo "== $wl $kv"; python $R/tools/prof_summary.py $d 2>&1 | head -5 | cut -c1-150
done
