# Per-kernel durations (rocprofv3 --kernel-trace --stats) of one workload's step under several builds of the library:
#   tools/kernel_stats_libs.sh <workload> <lib.so> [<lib.so> ...]
R=$PWD; wl=$1; shift
libs=$(for l in "$@"; do realpath $l; done)
cd /tmp; export TMPDIR=/tmp
for lib in $libs; do
  d=/tmp/ksl_$$_$(basename $lib .so)
  DEFT_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $d -- python $R/bench.py --workload $wl --steps 100 --warmup 10 --step-only > /dev/null 2>&1
  echo "== $wl $(basename $lib)"; python $R/tools/prof_summary.py $d 2>&1 | sed -n 3,4p | cut -c1-150
done
