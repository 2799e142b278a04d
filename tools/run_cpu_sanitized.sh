#!/bin/bash
# The host-side C++ of the library (deft_amd/csrc/tree.cpp: the native tree, its layout and journal; host.cpp: the metadata
# builder and the error text -- 1170 lines) built with AddressSanitizer + UndefinedBehaviorSanitizer and driven by the CPU test
# suites that exercise it.  CPU only (GPU sanitizers are not available on this pool): every entry point that lives in
# deft_kernels.hip is a STUB here that returns DEFT_EUNSUPPORTED, generated from the library's own symbol table, so that
# deft_amd/_lib.py finds all its names (74).
#   tools/run_cpu_sanitized.sh [pytest args ...]        (default: tests/test_host_logic.py tests/test_forest_tree.py tests/test_replay.py
#                                                         tests/test_replay_golden.py tests/test_workloads.py, -m "not gpu")
# A sanitizer report makes the run fail (halt_on_error, abort).  `make -C deft_amd/csrc asan` builds the library only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $ROOT/deft_amd/csrc asan
export DEFT_AMD_LIB=$ROOT/build/asan/libdeft_amd_asan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
# (python itself is not instrumented: the sanitizer runtime has to be in the process first)
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
cd $ROOT
if [ $# -eq 0 ]; then set -- tests/test_host_logic.py tests/test_forest_tree.py tests/test_replay.py tests/test_replay_golden.py tests/test_workloads.py; fi
# (not under the sanitized library: the tests that compare `nm -D` of the SHIPPED library with the header, the C program linked
#  against it, and the two that call entry points of deft_kernels.hip -- stubs here)
python -m pytest -q -m "not gpu" -p no:cacheprovider "$@" -k "not exports and not plain_c_program and not supported_geometries and not argument_errors and not stage_copy"
