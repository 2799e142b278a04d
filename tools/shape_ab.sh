#!/bin/bash
# tools/shape_ab.sh "<model> <mode> <shape> ..." <lib.so> [<lib.so> ...]   -- tools/shape_ab.py under each library, same box
ARGS=$1; shift
for lib in "$@"; do
  echo "== $(basename $lib)"
  DEFT_AMD_LIB=$(realpath $lib) python tools/shape_ab.py $ARGS 2>/dev/null
done
