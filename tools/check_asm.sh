#!/bin/bash
# Build-time check for the fixed VGPRs named in inline asm (v255 in stage1_np.h, v200/v201 in stage1_stream.h):
# the compiler's own allocation, single registers and tuples, must never reach them.
set -e
cd "$(dirname "$0")/.."
make -C deft_amd/csrc asm > /dev/null
S=build/asm/deft_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
check() {  # kernel symbol, first reserved register, allowed literal uses
  local sym=$1 lo=$2
  local start=$(grep -n "^$sym:" $S | cut -d: -f1)
  awk -v s=$start 'NR>=s' $S | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/_k.s
  # highest register index touched by any operand other than the asm statements that name the reserved register
  grep -v "global_atomic_add v$3\|v_readfirstlane_b32 s[0-9]*, v$3\|global_atomic_add v$4\|v_readfirstlane_b32 s[0-9]*, v$4" /tmp/_k.s \
    | grep -o "v\[[0-9]*:[0-9]*\]\|v[0-9][0-9]*" | sed 's/v\[\([0-9]*\):\([0-9]*\)\]/\2/; s/^v//' | sort -n | tail -1 > /tmp/_max
  local max=$(cat /tmp/_max)
  echo "$sym: highest compiler-allocated VGPR v$max, reserved from v$lo"
  [ "$max" -lt "$lo" ]
}
check _ZN4deft16stage1_np_kernelILi128ELb1EEEvNS_8NpParamsE 255 255 255
check _ZN4deft20stage1_stream_kernelILi128EEEvNS_12StreamParamsE 200 200 201
