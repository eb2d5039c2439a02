#!/bin/bash
# hazard / race hunt for the two-row-group block kernel: edited copies of the tree under /tmp, rebuilt on the GPU box
set -e
for v in A B C; do
  rm -rf /tmp/var$v; mkdir -p /tmp/var$v; cp -r jammy_flows_amd tests include /tmp/var$v/
  f=/tmp/var$v/jammy_flows_amd/csrc/cond_split_kernels.hip
  case $v in
    A) sed -i 's/asm("s_nop 1\\n\\tv_permlane16_swap_b32 %0, %1"/asm volatile("s_nop 7\\n\\tv_permlane16_swap_b32 %0, %1\\n\\ts_nop 7"/; s/asm("s_nop 1\\n\\tv_permlane32_swap_b32 %0, %1"/asm volatile("s_nop 7\\n\\tv_permlane32_swap_b32 %0, %1\\n\\ts_nop 7"/' $f ;;
    B) python3 - $f <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
i=s.index("template <typename Op> __device__ __forceinline__ float cs_rreduce"); j=s.index("__device__ __forceinline__ float cs_rsum")
s=s[:i]+"template <typename Op> __device__ __forceinline__ float cs_rreduce(float v, Op op) { v = op(v, __shfl_xor(v, 16)); return op(v, __shfl_xor(v, 32)); }\n"+s[j:]
open(p,"w").write(s)
PY
    ;;
    C) sed -i 's/auto landed = \[&\]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_barrier" ::: "memory"); };/auto landed = [\&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_barrier\\n\\ts_nop 7" ::: "memory"); __syncthreads(); };/' $f ;;
  esac
  diff <(cat jammy_flows_amd/csrc/cond_split_kernels.hip) $f | head -8
  (cd /tmp/var$v/jammy_flows_amd/csrc && rm -f cond_split_kernels.o && make >/dev/null 2>&1)
  echo "== variant $v"; JF_ROOT=/tmp/var$v JF_CS_RG=2 python3 scripts/probe/rgcheck.py 2>&1 | grep deterministic
done
