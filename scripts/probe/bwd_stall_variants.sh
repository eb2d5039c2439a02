#!/bin/bash
# Where does cond_gf_split_bwd_kernel wait?  (38 % vector, 21 % matrix, 4 % LDS busy: profiles/r06_c3_train_pmc.txt.)  Builds timing-only variants
# of the library (WRONG results by construction) into jammy_flows_amd/_probe/ -- run here (build container), then scripts/probe/bwd_stall_probe.py
# on the GPU box times the adjoint launch under each:
#   nodma      the chunk streams are not issued after the first one (stale weight chunks in LDS): the cost of streaming W2 / W2^T per 64 rows
#   nobarrier  the steps wait for their own loads but not for the other waves (racy): the cost of the 32 workgroup barriers per 64 rows
#   both
set -e
cd "$(dirname "$0")/../../jammy_flows_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=fast -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -I. -I../../include"
mkdir -p /tmp/bwdvar ../_probe
python3 - <<'PY'
src = open("cond_bwd_kernels.hip").read()
a = "    auto dma = [&](int l, int c, int buf) {\n"
b = 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_barrier" ::: "memory");'
assert a in src and b in src
nodma = src.replace(a, a + "        if (l + c != 0) return;\n")
nobar = src.replace(b, 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");')
both = nodma.replace(b, 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");')
for n, s in (("nodma", nodma), ("nobarrier", nobar), ("both", both)):
    open("/tmp/bwdvar/cond_bwd_kernels_%s.hip" % n, "w").write(s)
PY
OBJS=$(ls *.o | grep -v cond_bwd_kernels.o)
for v in nodma nobarrier both; do
  /opt/rocm/bin/hipcc $FL -c /tmp/bwdvar/cond_bwd_kernels_$v.hip -o /tmp/bwdvar/cb_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/bwdvar/cb_$v.o -o ../_probe/libjammy_hip_$v.so
  echo built $v
done
