#!/bin/bash
# private build of the library with the step trace of cond_pp_kernels.hip compiled in (-DPP_TRACE), then the timeline script (GPU box)
set -e
cd "$(dirname "$0")/../../jammy_flows_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=fast -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops \
    -DPP_TRACE "$@" -c cond_pp_kernels.hip -o /tmp/cond_pp_trace.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v cond_pp_kernels.o) /tmp/cond_pp_trace.o -o /tmp/libjammy_trace.so
cd ../..
JF_LIB_PATH=/tmp/libjammy_trace.so python3 scripts/probe/pp_trace.py
