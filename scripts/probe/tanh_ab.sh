#!/bin/bash
# A/B of the float64 hidden-layer tanh: the product library (table, jf_math.h: tanh_tab) against a private build with the exp-based tanh_fast.
#   scripts/probe/tanh_ab.sh   (on the GPU box; builds into build_probe/)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_probe/notab
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=fast -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -DJF_PROBE_NO_TANH_TAB"
for f in jammy_flows_amd/csrc/*.hip; do
  o=build_probe/notab/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_probe/notab/*.o -o build_probe/notab/libjammy_hip.so
for lib in "" build_probe/notab/libjammy_hip.so; do
  echo "== ${lib:-product (tanh table)}"
  JF_LIB_PATH=$lib python3 scripts/probe/tanh_ab.py
done
