#!/bin/bash
# Probe builds behind the float32 error budget: the library with ONE hardware approximation replaced by its correctly rounded
# library function (csrc/jf_math.h JF_PROBE_ACCURATE_*), plus all three.  Run in the build container from the repo root (hipcc cross-compiles);
# the libraries land under build_probe/f32_budget/<variant>/ (git-ignored, they travel to the GPU box), scripts/probe/f32_error_budget.py runs them.
set -e
ROOT=$(pwd)
for v in base exp log rcp all; do
  d=$ROOT/build_probe/f32_budget/$v
  rm -rf $d; mkdir -p $d/jammy_flows_amd/csrc $d/include
  cp jammy_flows_amd/csrc/*.hip jammy_flows_amd/csrc/*.h jammy_flows_amd/csrc/Makefile $d/jammy_flows_amd/csrc/
  cp include/*.h $d/include/
  case $v in
    base) X="";;
    exp) X="-DJF_PROBE_ACCURATE_EXP";;
    log) X="-DJF_PROBE_ACCURATE_LOG";;
    rcp) X="-DJF_PROBE_ACCURATE_RCP";;
    all) X="-DJF_PROBE_ACCURATE_EXP -DJF_PROBE_ACCURATE_LOG -DJF_PROBE_ACCURATE_RCP";;
  esac
  make -C $d/jammy_flows_amd/csrc -j${JOBS:-8} CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=fast -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops $X" > $d/build.log 2>&1
  mv $d/jammy_flows_amd/libjammy_hip.so $d/libjammy_hip.so
  rm -rf $d/jammy_flows_amd $d/include
  echo "$v: $(ls -la $d/libjammy_hip.so | awk '{print $5}') bytes"
done
