#!/bin/bash
# GPU box: one library per variant object of scripts/probe/hazard/out/ (stock objects + the variant's cond_split_kernels), then the full-batch
# determinism check (scripts/probe/rgcheck.py: 3 launches of the C3 step per trial, every 192-row replica against the small-batch values) on each.
#   bash scripts/probe/hazard/run_variants.sh [trials] [log2 rows] [variant ...]
set -u
TRIALS=${1:-4}; LOG2=${2:-18}; shift 2 2>/dev/null
STEM=${HZ_STEM:-cond_split_kernels}
OUT=scripts/probe/hazard/out
VARS="$*"
[ -z "$VARS" ] && VARS=$(ls $OUT/${STEM}_*.o | sed "s#.*/${STEM}_##; s#\.o##")
STOCK=$(ls jammy_flows_amd/csrc/*.o | grep -v "/${STEM}.o")
mkdir -p gpurun_out
for v in stock $VARS; do
  lib=/tmp/lib_$v.so
  if [ $v = stock ]; then lib=$PWD/jammy_flows_amd/libjammy_hip.so
  else /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $STOCK $OUT/${STEM}_$v.o -o $lib || { echo "link failed: $v"; continue; }; fi
  for rg in ${HZ_RGS:-2 1}; do
    log2=$LOG2; [ $rg = 1 ] && log2=$((LOG2 + 2))
    r=$(JF_LIB=$lib JF_CS_RG=$rg JF_TRIALS=$TRIALS JF_LOG2=$log2 timeout 300 python3 scripts/probe/rgcheck.py 2>&1 | grep -E "deterministic|bad rows")
    nbad=$(echo "$r" | grep "bad rows" | awk '{s += ($3 > 0)} END {print s + 0}')
    nl=$(echo "$r" | grep -c "bad rows")
    ndet=$(echo "$r" | grep deterministic | grep -c "\[True, True\]")
    echo "variant $v rg=$rg rows=2^$log2: launches with bad rows $nbad / $nl, trials fully deterministic $ndet / $TRIALS" | tee -a gpurun_out/hazard_variants.txt
  done
done
