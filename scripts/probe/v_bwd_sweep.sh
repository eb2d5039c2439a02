#!/bin/bash
# 'v' backward kernel, C5's block at 2^17 rows: lanes per row (JF_V_BWD_LANES = 1 / 4 / 8) and the dual-number check build of the same kernel
for lv in 1 4 8; do
  echo "lanes $lv: $(JF_V_BWD_LANES=$lv timeout 120 python3 scripts/probe/v_bwd_run.py 2>&1 | tail -1)"
done
echo "dual-number replay (JF_V_BWD_DUAL=1): $(JF_V_BWD_DUAL=1 timeout 120 python3 scripts/probe/v_bwd_run.py 2>&1 | tail -1)"
