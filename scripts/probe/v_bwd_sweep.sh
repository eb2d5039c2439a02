#!/bin/bash
# 'v' backward kernel, C5's block at 2^17 rows: the product kernel (hand-written reverse mode) and the dual-number check build of the same kernel
# (the lanes-per-row / register-cap variants of section 7 were instantiated for the experiment only)
echo "reverse mode: $(timeout 120 python3 scripts/probe/v_bwd_run.py 2>&1 | tail -1)"
echo "dual-number replay (JF_V_BWD_DUAL=1): $(JF_V_BWD_DUAL=1 timeout 120 python3 scripts/probe/v_bwd_run.py 2>&1 | tail -1)"
