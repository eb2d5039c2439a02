#!/bin/bash
export TMPDIR=/tmp
G=gpurun_out
timeout 1500 python3 -m pytest tests -q -m gpu -k "g_e40 or g_e64 or caps or block_sums or g_e20 or g_e10 or e12" > $G/late4_tests.txt 2>&1; tail -15 $G/late4_tests.txt
