#!/bin/bash
export TMPDIR=/tmp
G=gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_split_gemm.py -q -x -k "mlp2" > $G/late_mlp2.txt 2>&1; tail -3 $G/late_mlp2.txt
timeout 900 python3 -m pytest tests/test_gpu_grad.py -q -x > $G/late_grad.txt 2>&1; tail -3 $G/late_grad.txt
python3 bench.py --no-pmc --workload c3b --scaling weak --train > $G/bench_r06_c3b_train.json 2> $G/bench_r06_c3b_train.err
( for a in "r_i1 f32 3000" "r_i1 f64 3000" "f_s2_cond_ff f32 2000"; do python3 scripts/probe/stall_probe.py $a; done ) > $G/stall_probe.txt 2>&1
