#!/bin/bash
# late round-6 checks after the wider narrow-MLP kernel and the approach phase of the general-option sampler: tests, the c3b lines, both scans
export TMPDIR=/tmp
G=gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_split_gemm.py -q -x -k "mlp2" > $G/late_mlp2.txt 2>&1; tail -3 $G/late_mlp2.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k "sampling or newton_rule or roundtrip or solver or random_conf" > $G/late_sampling.txt 2>&1; tail -3 $G/late_sampling.txt
python3 bench.py --workload c3b --no-sweep > $G/bench_r06_c3b.json 2> $G/bench_r06_c3b.err
python3 bench.py --no-pmc --workload c3b --scaling weak --train > $G/bench_r06_c3b_train.json 2> $G/bench_r06_c3b_train.err
python3 bench.py --no-pmc --workload c3b --scaling weak --direction sample > $G/bench_r06_c3b_sample.json 2> $G/bench_r06_c3b_sample.err
python3 scripts/probe/scan_fixtures.py 65536 "" f64 > $G/scan_r06_f64.txt 2>&1
python3 scripts/probe/scan_fixtures.py 65536 "" f32 > $G/scan_r06_f32.txt 2>&1
tail -2 $G/scan_r06_f32.txt
