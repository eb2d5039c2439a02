#!/bin/bash
export TMPDIR=/tmp
G=gpurun_out
rm -f $G/ab_c3_train_*.json
for i in 1 2; do
JF_BACKWARD_FORK=0 JF_COMBINE_ROWS_FN=0 python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train > $G/ab_c3_train_base_$i.json 2>/dev/null
JF_BACKWARD_FORK=0 JF_COMBINE_ROWS_FN=1 python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train > $G/ab_c3_train_combine_$i.json 2>/dev/null
JF_BACKWARD_FORK=1 JF_COMBINE_ROWS_FN=0 python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train > $G/ab_c3_train_fork_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_c3_train_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],4), d.get('eager',{}).get('ms_per_step'), d.get('final_loss'))
PY
