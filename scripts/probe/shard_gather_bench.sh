#!/bin/bash
# C3 shard step (2^17 rows) with the N > 1 path forced on one GPU: one all-gather per step, one per four steps, no exchange
for g in 1 4; do
  JF_FORCE_COLLECTIVES=1 python3 bench.py --batch 131072 --gather-steps $g --no-cpu-baseline --no-pmc --no-sweep --steps 400 --warmup 20 2>/dev/null > gpurun_out/shard_g$g.json
  python3 -c "import json; d=json.loads(open('gpurun_out/shard_g$g.json').readline()); print('gather-steps $g ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'), d.get('exchange'))"
done
python3 bench.py --batch 131072 --no-cpu-baseline --no-pmc --no-sweep --steps 400 --warmup 20 2>/dev/null > gpurun_out/shard_none.json
python3 -c "import json; d=json.loads(open('gpurun_out/shard_none.json').readline()); print('no exchange ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'))"
JF_RCCL_DIRECT=1 JF_FORCE_COLLECTIVES=1 python3 bench.py --batch 131072 --gather-steps 1 --no-cpu-baseline --no-pmc --no-sweep --steps 400 --warmup 20 2>/dev/null > gpurun_out/shard_direct.json
python3 -c "import json; d=json.loads(open('gpurun_out/shard_direct.json').readline()); print('direct g1 ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'), d.get('exchange'))"
JF_RCCL_DIRECT=1 JF_FORCE_COLLECTIVES=1 python3 bench.py --batch 131072 --gather-steps 4 --no-cpu-baseline --no-pmc --no-sweep --steps 400 --warmup 20 2>/dev/null > gpurun_out/shard_direct4.json
python3 -c "import json; d=json.loads(open('gpurun_out/shard_direct4.json').readline()); print('direct g4 ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'), d.get('exchange'))"
