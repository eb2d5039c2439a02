#!/bin/bash
# C3 shard step (2^17 rows), N > 1 path forced on one GPU (one-rank RCCL group): steps per all-gather
for g in 8 12 16 32 16 8; do
  JF_FORCE_COLLECTIVES=1 python3 bench.py --batch 131072 --gather-steps $g --no-cpu-baseline --no-pmc --no-sweep --steps 800 --warmup 40 2>/dev/null > gpurun_out/shard_g$g.json
  python3 -c "import json; d=json.loads(open('gpurun_out/shard_g$g.json').readline()); print('gather-steps $g ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'))"
done
python3 bench.py --batch 131072 --no-cpu-baseline --no-pmc --no-sweep --steps 800 --warmup 40 2>/dev/null > gpurun_out/shard_none.json
python3 -c "import json; d=json.loads(open('gpurun_out/shard_none.json').readline()); print('no exchange ms', d['ms_per_step'], 'host issue', d.get('host_issue_ms_per_step'))"
