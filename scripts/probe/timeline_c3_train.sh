#!/bin/bash
# kernel timeline of C3 training steps (start / end / queue per kernel) -> gpurun_out/timeline_c3_train.txt
export TMPDIR=/tmp
G=gpurun_out/timeline_c3_train; rm -rf $G; mkdir -p $G
rocprofv3 --kernel-trace -d $G/t -- python3 bench.py --no-pmc --no-sweep --no-cpu-baseline --workload c3 --scaling weak --train --steps 6 --warmup 3 > $G/t.log 2>&1
f=$(find $G/t -name "*.db" | head -1)
python3 - "$f" > gpurun_out/timeline_c3_train.txt 2>&1 <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels") or t == "kernels"]
print(tabs[:60])
view = "kernels" if "kernels" in tabs else kd[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
print(view, cols)
name_c = "name" if "name" in cols else "kernel_name"
q = "select %s, start, end, queue_id, stream_id from %s order by start" % (name_c, view) if "stream_id" in cols else "select %s, start, end, queue_id, 0 from %s order by start" % (name_c, view)
rows = list(cur.execute(q))
print(len(rows), "kernels")
# the last 700 kernels: the graph-replayed steps
t0 = rows[-700][1] if len(rows) > 700 else rows[0][1]
for n, s, e, qid, sid in rows[-700:]:
    print("%10.1f %8.1f q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qid, sid, n[:90]))
PY
rm -rf $G/t
