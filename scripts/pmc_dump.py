#!/usr/bin/env python3
"""print every counter of every jf:: kernel in a rocprofv3 --pmc result database (rocpd sqlite):  python scripts/pmc_dump.py <db> [...]"""
import sqlite3
import sys

for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name "
         "order by kernel_name, counter_name")
    for n, c, k, v in cur.execute(q):
        if "jf::" in n:
            print("%-60s %-34s n=%-3d mean=%.4g" % (n.replace("void ", "")[:60], c, k, v))
