#!/usr/bin/env python3
"""Turn rocprofv3 result databases (rocpd sqlite, the default output of this ROCm 7.2 image) into the small text summaries kept
under profiles/:   python scripts/rocprof_summary.py <stats.db> [--pmc fetch.db write.db] > profiles/rNN_....md"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    for junk in ("at::native::(anonymous namespace)::", "at::native::"):
        name = name.replace(junk, "")
    return name[:110]


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---|---|---|---|")
    for n, c, t, a, p in rows:
        print("| `%s` | %d | %.1f | %.1f | %.2f |" % (short(n), c, t, a, p))


def pmc(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name=? "
         "group by kernel_name, grid_size order by avg(value) desc")
    print("| kernel | grid | launches | mean %s (raw counter, KB) |" % counter)
    print("|---|---|---|---|")
    for n, g, c, v in cur.execute(q, (counter,)):
        if n.startswith("void jf::"):
            print("| `%s` | %d | %d | %.1f |" % (short(n), g, c, v))


if __name__ == "__main__":
    print("## kernel trace (rocprofv3 --kernel-trace --stats)\n")
    kernel_stats(sys.argv[1])
    if "--pmc" in sys.argv:
        i = sys.argv.index("--pmc")
        print("\n## FETCH_SIZE (separate --pmc pass; gfx950: multiply by 2 for wide coalesced reads, MI355X_MICROARCH.md HBM section)\n")
        pmc(sys.argv[i + 1], "FETCH_SIZE")
        print("\n## WRITE_SIZE (separate --pmc pass; uncalibrated on gfx950)\n")
        pmc(sys.argv[i + 2], "WRITE_SIZE")
