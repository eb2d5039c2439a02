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



def pmc_table(db, counter):
    """{(kernel_name, grid): mean raw counter value} for the jf:: kernels of one --pmc pass"""
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, avg(value) from counters_collection where counter_name=? group by kernel_name, grid_size")
    return {(short(n), g): v for n, g, v in cur.execute(q, (counter,)) if "jf::" in n}


if __name__ == "__main__":
    print("## kernel trace (rocprofv3 --kernel-trace --stats)\n")
    kernel_stats(sys.argv[1])
    if "--pmc" in sys.argv:
        i = sys.argv.index("--pmc")
        print("\n## FETCH_SIZE (separate --pmc pass; gfx950: multiply by 2 for wide coalesced reads, MI355X_MICROARCH.md HBM section)\n")
        pmc(sys.argv[i + 1], "FETCH_SIZE")
        print("\n## WRITE_SIZE (separate --pmc pass; calibrated on scripts/probe/wstore, see the calibration section)\n")
        pmc(sys.argv[i + 2], "WRITE_SIZE")
    if "--calib" in sys.argv:       # WRITE_SIZE pass over scripts/probe/wstore: every kernel writes 1048576 x 548 x 4 = 2298478592 bytes
        j = sys.argv.index("--calib")
        cur = sqlite3.connect(sys.argv[j + 1]).cursor()
        print("\n## WRITE_SIZE calibration (scripts/probe/wstore: each launch writes exactly 2 298 478 592 B)\n")
        print("| kernel | launches | mean WRITE_SIZE (raw, KB) | bytes written / (raw x 1024) |")
        print("|---|---|---|---|")
        for n, c, v in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='WRITE_SIZE' "
                                   "group by kernel_name order by kernel_name"):
            print("| `%s` | %d | %.1f | %.3f |" % (short(n), c, v, 2298478592.0 / (v * 1024.0) if v else float("nan")))
    if "--json" in sys.argv:        # per-kernel HBM traffic per launch for bench.py's roofline.traffic
        import json
        k = sys.argv.index("--json")
        i = sys.argv.index("--pmc")
        fetch, write = pmc_table(sys.argv[i + 1], "FETCH_SIZE"), pmc_table(sys.argv[i + 2], "WRITE_SIZE")
        out = {}
        for key in sorted(set(fetch) | set(write)):
            out["%s @grid %d" % key] = {"FETCH_SIZE_raw_KB": fetch.get(key), "WRITE_SIZE_raw_KB": write.get(key)}
        json.dump(out, open(sys.argv[k + 1], "w"), indent=1)
