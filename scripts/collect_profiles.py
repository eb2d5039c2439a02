#!/usr/bin/env python3
"""Turn the rocprofv3 result databases that scripts/profile_step.sh left under gpurun_out/prof_<round>_<workload>/ into the committed
summaries under profiles/:

    python scripts/collect_profiles.py r02 c3 c5

    profiles/<round>_<workload>_kernel_stats.md   per-kernel calls / total / average of the --kernel-trace --stats pass
    profiles/<round>_<workload>_counters.json     mean of every collected counter per jf:: kernel (FETCH_SIZE / WRITE_SIZE in raw KB, SQ_*, GRBM_*)
    profiles/<round>_traffic.json                 HBM bytes per launch per kernel (FETCH_SIZE x 2 on gfx950, WRITE_SIZE x bench.WRITE_CAL) + the hash
                                                   of the kernel sources they were taken at -- what bench.py falls back to when it cannot run
                                                   its own --pmc child passes
"""
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counters(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
    for name, counter, n, mean in cur.execute(q):
        if "jf::" in name:
            out.setdefault(name.replace("void ", ""), {})[counter] = {"launches": n, "mean": mean}
    return out


def main():
    rnd, workloads = sys.argv[1], sys.argv[2:]
    traffic = {"kernel_source_hash": bench.kernel_source_hash(), "kernels": {},
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python3 bench.py --pmc-child [--workload ...]` "
                      "(scripts/profile_step.sh); bytes = FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024 x %.3f" % bench.WRITE_CAL}
    for w in workloads:
        src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (rnd, w))
        shutil.copy(os.path.join(src, "kernel_stats.md"), os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.md" % (rnd, w)))
        allc = {}
        for db in ("fetch", "write", "sq1", "sq2"):
            p = os.path.join(src, db + ".db")
            if os.path.exists(p):
                for k, v in counters(p).items():
                    allc.setdefault(k, {}).update(v)
        json.dump(allc, open(os.path.join(ROOT, "profiles", "%s_%s_counters.json" % (rnd, w)), "w"), indent=1, sort_keys=True)
        for k, v in allc.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                rd, wr = v["FETCH_SIZE"]["mean"] * 1024 * 2, v["WRITE_SIZE"]["mean"] * 1024 * bench.WRITE_CAL
                traffic["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "workload": w}
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % rnd), "w"), indent=1, sort_keys=True)
    print("kernel sources", traffic["kernel_source_hash"], "kernels", len(traffic["kernels"]))


if __name__ == "__main__":
    main()
