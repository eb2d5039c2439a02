#!/usr/bin/env python3
"""gpurun_out/ (what scripts/profile_r05.sh left) -> the committed summaries under profiles/r05_*   (run in the build container)"""
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "collect_profiles.py"), "r05", "c3", "c5"])
for wl in ("c2", "c4"):
    shutil.copy(os.path.join(G, "prof_r05_%s" % wl, "kernel_stats.md"), os.path.join(P, "r05_%s_kernel_stats.md" % wl))
if os.path.exists(os.path.join(G, "prof_r05_c3_one_stream", "kernel_stats.md")):
    shutil.copy(os.path.join(G, "prof_r05_c3_one_stream", "kernel_stats.md"), os.path.join(P, "r05_c3_one_stream_kernel_stats.md"))
for wl in ("c3", "c5"):
    f = os.path.join(G, "prof_r05_%s_train" % wl, "kernel_stats.md")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "r05_%s_train_kernel_stats.md" % wl))
for src, dst in (("bench_r05_default.json", "r05_bench_c3.json"), ("bench_r05_one_stream.json", "r05_bench_c3_one_stream.json"),
                 ("bench_r05_c5.json", "r05_bench_c5.json"), ("bench_r05_c2.json", "r05_bench_c2.json"), ("bench_r05_c4.json", "r05_bench_c4.json"),
                 ("bench_r05_c3_train.json", "r05_bench_c3_train.json"), ("bench_r05_c5_train.json", "r05_bench_c5_train.json"),
                 ("bench_r05_c3_sample.json", "r05_bench_c3_sample.json"), ("bench_r05_c5_sample.json", "r05_bench_c5_sample.json"),
                 ("bench_r05_c2_train.json", "r05_bench_c2_train.json"), ("bench_r05_c4_train.json", "r05_bench_c4_train.json"),
                 ("bench_r05_c2_sample.json", "r05_bench_c2_sample.json"), ("bench_r05_c4_sample.json", "r05_bench_c4_sample.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))

# ---- the float64 step of C3: float64 vector instructions per row and class, the clock during each kernel
src = os.path.join(G, "prof_r05_c3_f64")
ENTRY = [(r"mlp2_i8_kernel", "jf_mlp2_i8_seg_f64"), (r"gf_chain_kernel<double, 4, false, false>", "jf_gf_chain_inv_f64[per-sample]"),
         (r"gfb_chain_inv_kernel<double, 4>", "jf_gf_chain_inv_f64[bcast]"), (r"mlp2_kernel<double", "jf_mlp2_f64"),
         (r"mchain_kernel<double, jf::FFam, false>", "jf_f_chain_inv_f64[per-sample]"), (r"conditioning_kernel<double>", "jf_conditioning_rows_f64"),
         (r"combine_rows_kernel<double>", "jf_combine_rows_f64")]
if os.path.exists(os.path.join(src, "a.db")):
    cnt = {}
    for db in ("a.db", "b.db"):
        cur = sqlite3.connect(os.path.join(src, db)).cursor()
        for name, counter, mean in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            if "jf::" in name:
                cnt.setdefault(name.replace("void ", ""), {})[counter] = mean
    dur = {}
    for line in open(os.path.join(src, "kernel_stats.md")):
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            dur[m.group(1)] = float(m.group(4))           # avg us
    B = 1 << 20
    out = {"workload": "c3", "rows": B, "kernel_source_hash": bench.kernel_source_hash(), "kernels": {},
           "how": "rocprofv3 --pmc (two passes) + --kernel-trace of `python3 bench.py --pmc-child --pmc-dtype f64` (scripts/profile_r05.sh); counts are wave "
                  "instructions per launch summed over the chip, divided by the rows; valu_issue_cycles = 4 x (SQ_INSTS_VALU - TRANS_F64) + 16 x TRANS_F64; "
                  "clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)"}
    for kname, c in cnt.items():
        key = next((e for pat, e in ENTRY if pat in kname), None)
        if key is None or "SQ_INSTS_VALU_FMA_F64" not in c:
            continue
        d_us = next((v for k, v in dur.items() if kname[:60] in k or k[:60] in kname), None)
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
        # issue cycles of a wave64 vector instruction on a 16-lane SIMD: 4, transcendental class (v_rcp_f64, v_rsq_f64, ...) 16 -- float64 add / mul /
        # fma included (78.6 TFLOP/s of float64 vector = 16 lanes x 2 flop x 4 SIMDs x 256 CUs x 2.4 GHz); counters are summed over the chip
        cycles = 4 * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_VALU_TRANS_F64"]) + 16 * c["SQ_INSTS_VALU_TRANS_F64"]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                                          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        clock = (gui / (d_us * 1e3)) if d_us else None                                     # cycles / ns = GHz
        out["kernels"][key] = {"device_kernel": kname[:100], "avg_us_in_profile": d_us,
                               "wave_insts_per_row": {k: c[k] / B for k in sorted(c) if k.startswith("SQ_INSTS")},
                               "valu_issue_cycles_per_row": cycles / B, "f64_share_of_valu_insts": (f64 + c["SQ_INSTS_VALU_TRANS_F64"]) / c["SQ_INSTS_VALU"],
                               "clock_ghz": round(clock, 3) if clock else 1.6,
                               "valu_busy_frac": (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui) if gui else None}
    if "jf_mlp2_i8_seg_f64" in out["kernels"]:
        out["kernels"]["jf_mlp2_i8_f64"] = out["kernels"]["jf_mlp2_i8_seg_f64"]        # (the same device kernel behind both entry points)
    json.dump(out, open(os.path.join(P, "r05_f64_issue.json"), "w"), indent=1, sort_keys=True)
    shutil.copy(os.path.join(src, "pmc.txt"), os.path.join(P, "r05_c3_f64_pmc.txt"))
    shutil.copy(os.path.join(src, "kernel_stats.md"), os.path.join(P, "r05_c3_f64_kernel_stats.md"))
    print("float64 issue profile:", {k: (round(v["valu_issue_cycles_per_row"], 1), v["clock_ghz"], round(v["valu_busy_frac"] or 0, 2)) for k, v in out["kernels"].items()})
print(sorted(f for f in os.listdir(P) if f.startswith("r05")))
