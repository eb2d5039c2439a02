#!/usr/bin/env python3
"""gpurun_out/ (what scripts/profile_r06.sh left) -> the committed summaries under profiles/r06_*   (run in the build container)"""
import json
import os
import re
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        return True
    return False


# ---- kernel stats of every profiled command
for d, name in (("prof_r06_c3", "c3"), ("prof_r06_c3_one_stream", "c3_one_stream"), ("prof_r06_c2", "c2"), ("prof_r06_c4", "c4"), ("prof_r06_c3b", "c3b"),
                ("prof_r06_c5", "c5"), ("prof_r06_c3_train", "c3_train"), ("prof_r06_c3b_train", "c3b_train"), ("prof_r06_c5_train", "c5_train"),
                ("prof_r06_c4_train", "c4_train"), ("prof_r06_issue_c3_f64", "c3_f64")):
    cp(os.path.join(d, "kernel_stats.md"), "r06_%s_kernel_stats.md" % name)
# ---- bench lines
for src, dst in [("bench_r06_default.json", "r06_bench_c3.json"), ("bench_r06_one_stream.json", "r06_bench_c3_one_stream.json"), ("bench_r06_c5.json", "r06_bench_c5.json")] + \
                [("bench_r06_%s.json" % w, "r06_bench_%s.json" % w) for w in ("c2", "c4", "c3b")] + \
                [("bench_r06_%s_%s.json" % (w, k), "r06_bench_%s_%s.json" % (w, k)) for w in ("c2", "c3", "c3b", "c4", "c5") for k in ("train", "sample")]:
    cp(src, dst)
for f in ("scan_r06_f64.txt", "scan_r06_f32.txt"):
    cp(f, "r06_" + f.replace("scan_r06_", "scan_fixtures_"))

# ---- HBM traffic of the default command
def counter_means(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, counter, mean in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "jf::" in name:
            out.setdefault(name.replace("void ", ""), {})[counter] = mean
    return out


fdb, wdb = (os.path.join(G, "prof_r06_c3_traffic_%s" % c, "p.db") for c in ("FETCH_SIZE", "WRITE_SIZE"))
if os.path.exists(fdb) and os.path.exists(wdb):
    f, w = counter_means(fdb), counter_means(wdb)
    traffic = {"kernel_source_hash": bench.kernel_source_hash(), "kernels": {},
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python3 bench.py --pmc-child` (scripts/profile_r06.sh); "
                      "bytes = FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024 x %.3f" % bench.WRITE_CAL}
    for k in f:
        if k in w and "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w[k]:
            rd, wr = f[k]["FETCH_SIZE"] * 1024 * 2, w[k]["WRITE_SIZE"] * 1024 * bench.WRITE_CAL
            traffic["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "workload": "c3"}
    json.dump(traffic, open(os.path.join(P, "r06_traffic.json"), "w"), indent=1, sort_keys=True)
    print("traffic: %d kernels" % len(traffic["kernels"]))

# ---- the vector-issue profiles: instructions per row by class, issue cycles, clock
# (device kernel name pattern, entry[tag] of bench.py's kernel tables) per workload / precision
ENTRY = {
    "c3/f32": [(r"cond_gf_split_kernel<2, false, false, 2>", "jf_cond_gf_chain_split3_f32"), (r"cond_gf_split_kernel", "jf_cond_gf_chain_split2_f32"),
               (r"merged_side_kernel", "jf_merge_end"), (r"gfb_chain_inv_kernel<float", "jf_gf_chain_inv_f32[bcast]"),
               (r"cond_mchain_kernel<float, jf::FFam", "jf_cond_f_chain_inv_f32"), (r"combine_rows_kernel<float>", "jf_combine_rows_f32")],
    "c3b/f32": [(r"cond_gf_split_kernel<2, false, false, 2>", "jf_cond_gf_chain_split3_f32"), (r"cond_gf_split_kernel", "jf_cond_gf_chain_split2_f32"),
                (r"merged_side_kernel", "jf_merge_end"), (r"gfb_chain_inv_kernel<float", "jf_gf_chain_inv_f32[bcast]"),
                (r"cond_mchain_kernel<float, jf::FFam", "jf_cond_f_chain_inv_f32"), (r"combine_rows_kernel<float>", "jf_combine_rows_f32")],
    "c2/f32": [(r"gfb_chain_inv_kernel<float", "jf_gf_chain_inv_f32[bcast]"), (r"gfbg_chain_inv_kernel<float", "jf_gf_chain_inv_f32[bcast]")],
    "c4/f32": [(r"mchain_kernel<float, jf::RFam", "jf_r_chain_inv_f32[bcast]"), (r"mchain_kernel<float, jf::OFam", "jf_o_chain_inv_f32[per-sample]"),
               (r"mlp2_narrow_kernel", "jf_mlp2_f32"), (r"cond_mchain_kernel<float, jf::OFam", "jf_cond_o_chain_inv_f32"), (r"embed_kernel<float", "jf_sphere_to_embedding_f32")],
    "c3/f64": [(r"mlp2_i8_kernel", "jf_mlp2_i8_seg_f64"), (r"gf_chain_kernel<double, 4, false, false>", "jf_gf_chain_inv_f64[per-sample]"),
               (r"gfb_chain_inv_kernel<double, 4>", "jf_gf_chain_inv_f64[bcast]"), (r"mlp2_kernel<double", "jf_mlp2_f64"),
               (r"mchain_kernel<double, jf::FFam, false>", "jf_f_chain_inv_f64[per-sample]"), (r"conditioning_kernel<double>", "jf_conditioning_rows_f64"),
               (r"combine_rows_kernel<double>", "jf_combine_rows_f64")],
}
ROWS = {"c3": 1 << 20, "c3b": 1 << 20, "c2": 1 << 20, "c4": 1 << 20}
# measured issue cost of a wave64 vector instruction (scripts/probe/f64_rates.hip, one instruction per kernel, every SIMD busy): float32 plain
# (fma / add / mul / cndmask / int) 2.75 cycles, float32 transcendental (v_exp / v_log / v_rcp / v_sqrt) 8.3; float64: 4 (add / mul / fma: 4.3-4.7
# measured; the round-5 profile's convention is kept) and 16 for the transcendental class.  MFMA instructions have their own pipe and are taken out.
ISSUE = {"f32": {"plain": 2.75, "trans": 8.3}, "f64": {"plain": 4.0, "trans": 16.0}}
out = {"profiles": {}, "how": "rocprofv3 --pmc (two passes) + --kernel-trace of `python3 bench.py --pmc-child --pmc-dtype <f32|f64> [--workload ...]` (scripts/profile_r06.sh); "
                              "counts are wave instructions per launch summed over the chip, divided by the rows; valu_issue_cycles = plain x (SQ_INSTS_VALU - TRANS - SQ_INSTS_MFMA) "
                              "+ trans x TRANS; clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)"}
for key, entries in ENTRY.items():
    wl, dt = key.split("/")
    src = os.path.join(G, "prof_r06_issue_%s_%s" % (wl, dt))
    if not os.path.exists(os.path.join(src, "a.db")):
        continue
    cnt = {}
    for db in ("a.db", "b.db"):
        for k, v in counter_means(os.path.join(src, db)).items():
            cnt.setdefault(k, {}).update(v)
    dur = {}
    for line in open(os.path.join(src, "kernel_stats.md")):
        m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            dur[m.group(1)] = float(m.group(4))           # avg us
    B = ROWS[wl]
    prof = {"rows": B, "kernel_source_hash": bench.kernel_source_hash(), "issue_cycles": ISSUE[dt], "kernels": {}}
    tkey = "SQ_INSTS_VALU_TRANS_F64" if dt == "f64" else "SQ_INSTS_VALU_TRANS_F32"
    for kname, c in cnt.items():
        ent = next((e for pat, e in entries if pat in kname), None)
        if ent is None or "SQ_INSTS_VALU" not in c:
            continue
        d_us = next((v for k, v in dur.items() if kname[:60] in k or k[:60] in kname), None)
        trans, mfma = c.get(tkey, 0.0), c.get("SQ_INSTS_MFMA", 0.0)
        plain = c["SQ_INSTS_VALU"] - trans - mfma
        cycles = ISSUE[dt]["plain"] * plain + ISSUE[dt]["trans"] * trans
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clock = (gui / (d_us * 1e3)) if d_us else None
        rec = {"device_kernel": kname[:110], "avg_us_in_profile": d_us, "valu_insts_per_row": c["SQ_INSTS_VALU"] / B, "trans_insts_per_row": trans / B,
               "mfma_insts_per_row": mfma / B, "valu_issue_cycles_per_row": cycles / B, "clock_ghz": round(clock, 3) if clock else (1.6 if dt == "f64" else 2.1),
               "valu_busy_frac": (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui) if (gui and "SQ_ACTIVE_INST_VALU" in c) else None,
               "mfma_busy_frac": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui) if (gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c) else None,
               "mfma_valu_coexec_frac": (c["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024 / gui) if (gui and "SQ_VALU_MFMA_COEXEC_CYCLES" in c) else None,
               "wave_insts_per_row": {k: c[k] / B for k in sorted(c) if k.startswith("SQ_INSTS")}}
        if dt == "f64":
            f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
            rec["f64_share_of_valu_insts"] = (f64 + trans) / c["SQ_INSTS_VALU"]
        prof["kernels"].setdefault(ent, rec)
    if dt == "f64" and "jf_mlp2_i8_seg_f64" in prof["kernels"]:
        prof["kernels"]["jf_mlp2_i8_f64"] = prof["kernels"]["jf_mlp2_i8_seg_f64"]
    out["profiles"][key] = prof
    shutil.copy(os.path.join(src, "pmc.txt"), os.path.join(P, "r06_%s_%s_pmc.txt" % (wl, dt)))
    print(key, {k: (round(v["valu_issue_cycles_per_row"], 1), v["clock_ghz"], None if v["valu_busy_frac"] is None else round(v["valu_busy_frac"], 2)) for k, v in prof["kernels"].items()})
if out["profiles"]:
    json.dump(out, open(os.path.join(P, "r06_valu_issue.json"), "w"), indent=1, sort_keys=True)
print(sorted(f for f in os.listdir(P) if f.startswith("r06")))
