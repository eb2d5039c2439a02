"""rocprofv3 PMC child passes (HBM traffic), the committed profiles they fall back to, the computed vector-issue roof and the per-kernel
algorithmic accounting of SURVEY 8d"""
import glob
import hashlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

from .workloads import ROOT, BENCH_PY, WORKLOADS

# ---------------------------------------------------------------------------------------------- HBM traffic (rocprofv3 PMC)
PROFILE_TRAFFIC = next((p for p in (os.path.join(ROOT, "profiles", "r06_traffic.json"), os.path.join(ROOT, "profiles", "r05_traffic.json")) if os.path.exists(p)),
                       os.path.join(ROOT, "profiles", "r06_traffic.json"))
WRITE_CAL = 0.965     # WRITE_SIZE calibration on scripts/probe/wstore (16-byte lane-per-row tile stores); FETCH_SIZE x 2 on gfx950 (guide)


def kernel_source_hash():
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "jammy_flows_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "jammy_hip.h")]):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_pass(counter, workload, rows, fuse):
    """one rocprofv3 --pmc pass of a few steps of this workload in a child process -> {kernel name: mean raw counter value}.
    The child is this script (`--pmc-child`): python itself is what follows `--`, nothing re-execs after the GPU is initialised."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="jf_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", counter, "-d", tmp, "--", sys.executable, BENCH_PY, "--pmc-child", "--workload", workload,
               "--batch", str(rows)] + ([] if fuse else ["--no-fuse"])
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None
        cur = sqlite3.connect(dbs[0]).cursor()
        q = "select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name"
        return {n: v for n, v in cur.execute(q, (counter,)) if "jf::" in n}
    except Exception:                                # noqa: BLE001 -- profiling is optional; the fallback is the committed profile
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(workload, rows, fuse):
    """{kernel name: {"read_bytes", "write_bytes", "hbm_bytes_per_launch"}} from two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share one)."""
    f = pmc_pass("FETCH_SIZE", workload, rows, fuse)
    if not f:
        return None
    w = pmc_pass("WRITE_SIZE", workload, rows, fuse)
    if not w:
        return None
    out = {}
    for k in f:
        if k in w:
            rd, wr = f[k] * 1024 * 2, w[k] * 1024 * WRITE_CAL
            out[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    return {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child passes of this run (FETCH_SIZE x 2, WRITE_SIZE x %.3f)" % WRITE_CAL,
            "kernels": out}


PROFILE_F64_ISSUE = os.path.join(ROOT, "profiles", "r05_f64_issue.json")
PROFILE_VALU_ISSUE = os.path.join(ROOT, "profiles", "r06_valu_issue.json")
N_SIMDS = 1024                           # 4 per CU x 256 CUs
SIDE_TABLE_STEPS = 3                     # eager steps behind the float64 leg's per-kernel table


def valu_issue_roofline(workload, dtype, rows, table):
    """a step against the VECTOR-ISSUE roof, computed: per kernel, the vector instructions per row by class (rocprofv3 --pmc SQ_INSTS_VALU*, a property
    of the code: profiles/r06_valu_issue.json, scripts/profile_r06.sh / collect_r06.py) x the measured issue cost of a wave64 instruction of that class
    (scripts/probe/f64_rates.hip: float32 plain 2.75 cycles, float32 transcendental 8.3; float64 4 / 16) = issue cycles per row; with THIS run's kernel
    times: achieved = issue cycles per second, peak = 1024 SIMDs x the clock measured during the kernel.  `table`: {(entry, tag): {"mean_ms": ...}}."""
    try:
        prof = json.load(open(PROFILE_VALU_ISSUE))["profiles"].get("%s/%s" % (workload, dtype))
    except (OSError, ValueError, KeyError):
        return None
    if not prof:
        return None
    out = {"bound": "vector issue", "unit": "T issue cycles/s", "kernels": {},
           "source": "profiles/r06_valu_issue.json (rocprofv3 --pmc: instructions per row by class, clock during the kernel) x this run's kernel times",
           "issue_cycles_per_wave_instruction": prof.get("issue_cycles"), "kernel_source_hash_match": prof.get("kernel_source_hash") == kernel_source_hash()}
    tot_c, tot_s, peak_w = 0.0, 0.0, 0.0
    for (name, tag), v in table.items():
        base = name.replace("_inv_total_", "_inv_").replace("_inv_sum_", "_inv_")       # the last block's entry points (..._inv_total / _inv_sum): the same device kernels
        k = prof["kernels"].get("%s[%s]" % (name, tag)) or prof["kernels"].get(name) or prof["kernels"].get("%s[%s]" % (base, tag)) or prof["kernels"].get(base)
        if not k:
            continue
        cyc = k["valu_issue_cycles_per_row"] * rows
        sec = v["mean_ms"] * 1e-3
        peak = N_SIMDS * k["clock_ghz"] * 1e9
        out["kernels"]["%s[%s]" % (name, tag)] = {"ms": round(v["mean_ms"], 4), "frac": cyc / sec / peak, "clock_ghz": k["clock_ghz"],
                                                 "valu_insts_per_row": k.get("valu_insts_per_row"), "trans_insts_per_row": k.get("trans_insts_per_row"),
                                                 "mfma_insts_per_row": k.get("mfma_insts_per_row"), "valu_busy_frac_in_profile": k.get("valu_busy_frac")}
        tot_c += cyc
        tot_s += sec
        peak_w += peak * sec
    if tot_s <= 0:
        return None
    out.update({"achieved": tot_c / tot_s / 1e12, "peak": peak_w / tot_s / 1e12, "frac": tot_c / peak_w, "peak_at_2.4GHz": N_SIMDS * 2.4e9 / 1e12})
    return out


def float64_issue_roofline(workload, rows, side_table):
    """the float64 step against the bound its counters show: VECTOR ISSUE (DESIGN "float64").  The committed profile holds, per kernel of the step,
    the vector instructions per row by class (rocprofv3 --pmc SQ_INSTS_VALU*: a property of the code), their issue cycles (4 per wave64
    instruction, float64 add / mul / fma included; 16 for the transcendental class) and the chip's clock during the kernel (GRBM_GUI_ACTIVE / 8 /
    duration: ~1.6 GHz under float64 load, not the 2.4 GHz of the headline peaks).  With THIS run's kernel times: achieved = vector issue cycles
    per second, peak = 1024 SIMDs x the measured clock."""
    try:
        prof = json.load(open(PROFILE_F64_ISSUE))
    except (OSError, ValueError):
        return None
    if prof.get("workload") != workload:
        return None
    out = {"bound": "vector issue (float64 arithmetic: %s of the instructions)", "unit": "T issue cycles/s",
           "source": "profiles/r05_f64_issue.json (rocprofv3 --pmc, per-row instruction counts and clocks) x this run's kernel times",
           "kernel_source_hash_match": prof.get("kernel_source_hash") == kernel_source_hash(), "kernels": {}}
    tot_c, tot_s, peak_w, f64w = 0.0, 0.0, 0.0, 0.0
    for (name, tag), v in side_table.items():
        k = prof["kernels"].get("%s[%s]" % (name, tag)) or prof["kernels"].get(name)
        if not k:
            continue
        cyc = k["valu_issue_cycles_per_row"] * rows
        sec = v["mean_ms"] * 1e-3
        peak = N_SIMDS * k["clock_ghz"] * 1e9
        out["kernels"]["%s[%s]" % (name, tag)] = {"ms": round(v["mean_ms"], 4), "frac": cyc / sec / peak, "clock_ghz": k["clock_ghz"],
                                                 "valu_busy_frac_in_profile": k.get("valu_busy_frac"), "f64_share_of_valu_insts": k.get("f64_share_of_valu_insts"),
                                                 "valu_insts_per_row": k["wave_insts_per_row"].get("SQ_INSTS_VALU")}
        tot_c += cyc
        tot_s += sec
        peak_w += peak * sec
        f64w += (k.get("f64_share_of_valu_insts") or 0.0) * cyc
    if tot_s <= 0:
        return None
    out["bound"] = out["bound"] % ("%.0f %%" % (100.0 * f64w / tot_c))
    out.update({"achieved": tot_c / tot_s / 1e12, "peak": peak_w / tot_s / 1e12, "frac": tot_c / peak_w, "peak_at_2.4GHz": N_SIMDS * 2.4e9 / 1e12,
                "note": "vector-issue cycles of the float64 step's kernels over (1024 SIMDs x the clock measured during each kernel).  The flow kernels sit "
                        "at ~0.8 of this roof; the 40 % HBM bar on SURVEY 8d bytes would need the step in 3.0 ms, i.e. fewer instructions, not more bandwidth"})
    return out


def committed_traffic():
    try:
        t = json.load(open(PROFILE_TRAFFIC))
    except (OSError, ValueError):
        return None
    if t.get("kernel_source_hash") != kernel_source_hash():
        return {"stale": True, "source": "profiles/%s (taken at kernel sources %s, now %s)" % (os.path.basename(PROFILE_TRAFFIC), t.get("kernel_source_hash"), kernel_source_hash())}
    t["source"] = "profiles/%s (committed rocprofv3 --pmc passes of this command; kernel sources unchanged since)" % os.path.basename(PROFILE_TRAFFIC)
    return t


# device-kernel name (as rocprofv3 reports it) of a (C entry point, tag) pair of the host-side timer
KERNEL_OF = {"jf_cond_f_chain_inv_f32": "cond_mchain_kernel<float, jf::FFam", "jf_cond_f_chain_inv_f64": "cond_mchain_kernel<double, jf::FFam",
             "jf_conditioning_rows_f32": "conditioning_kernel<float", "jf_conditioning_rows_f64": "conditioning_kernel<double",
             "jf_v_chain_inv_f64": "mchain_kernel<double, jf::VFam", "jf_amlp2_f64": "amlp2_mfma_kernel",
             "jf_cond_gf_chain_inv_split_f32": "cond_gf_split_kernel",
             "jf_cond_gf_chain_split2_f32": "cond_gf_split_kernel", "jf_cond_gf_chain_split3_f32": "cond_gf_split_kernel",
             "jf_mlp2_i8_f64": "mlp2_i8_kernel", "jf_mlp2_i8_seg_f64": "mlp2_i8_kernel",
             "jf_r_chain_inv_f32": "mchain_kernel<float, jf::RFam", "jf_o_chain_inv_f32": "mchain_kernel<float, jf::OFam",
             "jf_cond_gf_chain_inv_f32": "cond_gf_chain_kernel<float",
             "jf_cond_gf_chain_inv_f64": "cond_gf_chain_kernel<double", "jf_mlp2_f32": "mlp2_kernel<float", "jf_mlp2_f64": "mlp2_kernel<double",
             "jf_gf_chain_inv_f32": "gf_chain_kernel<float", "jf_gf_chain_inv_f64": "gf_chain_kernel<double",
             "jf_gf_chain_inv_total_f32": "gf_chain_kernel<float", "jf_gf_chain_inv_total_f64": "gf_chain_kernel<double",
             "jf_amlp_gf_chain_inv_f64": "amlp_gf_mfma_kernel", "jf_merge_end": "merged_side_kernel"}


def traffic_of(traffic, kname, ktag):
    if not traffic or traffic.get("stale") or "kernels" not in traffic:
        return None
    key = KERNEL_OF.get(kname)
    if key is None:
        return None
    cands = [(n, v) for n, v in traffic["kernels"].items() if key in n]
    if kname.startswith("jf_gf_chain_inv"):          # broadcast: the lane = row kernel gfb_chain_inv_kernel<T, D> (classic stretch) or
        if ktag == "bcast":                          # gf_chain_kernel<..., true, false>; per-sample: gf_chain_kernel<..., false, false>
            rows_kernel = [(n, v) for n, v in traffic["kernels"].items() if key.replace("gf_chain_kernel", "gfb_chain_inv_kernel") in n]
            cands = rows_kernel or [(n, v) for n, v in cands if ", true, false>" in n]
        else:
            cands = [(n, v) for n, v in cands if ", false, false>" in n]
    if kname.startswith("jf_mlp2") and len(cands) > 1:   # narrow-output variant (TN = 1) for N <= 16, wide otherwise
        narrow = int(ktag.split("_")[-1][1:]) <= 32
        cands = [(n, v) for n, v in cands if (", 1, true" in n) == narrow] or cands
    return cands[0][1] if len(cands) == 1 else None


# ---------------------------------------------------------------------------------------------- algorithmic accounting
def kernel_accounting(kname, ktag, s):
    """(algorithmic HBM bytes per row, MFMA flops per row, fused?) of one timed kernel; s = bytes per scalar."""
    if (kname.startswith("jf_cond_gf_chain_inv_split") or kname.startswith("jf_cond_gf_chain_split2")
            or kname.startswith("jf_cond_gf_chain_split3")):
        K1, H, L, D = (int(t[1:]) for t in ktag.split("_")[:4])
        N = L * (3 * 10 * D + D * D) + D                 # default g rows: 3 K D + D^2 (+ D offsets on the last layer)
        return s * (K1 + N) + s * (D + 1 + N + D + 1), 2 * (K1 * H + H * N), True
    if kname.startswith("jf_cond_gf_chain") or kname.startswith("jf_amlp_gf_chain"):
        K1, H, N, D = (int(t[1:]) for t in ktag.split("_")[:4])
        # fused launch (MLP + g layers): SURVEY 8d "materialised" accounting = MLP (reads inputs, writes block) + flow (reads block);
        # the block itself never reaches HBM, so the real traffic is only s (K1 + 2 D + 2) bytes per row
        return s * (K1 + N) + s * (D + 1 + N + D + 1), 2 * (K1 * H + H * N), True
    if kname.startswith("jf_linear"):
        K, N = int(ktag.split("_")[0][1:]), int(ktag.split("_")[1][1:])
        return s * (K + N), 2 * K * N, False
    if kname.startswith("jf_mlp2"):
        K1, H, N = (int(t[1:]) for t in ktag.split("_"))
        return s * (K1 + N), 2 * (K1 * H + H * N), False
    if kname.startswith("jf_gf_chain_inv"):
        if ktag == "bcast":
            return s * 10, 0, False
        return None, 0, False                            # per-sample: depends on the block (filled in by the caller)
    return 0, 0, False
