#!/usr/bin/env python3
"""gpurun_out/ (what scripts/profile_r04b.sh left) -> profiles/r04_train.md, profiles/r04_sample.md, profiles/r04_bench_c{3,5}_{train,sample}.json"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
DESC = {"c3": 'C3 `pdf("e4+s2+e4","gggg+f+gggg")`, float32', "c5": 'C5 `pdf("e8+s2","gggg+v")`, 16 conditioning inputs, low-rank MLP, float64'}
for mode, title in (("train", "Training step (forward + backward + one-launch Adam)"), ("sample", "Sampling step")):
    out = ["# %s -- round 4, second half" % title, "",
           "`bash scripts/profile_r04b.sh` on one MI355X: per configuration the `bench.py` line (HIP events around every C-ABI launch inside the timed",
           "region) and the `rocprofv3 --kernel-trace --stats` table of the same command.", ""]
    for wl in ("c3", "c5"):
        src = os.path.join(G, "bench_r04_%s_%s.json" % (wl, mode))
        shutil.copy(src, os.path.join(P, "r04_bench_%s_%s.json" % (wl, mode)))
        d = json.loads(open(src).read().strip().splitlines()[-1])
        out += ["## " + DESC[wl] + ", %d rows" % d["config"]["total_rows"], "", "```",
                "ms_per_step %.3f   value %.4g %s   dtype %s" % (d["ms_per_step"], d["value"], d["unit"], d["dtype"])]
        if d.get("optimizer"):
            out.append("optimizer   %s" % d["optimizer"])
        if d.get("step_issue"):
            out.append("step issue  %s   (eager: %s)" % (d["step_issue"], json.dumps(d.get("eager"))))
        if d.get("hip_graph_replay"):
            out.append("hip graph replay of the same step: %s" % json.dumps(d["hip_graph_replay"]))
        out.append("parity      %s" % json.dumps(d.get("parity")))
        for k, v in sorted(d["roofline"]["all_kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
            out.append("  %-52s %8.4f ms" % (k, v))
        out += ["```", ""]
        ks = open(os.path.join(G, "prof_r04_%s_%s" % (wl, mode), "kernel_stats.md")).read().splitlines()
        out += [l for l in ks if l.startswith("|")][:16] + [""]
    open(os.path.join(P, "r04_%s.md" % mode), "w").write("\n".join(out) + "\n")
    print("wrote", "profiles/r04_%s.md" % mode)
lc = os.path.join(G, "lowrank_check_r04.txt")
if os.path.exists(lc):
    txt = [l for l in open(lc).read().splitlines() if "amdgpu.ids" not in l]
    open(os.path.join(P, "r04_lowrank_ab.txt"), "w").write(
        "# C5 training step at 2^17 rows, A/B on one box: scripts/probe/lowrank_check.py 131072\n# flag True = chain on the low-rank last stage (DESIGN 3.5d), "
        "flag False = the (B, 1224)-block sequence of round 3; HIP events per C-ABI call, torch Adam excluded\n" + "\n".join(txt) + "\n")
