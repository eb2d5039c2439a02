#!/bin/bash
# gpurun_out/ (what scripts/profile_r03.sh left) -> the committed summaries under profiles/r03_*   (run in the build container)
set -e
cd "$(dirname "$0")/.."
python scripts/collect_profiles.py r03 c3 c5
cp gpurun_out/bench_r03_default.json profiles/r03_bench_c3.json
cp gpurun_out/bench_r03_c5.json profiles/r03_bench_c5.json
for wl in c3 c5; do
  { echo "# Sampling, $wl — \`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-pmc --workload $wl --direction sample\` (round 3)"; echo
    echo '`bench.py` line of the same command (HIP events):'; echo; echo '```'
    python - <<PY
import json
d = json.loads(open("gpurun_out/prof_r03_sample_$wl/bench.json").read().strip().split("\n")[-1])
print("ms_per_step %.3f  value %.4g %s  rows %d  dtype %s" % (d["ms_per_step"], d["value"], d["unit"], d["config"]["batch_per_gpu"], d["dtype"]))
print("parity", {k: v for k, v in d["parity"].items() if k != "note"})
for k, v in sorted(d["roofline"]["all_kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
    print("  %-52s %.4f ms" % (k, v))
PY
    echo '```'; echo; head -24 gpurun_out/prof_r03_sample_$wl/kernel_stats.md; } > profiles/r03_sample_$wl.md
  cp gpurun_out/prof_r03_sample_$wl/bench.json profiles/r03_bench_${wl}_sample.json
  cp gpurun_out/bench_r03_${wl}_train.json profiles/r03_bench_${wl}_train.json
done
{ echo "# Training step (forward + backward + Adam) — round 3"; echo
  echo '`bash scripts/profile_train.sh c3 262144 pmc` / `bash scripts/profile_train.sh c5 131072 pmc` on one MI355X (`scripts/bench_train.py`: weights = golden-fixture'
  echo 'state_dict, SURVEY 8d inputs, loss = -mean(log p), torch.optim.Adam).  Per section: HIP events around every C-ABI launch of one step, the'
  echo '`rocprofv3 --kernel-trace --stats` table of the same command (106 calls per kernel), and the SQ / HBM counters of the kernels that matter'
  echo '(`--pmc` passes of `--pmc-child`, means per launch; FETCH_SIZE / WRITE_SIZE in raw KB).'; echo
  echo '## C3 `pdf("e4+s2+e4","gggg+f+gggg")`, float32, 2^18 rows'; echo; echo '```'
  grep -v "Warn\|amdgpu.ids\|args.workload\|Consider" gpurun_out/prof_train_c3/bench_train.txt; echo '```'; echo
  head -28 gpurun_out/prof_train_c3/kernel_stats.md | tail -26; echo; echo '```'
  grep "cond_gf_split_bwd\|wgrad_split\|cond_gf_split_kernel" gpurun_out/prof_train_c3/pmc.txt; echo '```'; echo
  echo '## C5 conditional `pdf("e8+s2","gggg+v")`, AmortizableMLP rank 8, float64, 2^17 rows'; echo; echo '```'
  grep -v "Warn\|amdgpu.ids\|args.workload\|Consider" gpurun_out/prof_train_c5/bench_train.txt; echo '```'; echo
  head -24 gpurun_out/prof_train_c5/kernel_stats.md | tail -22; } > profiles/r03_train.md
{ echo "# Float64 C3 step — \`rocprofv3 --kernel-trace --stats -- python3 scripts/probe/stress.py c3_e4s2e4 f64 30\` (round 3, 2^20 rows)"; echo
  head -14 gpurun_out/prof_r03_c3_f64/kernel_stats.md; echo
  echo 'SQ / HBM counters per launch (separate `--pmc` passes; FETCH_SIZE / WRITE_SIZE in raw KB):'; echo; echo '```'
  cat gpurun_out/prof_r03_c3_f64/pmc.txt; echo '```'; } > profiles/r03_c3_f64.md
cp gpurun_out/bench_configs_r03.txt profiles/r03_bench_configs.txt
ls profiles | grep r03
