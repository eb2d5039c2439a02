#!/bin/bash
# gpurun_out/ (what scripts/profile_r04.sh left) -> the committed summaries under profiles/r04_*   (run in the build container)
set -e
cd "$(dirname "$0")/.."
python scripts/collect_profiles.py r04 c3 c5
for wl in c2 c4; do cp gpurun_out/prof_r04_$wl/kernel_stats.md profiles/r04_${wl}_kernel_stats.md; done
cp gpurun_out/bench_r04_default.json profiles/r04_bench_c3.json
for wl in c5 c2 c4; do cp gpurun_out/bench_r04_$wl.json profiles/r04_bench_$wl.json; done
for wl in c3 c5; do
  cp gpurun_out/bench_r04_${wl}_train.json profiles/r04_bench_${wl}_train.json
  cp gpurun_out/bench_r04_${wl}_sample.json profiles/r04_bench_${wl}_sample.json
done
{ echo "# Step time against the batch size, one MI355X (round 4) -- \`python3 scripts/rows_sweep.py c3_e4s2e4 f32 12 20\` / \`... c1_e2_gg f64 12 12\`"; echo
  echo 'eager = pdf.forward (one ctypes call per launch), plan = pdf.planned_forward (jf_plan_launch: one call per step), graph = HIP-graph replay;'
  echo 'kernels_ms = per-kernel HIP events of the plan replays.  Strong scaling over 8 GPUs needs t(2^17) <= t(2^20) / (8 x 0.85).'; echo; echo '```'
  grep -v "amdgpu.ids" gpurun_out/rows_sweep_r04_c3.txt; grep -v "amdgpu.ids" gpurun_out/rows_sweep_r04_c1.txt; echo '```'; } > profiles/r04_rows_sweep.md
ls profiles | grep r04
