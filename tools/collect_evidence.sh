#!/bin/bash
# Runs HERE after tools/final_evidence.sh has run on the GPU box (both stages): copies what profiles/ keeps for the
# current library build out of gpurun_out/ and regenerates the two counter summaries.  ROUND=NN picks the file prefix.
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=r${ROUND:-03}
G=$ROOT/gpurun_out
P=$ROOT/profiles
python3 "$ROOT/tools/pmc_summary.py" "$G/pmc" "$P/${R}_family_traffic.json" > "$P/${R}_pmc_traffic.txt"
python3 "$ROOT/tools/pmc_sq_summary.py" "$G/pmc_sq" > "$P/${R}_pmc_sq.txt"
cp "$G/refresh/bench_c3.json" "$P/${R}_bench_c3.json"
cp "$G/refresh/bench_c2.json" "$P/${R}_bench_c2.json"
cp "$G/refresh/bench_c4.json" "$P/${R}_bench_c4.json"
cp "$G/refresh/bench_c3_kernel_stats.csv" "$P/${R}_bench_c3_kernel_stats.csv"
cp "$G/refresh/bench_c3_under_rocprof.json" "$P/${R}_bench_c3_under_rocprof.json"
cp "$G/refresh/forward_timeline_T200.txt" "$P/${R}_forward_timeline_T200.txt"
for w in clean clean_cfg clean_h1 rehearsal_2rank rehearsal_2rank_c4; do cp "$G/final/bench_$w.json" "$P/${R}_bench_$w.json"; done
cp "$G/final/gpu_tests_full.txt" "$P/${R}_gpu_tests_full.txt"
sha256sum "$ROOT/diffusionmodelscustom_amd/libdanra_hip.so" | cut -c1-16
grep -o '"library_sha256": "[0-9a-f]\{16\}' "$P/${R}_family_traffic.json"
# (the training-step lines: tools/train_step_bench.py and tools/prof_train_step.sh, run separately)
for f in train_step_64.json train_step_128.json; do [ -f "$G/final/$f" ] && cp "$G/final/$f" "$P/${R}_$f"; done
[ -f "$G/train_step_kernels.txt" ] && cp "$G/train_step_kernels.txt" "$P/${R}_train_step_64_kernels.txt"
