#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# and a per-dispatch timeline of one UNet evaluation.  Everything lands in gpurun_out/refresh/.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python3 bench.py > "$OUT/bench.log" 2>&1
grep -o '{"metric.*' "$OUT/bench.log" > "$OUT/bench_c3.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o r --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-strict-fp32 > "$OUT/bench_prof.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_prof.log" > "$OUT/bench_c3_under_rocprof.json"
cp "$OUT"/prof/r_kernel_stats.csv "$OUT/bench_c3_kernel_stats.csv"
rm -f "$OUT"/prof/r_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/prof_t" -o r1 --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 200 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/prof_t.log" 2>&1
cd "$ROOT" && python3 tools/prof_forward.py "$OUT/prof_t/r1_kernel_trace.csv" > "$OUT/forward_timeline_T200.txt"
rm -f "$OUT"/prof_t/r1_kernel_trace.csv
# the other two benchmark configurations (bench lines only)
cd "$ROOT"
timeout -k 10 400 python3 bench.py --workload c2 > "$OUT/bench_c2.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_c2.log" > "$OUT/bench_c2.json"
timeout -k 10 600 python3 bench.py --workload c4 > "$OUT/bench_c4.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_c4.log" > "$OUT/bench_c4.json"
cat "$OUT/bench_c3.json"
