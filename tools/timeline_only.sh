#!/bin/bash
# Runs on the GPU box (via gpurun): per-dispatch timeline of one UNet evaluation -> gpurun_out/timeline.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/prof_t" -o r1 --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 200 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 ${BENCH_ARGS:-} > "$OUT/prof_t.log" 2>&1
cd "$ROOT" && python3 tools/prof_forward.py "$OUT/prof_t/r1_kernel_trace.csv" > "$ROOT/gpurun_out/timeline.txt"
rm -rf "$OUT"
tail -${TAIL:-24} "$ROOT/gpurun_out/timeline.txt"
