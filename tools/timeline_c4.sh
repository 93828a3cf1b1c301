#!/bin/bash
# Runs on the GPU box: per-dispatch timeline of one UNet evaluation of the 128x128 configuration (batch 32).
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/prof_c4" -o r1 --output-format csv -- python3 "$ROOT/bench.py" --workload c4 --steps 1 --warmup 0 --timesteps 40 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/prof_c4.log" 2>&1
cd "$ROOT" && python3 tools/prof_forward.py "$OUT/prof_c4/r1_kernel_trace.csv" > "$OUT/forward_timeline_c4.txt"
rm -f "$OUT"/prof_c4/r1_kernel_trace.csv
tail -40 "$OUT/forward_timeline_c4.txt"
