#!/bin/bash
# Runs on the GPU box: HBM-side traffic of every kernel of a short sampling run, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/pmc/{fetch,write}/..._counter_collection.csv
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 12 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 12 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/write.log" 2>&1
ls -la "$OUT"/fetch "$OUT"/write
head -3 "$OUT"/fetch/*counter_collection.csv
