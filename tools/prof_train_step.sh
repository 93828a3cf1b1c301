#!/bin/bash
# Runs on the GPU box: the kernels of one replay of the graphed training step -> gpurun_out/train_step_kernels.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trainprof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$ROOT/tools/prof_train_step.py" ${STEP_ARGS:---size 64 --batch 32} > "$OUT/run.log" 2>&1
cd "$ROOT" && python3 tools/prof_train_step.py --trace "$OUT/t_kernel_trace.csv" --top 60 --width ${WIDTH:-240} ${SEQ:+--sequence} > "$ROOT/gpurun_out/train_step_kernels.txt"
rm -f "$OUT"/t_kernel_trace.csv
cat "$ROOT/gpurun_out/train_step_kernels.txt"
