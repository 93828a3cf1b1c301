#!/bin/bash
# Runs on the GPU box: two SQ counter passes over a 3-step sampling run.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/p1" -o a --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 4 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/p1.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$OUT/p2" -o b --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --timesteps 4 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > "$OUT/p2.log" 2>&1
ls "$OUT"/p1 "$OUT"/p2
