#!/bin/bash
# Runs on the GPU box: SQ counter passes over tune_igemm on one shape and configuration.
#   tools/pmc_igemm.sh <shape index> <config index> <tag>
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${1:-0}; CFG=${2:-4}; TAG=${3:-x}
OUT=$ROOT/gpurun_out/pmc_igemm_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DANRA_TUNE_HALF=1
BIN=$ROOT/diffusionmodelscustom_amd/csrc/tune_igemm
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$OUT/p1" -o a --output-format csv -- "$BIN" 64 "$SHAPE" "$CFG" 5 > "$OUT/p1.log" 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d "$OUT/p2" -o b --output-format csv -- "$BIN" 64 "$SHAPE" "$CFG" 5 > "$OUT/p2.log" 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS -d "$OUT/p3" -o c --output-format csv -- "$BIN" 64 "$SHAPE" "$CFG" 5 > "$OUT/p3.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_igemm" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(f"{k:32s} {v / n:16.1f} per launch ({n} launches)")
PY
