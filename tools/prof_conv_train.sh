#!/bin/bash
# Runs on the GPU box: kernel trace of the native training convolutions on every layer shape, summarised per
# (kernel, grid) -> gpurun_out/conv_train_kernels.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/convprof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o c --output-format csv -- python3 "$ROOT/tools/conv_train_bench.py" --native-only ${BENCH_ARGS:-} > "$OUT/run.log" 2>&1
cd "$ROOT" && python3 - > "$ROOT/gpurun_out/conv_train_kernels.txt" <<'PY'
import csv, collections, re
rows = collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/convprof/c_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "danra::" not in n:
        continue
    n = re.sub(r"\(.*$", "", re.sub(r"^(void )?danra::", "", n))
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, gx, gy, gz), d in rows.items():
    d.sort()
    print(f"{n:34s} grid=({gx},{gy},{gz})  x{len(d):3d}  median {d[len(d)//2]/1e3:7.1f} us")
PY
rm -f "$OUT"/c_kernel_trace.csv
cat "$ROOT/gpurun_out/conv_train_kernels.txt"; tail -12 "$OUT/run.log"
