#!/bin/bash
# Runs on the GPU box: kernel trace of tools/edm_bench.py, summarised per (kernel, grid) -> gpurun_out/edm_kernels.txt
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/edmprof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o e --output-format csv -- python3 "$ROOT/tools/edm_bench.py" --iters 20 ${BENCH_ARGS:-} > "$OUT/run.log" 2>&1
cd "$ROOT" && python3 - > "$ROOT/gpurun_out/edm_kernels.txt" <<'PY'
import csv, collections, re
rows = collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/edmprof/e_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_edm" not in n:
        continue
    n = re.sub(r"\(.*$", "", re.sub(r"^(void )?danra::(\(anonymous namespace\)::)?", "", n))
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, gx, gy, wg), d in rows.items():
    d.sort()
    print(f"{n:36s} grid=({gx},{gy}) wg={wg:>4s}  x{len(d):4d}  median {d[len(d)//2]/1e3:7.2f} us")
PY
rm -f "$OUT"/e_kernel_trace.csv
cat "$ROOT/gpurun_out/edm_kernels.txt"
