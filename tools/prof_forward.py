"""Print the per-dispatch timeline of the last complete UNet forward found in a rocprofv3
--kernel-trace CSV, and per-kernel totals for that forward.

    python tools/prof_forward.py gpurun_out/prof_r1/r1_kernel_trace.csv [--brief]
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    brief = "--brief" in sys.argv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # first kernel of a UNet evaluation: the per-step gather inside the sampler, else the time-projection kernel
    marks = [i for i, r in enumerate(rows) if "k_tbias_gather" in r["Kernel_Name"]]
    if len(marks) < 2:  # the sampler's update kernel gathers for the next step: a step starts behind an update kernel
        marks = [i + 1 for i, r in enumerate(rows) if "k_ddpm_update_state" in r["Kernel_Name"] and i + 1 < len(rows)]
    if len(marks) < 2:
        marks = [i for i, r in enumerate(rows) if "k_time_mlp" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    tot = 0.0
    per = defaultdict(lambda: [0, 0.0])
    for r in rows[a:b]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
        tot += d
        name = r["Kernel_Name"].split("(")[0].replace("danra::", "").replace("void ", "")
        if name.startswith("_ZN5danra"):  # a name the profiler's demangler gave up on (an _Float16 parameter)
            m = re.match(r"_ZN5danra\d+([a-z_0-9]+?)I((?:L[ib]\d+E)*)E", name)
            if m:
                name = m.group(1) + "<" + ", ".join(re.findall(r"L[ib](\d+)E", m.group(2))) + ">"
        per[name][0] += 1
        per[name][1] += d
        if not brief:
            wg = int(r["Workgroup_Size_X"])
            grid = (int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            print(f"{name:34s} blocks={str(grid):18s} wg={wg:5d} vgpr={r['VGPR_Count']:>4s} lds={r['LDS_Block_Size']:>6s} {d:8.1f} us")
    wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1000
    print(f"--- kernels {b - a}, busy {tot:.1f} us, wall {wall:.1f} us")
    for name, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:34s} x{n:3d} {d:8.1f} us  {100 * d / tot:5.1f} %")


if __name__ == "__main__":
    main()
