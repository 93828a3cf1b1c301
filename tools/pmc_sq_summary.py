"""Summarise the two SQ counter passes of tools/pmc_sq.sh: for the kernels with the largest total time, the counters of
the longest launch of each (kernel, grid) and the per-SIMD shares they imply.

    python tools/pmc_sq_summary.py gpurun_out/pmc_sq > profiles/r02_pmc_sq.txt

SQ_ACTIVE_* and SQ_WAVE_CYCLES count in units of 4 cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles (summed over the 1024 SIMDs).
The SIMD clock is taken from GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration."""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    launches = defaultdict(dict)  # (pass-local dispatch id) -> {counter: value, "_k": key, "_dur": ns}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "danra::" not in r["Kernel_Name"]:
                continue
            name = re.sub(r"\(.*$", "", re.sub(r"^(void )?danra::", "", r["Kernel_Name"]))
            d = launches[int(r["Dispatch_Id"])]
            d["_k"] = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            d["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return launches


def longest_by_key(launches):
    best, total, count = {}, defaultdict(int), defaultdict(int)
    for d in launches.values():
        total[d["_k"]] += d["_dur"]
        count[d["_k"]] += 1
        if d["_k"] not in best or d["_dur"] > best[d["_k"]]["_dur"]:
            best[d["_k"]] = d
    for k in [k for k in total if count[k] < 3 or k[0].startswith(("k_maxabs", "k_split_weights", "k_pack"))]:
        del total[k]  # kernels of the one-time weight packing
    return best, total


def main():
    base = sys.argv[1]
    p1, t1 = longest_by_key(load(base + "/p1/a_counter_collection.csv"))
    p2, _ = longest_by_key(load(base + "/p2/b_counter_collection.csv"))
    print("# SQ counters per launch, MI355X, conditional 64x64 at batch 64 (tools/pmc_sq.sh: two rocprofv3 --pmc passes over a")
    print("# 3-step sampling run; tools/pmc_sq_summary.py).  Shown: the longest launch of each (kernel, grid), for the eight")
    print("# (kernel, grid) pairs with the most time.  Per-SIMD shares = counter / 1024 SIMDs / kernel cycles, with SQ_ACTIVE_* in")
    print("# units of 4 cycles and SQ_VALU_MFMA_BUSY_CYCLES in cycles; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (that counter also")
    print("# runs a little before and after the kernel: for launches of 20-30 us the shares read a tenth or two low).")
    for k in sorted(t1, key=t1.get, reverse=True)[:8]:
        a, b = p1[k], p2.get(k, {})
        print(f"\n{k[0]}  grid={k[1]} wg={k[2]}  duration {a['_dur'] / 1e3:.1f} us under the profiler")
        for src in (a, b):
            for c in sorted(c for c in src if not c.startswith("_")):
                print(f"    {c:34s} {src[c]:14.0f}")
        cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            simd = 1024.0 * cyc
            parts = [f"MFMA busy {100 * a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd:.0f} %",
                     f"VALU issue (matrix instructions included) {100 * 4 * a.get('SQ_ACTIVE_INST_VALU', 0) / simd:.0f} %",
                     f"LDS issue {100 * 4 * a.get('SQ_ACTIVE_INST_LDS', 0) / simd:.0f} %"]
            if "SQ_ACTIVE_INST_SCA" in b:
                parts.append(f"scalar {100 * 4 * b['SQ_ACTIVE_INST_SCA'] / simd:.0f} %")
            if "SQ_ACTIVE_INST_VMEM" in b:
                parts.append(f"vector memory issue {100 * 4 * b['SQ_ACTIVE_INST_VMEM'] / simd:.0f} %")
            print(f"    => {cyc:.0f} kernel cycles ({cyc / a['_dur']:.2f} GHz); per-SIMD share: " + ", ".join(parts))


if __name__ == "__main__":
    main()
