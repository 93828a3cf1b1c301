"""Summarise the two counter passes of tools/pmc_traffic.sh into the table kept as profiles/rNN_pmc_traffic.txt and the
per-family averages that bench.py quotes as roofline.traffic, kept as profiles/rNN_family_traffic.json.

    python tools/pmc_summary.py gpurun_out/pmc profiles/rNN_family_traffic.json > profiles/rNN_pmc_traffic.txt

Counter unit: KiB.  gfx950 correction: FETCH_SIZE counts a 128-byte read request as 64 bytes, so reads = 2 x FETCH_SIZE;
WRITE_SIZE is exact.  Median over the launches of each (kernel, grid size)."""
import csv
import json
import os
import re
import statistics
import sys
from collections import defaultdict


def load(path, counter):
    rows = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter or "danra::" not in r["Kernel_Name"]:
                continue
            name = re.sub(r"^(void )?danra::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            rows[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return rows


def main():
    base = sys.argv[1]
    fetch = load(os.path.join(base, "fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(base, "write", "w_counter_collection.csv"), "WRITE_SIZE")
    print("# HBM-side traffic per launch, MI355X, conditional 64x64 at batch 64: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE")
    print("# in separate passes (tools/pmc_traffic.sh; 11 denoise steps), summarised by tools/pmc_summary.py.  Counter unit: KiB.")
    print("# gfx950 correction applied in the MiB columns: FETCH_SIZE counts 128-byte read requests as 64 bytes, so")
    print("# reads = 2 x FETCH_SIZE; WRITE_SIZE is exact.  Median over the launches of each (kernel, grid size).")
    print(f"{'kernel':34s} {'grid threads':>12s} {'launches':>8s} {'FETCH_SIZE KiB':>15s} {'WRITE_SIZE KiB':>15s} {'read MiB':>9s} {'write MiB':>9s}")
    keys = sorted(fetch, key=lambda k: -(2 * statistics.median(fetch[k]) + statistics.median(write.get(k, [0]))) * len(fetch[k]))
    flash = {}
    for k in keys:
        f = statistics.median(fetch[k])
        w = statistics.median(write.get(k, [0.0]))
        print(f"{k[0]:34s} {k[1]:12d} {len(fetch[k]):8d} {f:15.1f} {w:15.1f} {2 * f / 1024:9.2f} {w / 1024:9.2f}")
        if k[0].startswith("k_flash_hd16"):
            flash[k[1]] = {"read_bytes": 2 * f * 1024, "write_bytes": w * 1024}
    if len(sys.argv) > 2:
        # per kernel family, as bench.py's roofline object averages it: HBM bytes over all launches / launches
        families = {"igemm": ("k_igemm", "k_gemm_k64", "k_splitk_epilogue", "k_conv_win", "k_conv8s2"), "flash": ("k_flash_hd16", "k_flash_wide")}
        out = {}
        for fam, prefixes in families.items():
            total, launches = 0.0, 0
            for k in fetch:
                if not k[0].startswith(prefixes):
                    continue
                wr = write.get(k, [])
                total += 1024.0 * (2.0 * sum(fetch[k]) + sum(wr))
                launches += len(fetch[k])
            if launches:
                out[fam] = {"bytes_per_launch": total / launches, "launches_counted": launches,
                            "source": "tools/pmc_traffic.sh + tools/pmc_summary.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes, "
                                      "mean over the family's launches of 11 denoise steps at batch 64"}
        if flash:
            grids = sorted(flash, reverse=True)  # larger grid = L 1024, smaller = L 256 (batch 64, 4 heads)
            out["k_flash_hd16"] = {"L1024_B64": flash[grids[0]], "L256_B64": flash[grids[-1]]}
        # the build these counters belong to: bench.py quotes the figures only while it runs this very library
        import hashlib
        lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusionmodelscustom_amd", "libdanra_hip.so")
        with open(lib, "rb") as fh:
            out["library_sha256"] = hashlib.sha256(fh.read()).hexdigest()
        with open(sys.argv[2], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
