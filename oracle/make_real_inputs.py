"""Cut a small fixture out of the sample files the reference ships
(DDPM_DANRA_conditional/final_generated_samples/*.npz, written by generation_DANRA_conditional.py:436-441): the first
8 conditioning images, land-sea masks, truth images and season labels of its ERA5-conditional run, plus the
(key, shape, dtype) of each of the six files of that run.  Data only; loaded with allow_pickle=False.

    python -m oracle.make_real_inputs
"""
import json
import os

import numpy as np

SRC = "/root/reference/DDPM_DANRA_conditional/final_generated_samples"
RUN = ("DDPM_conditional_ERA5__temp__64x64__ERA5_cond_lsm_topo_random__simple__4_seasons_ValidSplitInTime_9yrs"
       "__100_samples.npz")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    meta, keep = {}, {}
    for kind in ("gen", "eval", "lsm", "cond", "season", "point"):
        with np.load(os.path.join(SRC, f"{kind}_samples__{RUN}"), allow_pickle=False) as z:
            meta[kind] = {k: [list(z[k].shape), str(z[k].dtype)] for k in z.files}
            if kind in ("eval", "lsm", "cond", "season"):
                keep[kind] = z["arr_0"][:8]
    np.savez_compressed(os.path.join(OUT, "real_inputs.npz"), **keep)
    with open(os.path.join(OUT, "generation_files.json"), "w") as f:
        json.dump({"run": RUN, "files": meta}, f, indent=1)
    print({k: v.shape for k, v in keep.items()}, meta)


if __name__ == "__main__":
    main()
