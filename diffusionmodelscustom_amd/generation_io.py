"""On-disk format of a generation run, as the reference's generation script writes it
(DDPM_DANRA_conditional/generation_DANRA_conditional.py:432-441): six ``np.savez_compressed`` files per run,

    {gen,eval,lsm,cond,season,point}_samples__<model>__<var>__<HxW>__<cond>__<n>_samples.npz

each holding ONE positional array (key ``arr_0``): generated / truth / land-sea mask / conditioning images as float32
[n, 1, H, W], the season labels as int64 [n], and the cut-out corner points as int64 [4, n].  Entries that a model
variant does not use are saved as they are in the script (a ``None`` becomes a 0-d object array, which
``np.load(..., allow_pickle=False)`` refuses; ``load_generation_samples`` reports those as ``None``).
"""
import os

import numpy as np
import torch

KINDS = ("gen", "eval", "lsm", "cond", "season", "point")


def save_name(model_str, var_str, im_dim_str, cond_str, n_samples_gen):
    """generation_DANRA_conditional.py:433."""
    return model_str + '__' + var_str + '__' + im_dim_str + '__' + cond_str + '__' + str(n_samples_gen) + '_samples.npz'


def _as_array(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    return v


def save_generation_samples(save_path, name, generated_images, eval_img=None, eval_lsm=None, eval_cond=None,
                            eval_season=None, point=None):
    """Write the six files of generation_DANRA_conditional.py:436-441 under ``save_path``; returns their paths."""
    values = dict(gen=generated_images, eval=eval_img, lsm=eval_lsm, cond=eval_cond, season=eval_season, point=point)
    paths = {}
    for kind in KINDS:
        paths[kind] = os.path.join(save_path, kind + '_samples__' + name)
        np.savez_compressed(paths[kind], _as_array(values[kind]))
    return paths


def load_generation_samples(save_path, name):
    """The six arrays of a run (``None`` where the script saved a ``None`` or the file is absent).  Nothing is
    unpickled: a file that needs it is treated as holding ``None``."""
    out = {}
    for kind in KINDS:
        path = os.path.join(save_path, kind + '_samples__' + name)
        out[kind] = None
        if os.path.exists(path):
            try:
                with np.load(path, allow_pickle=False) as z:
                    out[kind] = z['arr_0']
            except ValueError:  # object array (a saved None)
                pass
    return out
