"""Helpers shared by the tests: build this package's model for a golden-spec variant."""
import torch

from diffusionmodelscustom_amd.modules_DANRA_conditional import Decoder, DiffusionNet, Encoder
from diffusionmodelscustom_amd.synthetic import fill_state_dict_
from oracle import golden_spec as gs


def build_model(variant="cond", size=64, seed=gs.SEED, n_heads=4):
    v = gs.VARIANTS[variant]
    lsm = torch.zeros(1, size, size) if v["lsm"] else None
    topo = torch.zeros(1, size, size) if v["topo"] else None
    enc = Encoder(1, gs.TIME_EMBEDDING, lsm_tensor=lsm, topo_tensor=topo, cond_on_img=v["cond"],
                  cond_img_dim=(1, size, size) if v["cond"] else None, block_layers=[2, 2, 2, 2],
                  num_classes=v["num_classes"], n_heads=n_heads)
    dec = Decoder(512, 1, gs.TIME_EMBEDDING, 64, n_heads=n_heads)
    model = DiffusionNet(enc, dec)
    fill_state_dict_(model, seed)
    model.eval()
    return model


def state_dict_cpu(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def build_clean_model(variant="clean_cond", size=64, seed=gs.SEED, n_heads=4):
    """The DDPM_clean_application network (reference src/main_ddpm.py:354-369), synthetic weights."""
    from diffusionmodelscustom_amd import unet
    v = gs.CLEAN_VARIANTS[variant]
    enc = unet.Encoder(1, gs.TIME_EMBEDDING, cond_on_lsm=v["lsm"], cond_on_topo=v["topo"], cond_on_img=v["cond"],
                       cond_img_dim=(1, size, size) if v["cond"] else None, block_layers=[2, 2, 2, 2],
                       num_classes=v["num_classes"], n_heads=n_heads)
    dec = unet.Decoder(512, 1, gs.TIME_EMBEDDING, 64, n_heads=n_heads)
    model = unet.DiffusionNet(enc, dec)
    fill_state_dict_(model, seed)
    model.eval()
    return model
