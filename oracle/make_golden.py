"""Generate ``tests/golden/*.npz`` by importing and running the reference's own code.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference); the
fixtures it writes are committed, the reference never travels.  The single stand-in is
``oracle/torchvision_standin.py`` (torchvision is not installed here).

    python -m oracle.make_golden [--skip-long]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF_DIR = os.environ.get("DANRA_REFERENCE_DIR", "/root/reference/DDPM_DANRA_conditional")
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import torchvision_standin  # noqa: E402
from oracle import golden_spec as gs  # noqa: E402
from diffusionmodelscustom_amd.synthetic import fill_state_dict_  # noqa: E402


def import_reference():
    torchvision_standin.install()
    sys.path.insert(0, REF_DIR)
    import modules_DANRA_conditional as ref_modules  # noqa: E402
    import diffusion_DANRA_conditional as ref_diffusion  # noqa: E402
    return ref_modules, ref_diffusion


def build_reference_model(ref_modules, variant, size):
    v = gs.VARIANTS[variant]
    lsm = torch.zeros(1, size, size) if v["lsm"] else None
    topo = torch.zeros(1, size, size) if v["topo"] else None
    enc = ref_modules.Encoder(1, gs.TIME_EMBEDDING, lsm_tensor=lsm, topo_tensor=topo,
                              cond_on_img=v["cond"], cond_img_dim=(1, size, size) if v["cond"] else None,
                              block_layers=[2, 2, 2, 2], num_classes=v["num_classes"])
    dec = ref_modules.Decoder(512, 1, gs.TIME_EMBEDDING, 64)
    model = ref_modules.DiffusionNet(enc, dec)
    fill_state_dict_(model, gs.SEED)
    model.eval()
    return model


def save(name, **arrays):
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def gen_noise_image(ref_diffusion):
    """G7: the forward process DiffusionUtils.noiseImage (diffusion_DANRA_conditional.py:85-103) for both schedules;
    the reference's own randn_like draw is stored, so that a restatement is checked given the same noise."""
    out = {}
    for s in ("linear", "cosine"):
        du = ref_diffusion.DiffusionUtils(gs.NOISE_IMAGE_T, gs.BETA_MIN, gs.BETA_MAX, "cpu", s)
        x = gs.randn(700, *gs.NOISE_IMAGE_SHAPE)
        t = torch.tensor(gs.NOISE_IMAGE_TIMESTEPS, dtype=torch.long)
        torch.manual_seed(gs.SEED + 7)
        noisy, noise = du.noiseImage(x, t)
        out[f"{s}_noisy"] = noisy.numpy()
        out[f"{s}_noise"] = noise.numpy()
    out["t"] = t.numpy()
    save("noise_image", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-long", action="store_true", help="skip the T=1000 chain")
    ap.add_argument("--only-state-dict", action="store_true", help="write only state_dict.json")
    ap.add_argument("--only-noise-image", action="store_true", help="write only noise_image.npz")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(gs.SEED)
    ref_modules, ref_diffusion = import_reference()

    if args.only_noise_image:
        gen_noise_image(ref_diffusion)
        return

    # G0: the checkpoint contract: state_dict keys (in registration order) and shapes per variant
    contract = {}
    for variant in gs.VARIANTS:
        m = build_reference_model(ref_modules, variant, 64)
        contract[variant] = {
            "state_dict": [[k, list(v.shape)] for k, v in m.state_dict().items()],
            "parameters": [k for k, _ in m.named_parameters()],
            "n_parameters": sum(p.numel() for p in m.parameters()),
        }
    with open(os.path.join(GOLDEN, "state_dict.json"), "w") as f:
        json.dump(contract, f, indent=0)
    print("wrote state_dict.json")
    if args.only_state_dict:
        return

    # G1: schedules (diffusion_DANRA_conditional.py:47-77)
    out = {}
    for T, s in gs.SCHEDULE_CASES:
        du = ref_diffusion.DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, "cpu", s)
        out[f"{s}_{T}_betas"] = du.betas.numpy()
        out[f"{s}_{T}_alphas"] = du.alphas.numpy()
        out[f"{s}_{T}_alpha_hat"] = du.alpha_hat.numpy()
    save("schedule", **out)

    with torch.no_grad():
        model = build_reference_model(ref_modules, "cond", 64)
        enc, dec = model.encoder, model.decoder

        # G2: both time embeddings (modules_DANRA_conditional.py:42-63,203-211)
        t = torch.tensor(gs.EMBED_TIMESTEPS, dtype=torch.long)
        save("embeddings",
             t=t.numpy(),
             encoder=enc.pos_encoding(t.unsqueeze(-1).type(torch.float), gs.TIME_EMBEDDING).numpy(),
             decoder=dec.residual_layers[0].sinusiodal_embedding(t).numpy())

        # G3: per-op outputs, inputs regenerated from seeds by the tests
        out = {}
        for lvl, C, side in gs.ATTN_LEVELS:
            x = gs.randn(100 + lvl, 1, C, side, side)
            out[f"attn{lvl}"] = enc.attention_layers[lvl](x).numpy()
        x = gs.randn(110, 1, 4, 64, 64)
        out["conv1"] = enc.conv1(x).numpy()
        x = gs.randn(111, 1, 64, 32, 32)
        out["conv2_bn_relu"] = enc.relu(enc.bn1(enc.conv2(x))).numpy()
        for lvl, cin, side in ((1, 64, 16), (2, 64, 16), (3, 128, 8), (4, 256, 4)):
            x = gs.randn(120 + lvl, 2, cin, side, side)
            out[f"layer{lvl}"] = getattr(enc, f"layer{lvl}")(x).numpy()
        tt = torch.tensor([7], dtype=torch.long)
        for k, (cin, cout, side) in enumerate(((512, 256, 2), (256, 128, 4), (128, 64, 8), (64, 64, 16))):
            f = gs.randn(130 + k, 1, cin, side, side)
            skip = gs.randn(140 + k, 1, cout, 2 * side, 2 * side)
            out[f"dec{k}"] = dec.residual_layers[k](f, skip, tt).numpy()
        f = gs.randn(150, 1, 64, 32, 32)
        out["dec_final"] = dec.final_layer(f).numpy()
        save("per_op", **out)

        # G4: full forwards, every variant, three timesteps; encoder feature maps for one case
        out = {}
        for variant in gs.VARIANTS:
            m = model if variant == "cond" else build_reference_model(ref_modules, variant, 64)
            inp = gs.variant_inputs(variant, 2, 64)
            for ti in gs.FORWARD_TIMESTEPS:
                t = torch.full((2,), ti, dtype=torch.long)
                out[f"{variant}_t{ti}"] = m(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"],
                                            inp["topo_cond"]).numpy()
        # per-sample timesteps in one batch (the general forward contract, used by training)
        inp = gs.variant_inputs("cond", 2, 64)
        t = torch.tensor([3, 731], dtype=torch.long)
        out["cond_tmixed"] = model(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"]).numpy()
        inp1 = gs.variant_inputs("cond", 1, 64)
        t1 = torch.full((1,), 500, dtype=torch.long)
        fmaps = enc(inp1["x"], t=t1, y=inp1["y"], cond_img=inp1["cond_img"], lsm_cond=inp1["lsm_cond"],
                    topo_cond=inp1["topo_cond"])
        for k, fm in enumerate(fmaps):
            out[f"cond_fmap{k + 1}"] = fm.numpy()
        save("forward64", **out)

        # G6: 128x128 forward (ddpm_DANRA_conditional_wValid__128x128.py:69)
        m128 = build_reference_model(ref_modules, "cond", 128)
        inp = gs.variant_inputs("cond", 1, 128)
        t = torch.full((1,), 500, dtype=torch.long)
        save("forward128", cond_t500=m128(inp["x"], t, inp["y"], inp["cond_img"], inp["lsm_cond"],
                                           inp["topo_cond"]).numpy())

        # G5: full sampling chains with host noise in the reference's draw order
        out = {}
        for name, c in gs.CHAIN_CASES.items():
            if name == "T1000" and args.skip_long:
                continue
            T, B = c["T"], c["B"]
            du = ref_diffusion.DiffusionUtils(T, gs.BETA_MIN, gs.BETA_MAX, "cpu", "linear")
            inp = gs.variant_inputs("cond", B, 64)
            want = gs.chain_checkpoints(T) if name != "T1000" else [1]
            # record intermediate x by wrapping the model call: x entering step i is the state after step i+1
            states = {}

            def spy(x, t, *a, _m=model, _s=states, **k):
                _s[int(t[0])] = x.clone()
                return _m(x, t, *a, **k)

            spy.eval = lambda: None
            torch.manual_seed(gs.SEED + 1)  # global generator = what randn_like consumes (diffusion:150)
            final = du.sample(inp["x"], spy, inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"],
                              topo_cond=inp["topo_cond"])
            for i in want:
                out[f"{name}_x_after_{i}"] = (final if i == 1 else states[i - 1]).numpy()
        save("chain", **out)


if __name__ == "__main__":
    main()
