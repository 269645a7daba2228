"""Helpers shared by the golden-vector tests: rebuild the reference-format checkpoint of a model
fixture from key-derived synthetic weights + the stored quantiser scales, and build the engine's
model for a fixture spec."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        import pytest
        pytest.skip(f"{name} not generated (tools/make_golden.py)")
    return torch.load(path, map_location="cpu", weights_only=False)


def build_ckpt(fx):
    """The checkpoint the reference saved (tools/make_golden.py step 2), regenerated locally."""
    from qdiff import synthetic
    ck = {}
    for key, shape in fx["keys"]:
        leaf = key.rsplit(".", 1)[-1]
        if leaf in ("delta", "zero_point"):
            ck[key] = fx["qparams"][key].clone()
        elif leaf == "alpha":
            ck[key] = synthetic.tensor_for(key, shape, seed=0)
        else:
            assert key.startswith("model.")
            ck[key] = synthetic.tensor_for(key[len("model."):], shape, seed=0)   # fp weights are keyed without prefix
    return ck


def fixture_inputs(fx, which):
    spec = fx["spec"]
    batch, seed = (1, fx["cal_seed"]) if which == "cal" else (2, fx["test_seed"])
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch,) + tuple(spec["x"]), generator=g)
    t = torch.randint(0, 1000, (batch,), generator=g)
    if spec["family"] == "cifar":
        t = t.float()
    c = torch.randn((batch,) + tuple(spec["ctx"]), generator=g) if spec["ctx"] else None
    return x, t, c


def oracle_cfg(spec):
    if spec["family"] == "cifar":
        if spec["tiny"]:
            return dict(ch=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8], resolution=16)
        return dict(ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16], resolution=32)
    return spec["unet"]


def build_engine_model(spec):
    """This repo's fp model for a fixture spec, with its key-derived weights."""
    from types import SimpleNamespace as NS
    from qdiff import synthetic
    from qdiff.arch import ddim_unet, ldm_unet
    if spec["family"] == "cifar":
        if spec["tiny"]:
            cfg = ddim_unet.cifar10_config(split_shortcut=spec["split"])
            cfg.model = NS(type="simple", in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1,
                           attn_resolutions=[8], dropout=0.1, resamp_with_conv=True)
            cfg.data = NS(image_size=16, channels=3)
        else:
            cfg = ddim_unet.cifar10_config(split_shortcut=spec["split"])
        m = ddim_unet.Model(cfg)
    else:
        m = ldm_unet.UNetModel(**spec["unet"])
        m.split = bool(spec["split"])
    return synthetic.load_synthetic_weights(m, seed=0).eval()


def quant_params(spec):
    wq = dict(n_bits=spec["w_bits"], channel_wise=True, scale_method="max")
    aq = dict(n_bits=spec["a_bits"], channel_wise=False, scale_method="max", leaf_param=True)
    if spec["a_sym"]:
        aq["symmetric"] = True
    return wq, aq
