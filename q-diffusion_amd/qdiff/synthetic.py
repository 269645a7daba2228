"""Deterministic synthetic weights / inputs for benchmarks and parity fixtures.

No pretrained or calibrated checkpoints can be downloaded here, so every experiment uses random-init
weights of the named architecture (BASELINE.json).  To make two *different* implementations of the
same architecture (this repo's and the reference's, whose construction order differs) hold identical
weights, every tensor is generated from a seed derived from its state-dict KEY, not from the order in
which modules were constructed.  torch's CPU generator is deterministic across machines.

This file must stay importable without the rest of the package (tools/make_golden.py loads it by
path inside a process that has the *reference's* `qdiff` on sys.path).
"""
import zlib

import torch


def _gen(key, seed):
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def tensor_for(key, shape, seed=0):
    """The synthetic value of state-dict entry `key` (fp32 CPU tensor)."""
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf == "alpha":
        return torch.rand(shape, generator=g) * 2 - 1                     # AdaRound rounding choice ~ U(-1,1)
    if leaf == "weight" and len(shape) == 1:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)                # norm scale
    if leaf == "bias":
        return 0.02 * torch.randn(shape, generator=g)
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5
    return 0.02 * torch.randn(shape, generator=g)


def fill_state_dict(sd, seed=0, skip=("delta", "zero_point")):
    """New dict with every entry of `sd` (except quantiser scales) replaced by its synthetic value."""
    out = {}
    for k, v in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf in skip or not torch.is_floating_point(v):
            out[k] = v
        else:
            out[k] = tensor_for(k, v.shape, seed).to(v.dtype)
    return out


def load_synthetic_weights(model, seed=0):
    """In-place: give `model` its key-derived weights (zero-initialised layers included, so that no
    quantiser meets an all-zero tensor: reference quant_layer.py:155-157)."""
    sd = model.state_dict()
    model.load_state_dict(fill_state_dict(sd, seed), strict=True)
    return model


def synthetic_inputs(kind, batch, seed=0):
    """(x, t, context) for 'cifar' | 'ldm' | 'sd' at the BASELINE shapes ('churches': LSUN-Churches LDM-8 latents)."""
    g = torch.Generator().manual_seed(1000 + seed)
    if kind == "cifar":
        return torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 1000, (batch,), generator=g).float(), None
    if kind == "ldm":
        return torch.randn(batch, 3, 64, 64, generator=g), torch.randint(0, 1000, (batch,), generator=g), None
    if kind == "churches":
        return torch.randn(batch, 4, 32, 32, generator=g), torch.randint(0, 1000, (batch,), generator=g), None
    if kind == "sd":
        return (torch.randn(batch, 4, 64, 64, generator=g), torch.randint(0, 1000, (batch,), generator=g),
                torch.randn(batch, 77, 768, generator=g))
    raise ValueError(kind)
