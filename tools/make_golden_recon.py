#!/usr/bin/env python3
"""Generate tests/golden/recon_<model>.pt by running the REAL reference's calibration (qdiff/block_recon.py,
qdiff/layer_recon.py, the `recon_model` walk of scripts/sample_diffusion_ddim.py:170-191) on CPU for a few iterations.

Build container only:   python tools/make_golden_recon.py [cifar_tiny] [sd_tiny]

The fixture holds, for fixed seeds: a summary of every AdaRound `alpha` after the weight phase (count of round-up
decisions, sum, L2 norm, 64 strided samples), every activation step size after the activation phase, and the calibrated
model's output on the test inputs.  tests/test_calibration.py runs THIS repo's calibration with the same seeds.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402  (puts /root/reference first on sys.path and installs the shims)

from qdiff import QuantModel, block_reconstruction, layer_reconstruction  # noqa: E402  (the reference's)
from qdiff.adaptive_rounding import AdaRoundQuantizer  # noqa: E402
from qdiff.quant_block import BaseQuantBlock  # noqa: E402
from qdiff.quant_layer import QuantModule, UniformAffineQuantizer  # noqa: E402

# the reference's loops move every mini-batch `.to('cuda')` (block_recon.py:122, layer_recon.py:92): on this CPU-only
# container the literal device is mapped to "stay where you are"
_to = torch.Tensor.to


def _to_cpu(self, *a, **k):
    if a and isinstance(a[0], str) and a[0].startswith("cuda"):
        a = a[1:]
        if not a and not k:
            return self
    return _to(self, *a, **k)


torch.Tensor.to = _to_cpu

N_CAL, BATCH, ITERS_W, ITERS_A, SEED = 8, 4, 6, 6, 1234


def summary(t):
    f = t.detach().flatten().double()
    step = max(1, f.numel() // 64)
    return dict(numel=f.numel(), n_up=int((f >= 0).sum()), sum=float(f.sum()), l2=float(f.norm()), sample=f[::step][:64].float().clone())


def recon_walk(qnn, module, kwargs):
    """scripts/sample_diffusion_ddim.py:170-191"""
    for name, child in module.named_children():
        if isinstance(child, QuantModule):
            if child.ignore_reconstruction is True:
                continue
            layer_reconstruction(qnn, child, **kwargs)
        elif isinstance(child, BaseQuantBlock):
            if child.ignore_reconstruction is True:
                continue
            block_reconstruction(qnn, child, **kwargs)
        else:
            recon_walk(qnn, child, kwargs)


def make(name):
    spec = MG.MODELS[name]
    cond = spec["ctx"] is not None
    wq, aq = MG.quant_params(spec)
    xs, ts, cs = MG.inputs(spec, N_CAL, seed=300)
    cali = (xs, ts, cs) if cond else (xs, ts)
    test = MG.inputs(spec, 2, seed=200)
    qnn = QuantModel(MG.build_fp(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    qnn.set_quant_state(True, False)
    MG.call(qnn, *cali) if cond else MG.call(qnn, xs, ts, None)
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    iters_w = ITERS_W
    if name.startswith("ldm"):
        # with quantised activations the reference leaves the LDM AttentionBlock unwrapped (quant_block.py:389-401) and its
        # QuantQKMatMul / QuantSMVMatMul blocks hold no weights: the weight phase of the reference's own scripts raises
        # "optimizer got an empty parameter list" on them (weights are calibrated in a weights-only run and resumed with
        # --resume_w, sample_diffusion_ldm.py:448-455).  The fixture therefore covers the activation phase only.
        iters_w = 0
    else:
        kw = dict(cali_data=cali, batch_size=BATCH, iters=ITERS_W, weight=0.01, asym=True, b_range=(20, 2), warmup=0.2,
                  act_quant=False, opt_mode='mse', cond=cond)
        recon_walk(qnn, qnn, kw)
    qnn.set_quant_state(True, False)
    alphas = {k: summary(m.alpha) for k, m in qnn.named_modules() if isinstance(m, AdaRoundQuantizer)}
    qnn.eval()                                            # the capture helper leaves the model in train mode (utils.py:249)
    out_w = MG.call(qnn, *test).clone()
    # activation phase (sample_diffusion_ddim.py:197-221)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        inds = np.random.choice(xs.shape[0], 4, replace=False)
        qnn(xs[inds], ts[inds], cs[inds]) if cond else qnn(xs[inds], ts[inds])
    kw = dict(cali_data=cali, batch_size=BATCH, iters=ITERS_A, act_quant=True, opt_mode='mse', lr=4e-4, p=2.4, cond=cond)
    recon_walk(qnn, qnn, kw)
    qnn.set_quant_state(True, True)
    deltas = {k: m.delta.detach().clone() for k, m in qnn.named_modules()
              if isinstance(m, UniformAffineQuantizer) and getattr(m, "leaf_param", False) and m.inited and torch.is_tensor(m.delta)}
    qnn.eval()
    out_wa = MG.call(qnn, *test).clone()
    fx = dict(name=name, spec=spec, n_cal=N_CAL, batch=BATCH, iters_w=iters_w, iters_a=ITERS_A, seed=SEED, cal_seed=300,
              test_seed=200, alphas=alphas, deltas=deltas, out_w=out_w, out_wa=out_wa, torch_version=torch.__version__)
    path = os.path.join(MG.OUT, f"recon_{name}.pt")
    torch.save(fx, path)
    print(f"[golden] recon_{name}: {len(alphas)} AdaRound quantisers, {len(deltas)} activation step sizes, "
          f"{os.path.getsize(path) / 1e3:.0f} KB")


def make_split_layer(name="cifar_tiny"):
    """A SPLIT QuantModule reconstructed as a single LAYER unit by the reference's layer_reconstruction (layer_recon.py:50-58
    switches soft targets on for `weight_quantizer` only: `weight_quantizer_0.alpha` is given to Adam but never receives a
    gradient).  The recon walks never reach this case for the shipped models (split layers sit inside block units), so it is
    driven directly: the first split layer of the model, weight phase only.  Fixture: both alpha tensors after the run and the
    untrained initialisation of the second one."""
    spec = MG.MODELS[name]
    cond = spec["ctx"] is not None
    wq, aq = MG.quant_params(spec)
    xs, ts, cs = MG.inputs(spec, N_CAL, seed=300)
    cali = (xs, ts, cs) if cond else (xs, ts)
    qnn = QuantModel(MG.build_fp(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    qnn.set_quant_state(True, False)
    MG.call(qnn, *cali) if cond else MG.call(qnn, xs, ts, None)
    key, layer = next((k, m) for k, m in qnn.named_modules() if isinstance(m, QuantModule) and m.split != 0)
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    layer_reconstruction(qnn, layer, cali_data=cali, batch_size=BATCH, iters=ITERS_W, weight=0.01, asym=True, b_range=(20, 2),
                         warmup=0.2, act_quant=False, opt_mode='mse', cond=cond)
    a0, a1 = layer.weight_quantizer.alpha.detach().clone(), layer.weight_quantizer_0.alpha.detach().clone()
    fx = dict(name=name, spec=spec, layer=key, split=int(layer.split), n_cal=N_CAL, batch=BATCH, iters_w=ITERS_W, seed=SEED, cal_seed=300,
              alpha=a0, alpha_0=a1, soft_targets=(bool(layer.weight_quantizer.soft_targets), bool(layer.weight_quantizer_0.soft_targets)),
              torch_version=torch.__version__)
    path = os.path.join(MG.OUT, "recon_split_layer.pt")
    torch.save(fx, path)
    print(f"[golden] recon_split_layer: {key} split at {layer.split}, soft targets {fx['soft_targets']}, {os.path.getsize(path) / 1e3:.0f} KB")


if __name__ == "__main__":
    for n in sys.argv[1:] or ["cifar_tiny", "sd_tiny", "ldm_tiny"]:
        if n == "split_layer":
            make_split_layer()
        else:
            make(n)
