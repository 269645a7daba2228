#!/usr/bin/env python3
"""Debug aid: run a golden model fixture through the CPU oracle and the HIP engine and print the
per-block divergence (max|diff| / range, cosine).  Usage: python tools/trace_compare.py sd_tiny"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "q-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from golden_util import build_ckpt, build_engine_model, fixture_inputs, load_fixture, oracle_cfg, quant_params  # noqa: E402
from oracle import unet_ref as U  # noqa: E402
import qdiff  # noqa: E402
from qdiff.utils import resume_cali_model  # noqa: E402


def main(name):
    fx = load_fixture(f"model_{name}.pt")
    spec = fx["spec"]
    ck = build_ckpt(fx)
    x, t, c = fixture_inputs(fx, "test")
    Q = U.QuantCkpt(ck, spec["w_bits"], spec["a_bits"], spec["a_sym"], spec["sm_abit"])
    Q.trace = []
    with torch.no_grad():
        yo = U.ldm_forward(Q, oracle_cfg(spec), x, t, c, split=spec["split"])
    ref = dict(Q.trace)
    dev = torch.device("cuda:0")
    wq, aq = quant_params(spec)
    qnn = qdiff.QuantModel(build_engine_model(spec).to(dev), wq, aq, sm_abit=spec["sm_abit"]).to(dev).eval()
    cal = tuple(a for a in fixture_inputs(fx, "cal") if a is not None)
    with tempfile.TemporaryDirectory() as td:
        torch.save(ck, os.path.join(td, "c.pth"))
        resume_cali_model(qnn, os.path.join(td, "c.pth"), cal, quant_act=True, cond=c is not None)
    got = {}

    def hook(nm):
        def f(mod, inp, out):
            got[nm] = out.detach().float().cpu()
        return f
    for nm, mod in qnn.model.named_modules():
        if nm in ref:
            mod.register_forward_hook(hook(nm))
    with torch.no_grad():
        y = qnn(x.to(dev), t.to(dev), c.to(dev)) if c is not None else qnn(x.to(dev), t.to(dev))
    for nm, r in Q.trace:
        if nm in got:
            g = got[nm]
            d = (g - r).abs().max().item()
            cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
            print(f"{nm:28s} max|diff|={d:.3e} range={r.abs().max().item():.3e} rel={d / r.abs().max().item():.2e} cos={cos:.7f}")
    d = (y.float().cpu() - yo).abs().max().item()
    print(f"{'output':28s} max|diff|={d:.3e} rel={d / yo.abs().max().item():.2e}")


if __name__ == "__main__":
    main(sys.argv[1])
