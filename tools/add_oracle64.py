#!/usr/bin/env python3
"""Append `out_wa_oracle64` (fp64 evaluation of the fake-quant network by the ORACLE, tier T2x) to
full-size model fixtures, so the GPU suite can state the engine's distance to the reference relative
to the reference's own fp32 rounding-noise envelope without minutes of CPU work per test run.
Also re-checks oracle(fp32) == reference output bit-exactly for the full-size models."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "q-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from golden_util import GOLDEN, build_ckpt, fixture_inputs, oracle_cfg  # noqa: E402
from oracle import unet_ref as U  # noqa: E402

for name in sys.argv[1:] or ["cifar_full", "ldm_full", "sd_full"]:
    path = os.path.join(GOLDEN, f"model_{name}.pt")
    fx = torch.load(path, map_location="cpu", weights_only=False)
    spec, ck = fx["spec"], build_ckpt(fx)
    x, t, c = fixture_inputs(fx, "test")
    for cls, key in ((U.QuantCkpt, None), (U.QuantCkpt64, "out_wa_oracle64")):
        t0 = time.time()
        Q = cls(ck, spec["w_bits"], spec["a_bits"], spec["a_sym"], spec["sm_abit"])
        cast = (lambda v: v.double()) if key else (lambda v: v)
        with torch.no_grad():
            if spec["family"] == "cifar":
                y = U.cifar_forward(Q, oracle_cfg(spec), cast(x), t, split_shortcut=spec["split"])
            else:
                y = U.ldm_forward(Q, oracle_cfg(spec), cast(x), t, None if c is None else cast(c), split=spec["split"])
        if key is None:
            d = (y - fx["out_wa"]).abs().max().item()
            print(f"[{name}] oracle fp32 vs reference: max|diff| = {d:g}  ({time.time() - t0:.0f}s)", flush=True)
            fx["oracle32_equals_reference"] = (d == 0.0)
        else:
            fx[key] = y.clone()
            d = (y - fx["out_wa"].double()).abs().max().item() / fx["out_wa"].abs().max().item()
            print(f"[{name}] reference fp32 vs fp64 oracle: {d:.3e} of range ({time.time() - t0:.0f}s)", flush=True)
    torch.save(fx, path)
