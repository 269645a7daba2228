#!/usr/bin/env python3
"""Fused fake-quant forward+backward (csrc/fakequant.hip) vs the autograd composition, one SD level-1 activation
(16 x 320 x 64 x 64 fp32).  GPU box: python tools/bench_fakequant.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
from qdiff import quant_layer as ql  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(16, 320, 64, 64, device=dev)
w = torch.randn_like(x)
q = ql.UniformAffineQuantizer(n_bits=8, symmetric=False, channel_wise=False, scale_method="max", leaf_param=True)
with torch.no_grad():
    q(x)
for fused in (False, True):
    ql.FUSED_FAKEQUANT = fused
    for it in range(13):
        if it == 3:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        xi = x.clone().requires_grad_(True)
        (q(xi) * w).sum().backward()
    e1.record()
    torch.cuda.synchronize()
    print(f"fused={fused}: {e0.elapsed_time(e1) / 10:.3f} ms per forward+backward ({x.numel() * 4 / 1e6:.0f} MB tensor)")
