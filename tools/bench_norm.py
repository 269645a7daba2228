#!/usr/bin/env python3
"""Micro-benchmark of the normalise->quantise producers (K5 GroupNorm+SiLU, K9a LayerNorm) and the row
quantiser (K1) at SD-v1 shapes, batch 16: us and achieved GB/s (algorithmic bytes: fp32 in once, int8 out)."""
import os
import sys
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
from qdiff import engine, hip  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(5e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    qp = torch.tensor([0.05, 128.0], device=dev)
    grid = hip.Grid(0, 255, 128)
    for M, C in ((65536, 320), (16384, 640), (4096, 1280)):
        x = torch.randn(M, C, device=dev, generator=g)
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        for nout in (1, 3):
            outs = [torch.empty(M, C, dtype=torch.int8, device=dev) for _ in range(nout)]
            us = timeit(lambda: hip.layernorm_quant(x, M, C, C, 1e-5, gam, bet, [qp] * nout, [grid] * nout, outs, C))
            print(f"LN  M={M:6d} C={C:5d} nout={nout}  {us:7.1f} us  {(4 * M * C + nout * M * C) / us / 1e3:7.1f} GB/s")
        o = torch.empty(M, C, dtype=torch.int8, device=dev)
        us = timeit(lambda: hip.quantize_act(x, 1, C, M, (0, 1, C), qp, grid, o, C))
        print(f"Q   M={M:6d} C={C:5d}         {us:7.1f} us  {(5 * M * C) / us / 1e3:7.1f} GB/s")
        B, S = 16, M // 16
        ws = torch.empty(hip.groupnorm_ws_bytes(B, C, S), dtype=torch.uint8, device=dev)
        us = timeit(lambda: hip.groupnorm_silu_quant(x, B, S, C, C, 32, 1e-5, gam, bet, True, qp, grid, o, C, ws))
        print(f"GN  M={M:6d} C={C:5d} silu     {us:7.1f} us  {(9 * M * C) / us / 1e3:7.1f} GB/s (two reads + one byte)")
    x = torch.randn(65536, 640, device=dev, generator=g)
    y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x))
    print(f"copy 168 MB fp32              {us:7.1f} us  {2 * x.numel() * 4 / us / 1e3:7.1f} GB/s (reference point)")


if __name__ == "__main__":
    main()
