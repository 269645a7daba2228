#!/usr/bin/env python3
"""Micro-benchmark of qd_conv2d_i8 on the SD-v1 / CIFAR layer shapes (batch 16): us and TOP/s per shape.
Usage (GPU box): python tools/bench_igemm.py [w_bits=4] [iters=20]"""
import os
import sys
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
from qdiff import engine  # noqa: E402

SHAPES = [
    # name, B, Cin, H, Cout, k, stride
    ("c3 320->320 @64", 16, 320, 64, 320, 3, 1),
    ("c3 640->640 @32", 16, 640, 32, 640, 3, 1),
    ("c3 1280->1280 @16", 16, 1280, 16, 1280, 3, 1),
    ("c3 1280->1280 @8", 16, 1280, 8, 1280, 3, 1),
    ("c3 2560->1280 @16", 16, 2560, 16, 1280, 3, 1),
    ("c3 960->320 @64", 16, 960, 64, 320, 3, 1),
    ("c3 640->640 @64 (up)", 16, 640, 64, 640, 3, 1),
    ("c3s2 320->320 @64", 16, 320, 64, 320, 3, 2),
    ("c1 320->320 @64 (proj)", 16, 320, 64, 320, 1, 1),
    ("lin 320->2560 T4096 (geglu)", 16, 320, 64, 2560, 1, 1),
    ("lin 1280->320 T4096 (ff out)", 16, 1280, 64, 320, 1, 1),
    ("lin 640->5120 T1024", 16, 640, 32, 5120, 1, 1),
    ("lin 1280->10240 T256", 16, 1280, 16, 10240, 1, 1),
    ("lin 5120->1280 T256", 16, 5120, 16, 1280, 1, 1),
    ("c3 4->320 @64 (stem)", 16, 4, 64, 320, 3, 1),
    ("c3 320->4 @64 (out)", 16, 320, 64, 4, 3, 1),
    ("cifar c3 256->256 @16 B64", 64, 256, 16, 256, 3, 1),
    ("cifar c3 128->128 @32 B64", 64, 128, 32, 128, 3, 1),
]


def main():
    w_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    tot_ops = tot_us = 0.0
    only = os.environ.get("IGEMM_ONLY")
    shapes = SHAPES
    if os.environ.get("IGEMM_SHAPES"):          # "B,Cin,H,Cout,k,stride;..." custom sweep
        shapes = []
        for spec in os.environ["IGEMM_SHAPES"].split(";"):
            B, Cin, H, Cout, k, stride = (int(v) for v in spec.split(","))
            shapes.append((f"custom {spec}", B, Cin, H, Cout, k, stride))
    for name, B, Cin, H, Cout, k, stride in shapes:
        if only and not any(o in name for o in only.split("|")):
            continue
        w = torch.randn(Cout, Cin, k, k, device=dev, generator=g) * 0.05
        mx, mn = w.flatten(1).max(1)[0], w.flatten(1).min(1)[0]
        lv = 2 ** w_bits
        delta = ((mx - mn) / (lv - 1)).view(-1, 1, 1, 1)
        zp = torch.round(-mn.view(-1, 1, 1, 1) / delta)
        q = NS(delta=delta, zero_point=zp, n_levels=lv, n_bits=w_bits, sym=False)
        aq = NS(delta=torch.tensor(0.02, device=dev), zero_point=12, n_bits=8, sym=False)
        pack = engine.pack_module_weights(w, [q], 0)
        plan = engine.build_conv_plan(pack, [aq], k, k, stride, k // 2, torch.zeros(Cout, device=dev))
        xq = torch.randint(-128, 127, (B * H * H, plan.ldx), dtype=torch.int8, device=dev, generator=g)
        Ho, Wo = engine.conv_out_hw(H, H, plan)
        out = torch.empty((B * Ho * Wo, Cout), dtype=torch.float16 if os.environ.get("IGEMM_OUT") == "fp16" else torch.float32, device=dev)
        for _ in range(3):
            engine.conv_forward(plan, xq, B, H, H, Ho, Wo, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            engine.conv_forward(plan, xq, B, H, H, Ho, Wo, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / iters
        ops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        tot_ops += ops
        tot_us += us
        print(f"{name:34s} M={B * Ho * Wo:6d} N={Cout:5d} K={Cin * k * k:5d}  {us:9.1f} us  {ops / us / 1e6:8.1f} TOP/s")
    print(f"{'TOTAL':34s} {tot_us:9.1f} us  {tot_ops / tot_us / 1e6:8.1f} TOP/s")


if __name__ == "__main__":
    main()
