#!/usr/bin/env python3
"""Per-shape timing of qd_conv2d_bf16 on the convolutions of the SD-v1 KL-f8 decoder (one 512 x 512 image per latent):
us per launch and dense-bf16 TFLOP/s.  Usage (GPU box): python tools/bench_decoder.py [batch=4] [iters=5]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
from qdiff import hip  # noqa: E402

SHAPES = [
    # name, H(out), Cin, Cout, k, upsample2x, count per decode
    ("conv_in", 64, 4, 512, 3, False, 1),
    ("res 512 @64", 64, 512, 512, 3, False, 20),
    ("attn qkv @64", 64, 512, 1536, 1, False, 1),
    ("attn proj @64", 64, 512, 512, 1, False, 1),
    ("up 512 @128", 128, 512, 512, 3, True, 1),
    ("res 512 @128", 128, 512, 512, 3, False, 6),
    ("up 512 @256", 256, 512, 512, 3, True, 1),
    ("res 512->256 @256", 256, 512, 256, 3, False, 1),
    ("nin 512->256 @256", 256, 512, 256, 1, False, 1),
    ("res 256 @256", 256, 256, 256, 3, False, 5),
    ("up 256 @512", 512, 256, 256, 3, True, 1),
    ("res 256->128 @512", 512, 256, 128, 3, False, 1),
    ("nin 256->128 @512", 512, 256, 128, 1, False, 1),
    ("res 128 @512", 512, 128, 128, 3, False, 5),
    ("conv_out", 512, 128, 3, 3, False, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    tot_us, tot_fl = 0.0, 0.0
    for name, H, Cin, Cout, k, ups, cnt in SHAPES:
        hin = H // 2 if ups else H
        cpad = hip.pad8(Cin)
        x = torch.randn(B * hin * hin, cpad, device=dev).bfloat16()
        wt = hip.pack_weights_bf16(torch.randn(Cout, Cin, k, k, device=dev) * 0.02)
        bias = torch.zeros(Cout, device=dev)
        out = torch.empty(B * H * H, Cout, device=dev)
        part = torch.empty(B, H * H // 128, Cout, 2, device=dev)
        run = lambda: hip.conv2d_bf16(x, wt, bias, out, B, H, H, cpad, Cout, k=k, pad=k // 2, gn_part=part, upsample2x=ups)
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / iters
        fl = 2.0 * B * H * H * Cin * k * k * Cout
        tot_us += us * cnt
        tot_fl += fl * cnt
        print(f"{name:20s} M={B * H * H:8d} K={Cin * k * k:5d} N={Cout:5d}  {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  x{cnt}")
    print(f"all convolutions of {B} decodes: {tot_us / 1000:.2f} ms = {tot_us / 1000 / B:.2f} ms per image, {tot_fl / tot_us / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
