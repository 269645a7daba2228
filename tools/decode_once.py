#!/usr/bin/env python3
"""One SD-v1 first-stage decode (KL-f8, synthetic weights) of `n` latents on the bf16 kernels, for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o dec -- python tools/decode_once.py [n=4] [reps=3]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
from qdiff import synthetic  # noqa: E402
from qdiff.arch import first_stage as fs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    m, scale = fs.sd_v1_first_stage()
    synthetic.load_synthetic_weights(m, seed=0)
    m = m.to(dev).eval()
    z = torch.randn(n, 4, 64, 64, device=dev)
    for _ in range(reps):
        img = fs.decode_first_stage(m, z, scale, to_uint8=True, engine="hip")
    torch.cuda.synchronize()
    print("decoded", tuple(img.shape), img.dtype)


if __name__ == "__main__":
    main()
