#!/usr/bin/env python3
"""GPU-side denominators on their own (bench.py also reports them): python tools/gpu_denoms.py [sd|ldm|cifar] [n images]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import bench
from qdiff import synthetic

kind = sys.argv[1] if len(sys.argv) > 1 else "sd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
qnn, _ = bench.build_quantised_unet(kind, dev)
x, t, c = synthetic.synthetic_inputs(kind, 2 * n if kind == "sd" else n)
args = [a.to(dev) for a in (x, t, c) if a is not None]
print(json.dumps(bench.gpu_denominators(qnn, args, k=2)))
