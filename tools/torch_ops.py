#!/usr/bin/env python3
"""Which torch (non-qdiff) ops still run inside one SD UNet evaluation, with call sites: finds glue worth fusing."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import bench
from qdiff import synthetic
from torch.profiler import profile, ProfilerActivity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
qnn, _ = bench.build_quantised_unet("sd", dev)
x, t, c = synthetic.synthetic_inputs("sd", 2 * n)
args = [a.to(dev) for a in (x, t, c)]
with torch.no_grad():
    qnn.model(*args); qnn.model(*args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        qnn.model(*args)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=40, max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=30, max_src_column_width=90))
