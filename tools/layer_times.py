#!/usr/bin/env python3
"""Per-launch table of qd_conv2d_i8 inside one SD UNet evaluation (batch 2n): shape, us, TOP/s, GB/s."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import bench
from qdiff import hip, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
qnn, _ = bench.build_quantised_unet("sd", dev)
x, t, c = synthetic.synthetic_inputs("sd", 2 * n)
args = [a.to(dev) for a in (x, t, c)]
recs = []
orig = hip.conv2d_i8
def timed(call, acc_out=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(call, acc_out); e1.record()
    recs.append((e0, e1, call.B * call.Ho * call.Wo, call.Cout, call.kh * call.kw * sum(s["clen"] for s in call.segs), call.kh, call.stride, len(call.segs),
                 call.residual is not None, call.rowbias is not None, getattr(call, 'epilogue', 0) or 0, hip.splitk_ws_bytes(call) > 0 if getattr(call, 'splitk', None) is not False else False))
with torch.no_grad():
    qnn.model(*args)
    hip.conv2d_i8 = timed
    qnn.model(*args)
torch.cuda.synchronize()
hip.conv2d_i8 = orig
agg = collections.OrderedDict()
for e0, e1, M, N, K, kh, st, nseg, res, rb, epi, spk in recs:
    key = (M, N, K, kh, st, nseg, epi, spk, res)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1000
tot = sum(a[1] for a in agg.values())
print(f"{len(recs)} launches, {tot/1000:.2f} ms total")
for (M, N, K, kh, st, nseg, epi, spk, res), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
    ops = 2.0 * M * N * K
    byts = M * K / (kh * kh) + N * K / 2 + 4 * M * N
    print(f"M={M:6d} N={N:5d} K={K:5d} k{kh} s{st} seg{nseg} epi{epi} {'splitK' if spk else '      '} {'res' if res else '   '} x{cnt:3d}  {us/cnt:8.1f} us each  {us/1000:7.2f} ms ({100*us/tot:4.1f}%)  {ops/(us/cnt)/1e6:7.1f} TOP/s  {byts/(us/cnt)/1e3:7.1f} GB/s(min-traffic)")
