#!/usr/bin/env python3
"""What do the probability codes of the benchmark's 4096-token self-attentions look like?  (GPU box.)
Hooks hip.attn_i8 during one eager SD evaluation (the model bench.py builds, batch 2), recomputes the codes of a few heads in
torch from the captured operands, and prints per call: code range per row, spread, and the share of 32 x 32 (query x key) wave
tiles whose codes fit one byte after a per-row shift to (a) the row top, (b) the row mean."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
import bench  # noqa: E402
from qdiff import hip  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    qnn, _ = bench.build_quantised_unet("sd", dev)
    from qdiff import synthetic
    x, t, c = synthetic.synthetic_inputs("sd", 2, seed=3)
    args = [a.to(dev) for a in (x, t, c)]
    seen = []
    real = hip.attn_i8

    def spy(q, k, vt, vsum, BH, H, T, S, d, Tpad, Spad, dpad, prm, *a, **kw):
        if T == S and T >= 1024:
            seen.append((q.clone(), k.clone(), prm.clone(), BH, T, S, d))
        return real(q, k, vt, vsum, BH, H, T, S, d, Tpad, Spad, dpad, prm, *a, **kw)
    hip.attn_i8 = spy
    try:
        qnn._graphs = None                              # eager launches: the spy sees every call
        with torch.no_grad():
            qnn(*args)
    finally:
        hip.attn_i8 = real
    torch.cuda.synchronize()
    for n, (q, k, prm, BH, T, S, d) in enumerate(seen):
        cs, zq, zk, dw, zpw = [float(v) for v in prm[:5].cpu()]
        rows = []
        for bh in range(0, BH, max(1, BH // 4)):
            qf = q[bh, :T, :d].float() - zq
            kf = k[bh, :S, :d].float() - zk
            p = (cs * qf @ kf.t()).softmax(-1)
            code = torch.round(p / dw) + zpw
            rows.append(code)
        code = torch.stack(rows)                        # [heads, T, S]
        mx, mn, mean, sd = code.amax(-1), code.amin(-1), code.mean(-1), code.std(-1)
        tiles = code.view(code.shape[0], T // 32, 32, S // 32, 32)
        rmax = mx.view(code.shape[0], T // 32, 32, 1, 1)
        rmean = mean.view(code.shape[0], T // 32, 32, 1, 1)
        top = ((tiles >= rmax - 254) & (tiles <= rmax)).all(-1).all(2).float().mean().item()
        mid = ((tiles >= rmean - 128) & (tiles <= rmean + 126)).all(-1).all(2).float().mean().item()
        raw = (tiles < 256).all(-1).all(2).float().mean().item()
        print(f"call {n}: T={T} d={d} dw={dw:.3e} mean code {mean.mean().item():.1f}  row max: median {mx.median().item():.0f} p99 {mx.flatten().quantile(0.99).item():.0f}"
              f"  row min: median {mn.median().item():.0f}  row std: median {sd.median().item():.1f}"
              f"  | wave tiles with hi bytes all zero: unshifted {raw:.3f}, shifted to the row top {top:.3f}, to the row mean {mid:.3f}")


if __name__ == "__main__":
    main()
