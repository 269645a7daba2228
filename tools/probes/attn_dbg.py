import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/q-diffusion_amd"); sys.path.insert(0, "/root/repo/tests")
import test_hip_kernels as T
from types import SimpleNamespace as NS
from oracle import quant_ref as R
from qdiff import engine
cuda = torch.device("cuda:0")
for case in T.ATTN_CASES:
    name, B, H, Tn, S, d, smb, sym, scale = case
    g = torch.Generator().manual_seed(12)
    q = torch.randn(B, Tn, H * d, generator=g); k = torch.randn(B, S, H * d, generator=g); v = torch.randn(B, S, H * d, generator=g)
    pre = (d ** -0.25) if name.startswith("ldm") else 1.0
    def mk(t, n_bits=8, s=sym, always_zero=False):
        dd, zz = R.uaq_init_scale(t, n_bits, s, False, "max", always_zero)
        return dict(delta=dd, zero_point=zz, n_bits=n_bits, sym=s)
    aq_q, aq_k, aq_v = mk(q * pre), mk(k * pre), mk(v)
    heads = lambda t, L: t.view(B, L, H, d).permute(0, 2, 1, 3).reshape(B * H, L, d)
    sim = torch.einsum("bid,bjd->bij", heads(q, Tn) * pre, heads(k, S) * pre) * scale
    p = sim.softmax(-1)
    w_sym = sym if name.startswith("cifar") else False
    aq_w = mk(p, smb, w_sym, always_zero=not name.startswith("cifar"))
    want_int, pc = R.attention_int(heads(q, Tn), heads(k, S), heads(v, S), scale, aq_q, aq_k, aq_v, aq_w, pre_scale=pre)
    ns = lambda a: NS(delta=a["delta"], zero_point=a["zero_point"], n_bits=a["n_bits"], sym=a["sym"])
    ap = engine.build_attn_plan(ns(aq_q), ns(aq_k), ns(aq_v), ns(aq_w), scale, pre, cuda)
    C = H * d
    out = engine.attention(ap, q.to(cuda), k.to(cuda), v.to(cuda), B, Tn, S, H, d, (Tn * C, C, d, 1), (S * C, C, d, 1), (S * C, C, d, 1))
    got = out.cpu().view(B, Tn, H, d).permute(0, 2, 1, 3).reshape(B * H, Tn, d)
    rng = want_int.abs().max().item()
    diff = (got.double() - want_int).abs()
    print(name, "rel max", diff.max().item() / rng, "frac>1e-5:", (diff > 1e-5 * rng).float().mean().item(), "delta_w", float(aq_w["delta"]), "zpw", aq_w["zero_point"], "P code range", pc.min().item(), pc.max().item())
