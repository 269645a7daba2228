"""Oracle: quantiser arithmetic and the QuantModule forward, restated on CPU (fp32 torch ops).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Two tiers:
  T1 "fake-quant" tier: the reference's fp32 simulation, op for op (bit-exact vs the reference on CPU).
  T0 "integer" tier   : the integer codes the simulation implies and their exact accumulators
                        (numpy int64 / fp64 conv on integer-valued tensors), which the MFMA int32
                        path must reproduce bit for bit.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# grids
# ------------------------------------------------------------------------------------------------
def n_levels(n_bits, sym):
    """quant_layer.py:54 — 2**b (asymmetric) or 2**(b-1)-1 (symmetric)."""
    return 2 ** n_bits if not sym else 2 ** (n_bits - 1) - 1


def code_range(n_bits, sym):
    """Integer code interval implied by quant_layer.py:83-87 (the first clamp at :83 is dead code)."""
    nl = n_levels(n_bits, sym)
    return (-nl - 1, nl) if sym else (0, nl - 1)


# ------------------------------------------------------------------------------------------------
# T1: UniformAffineQuantizer.forward  (quant_layer.py:82-88)
# ------------------------------------------------------------------------------------------------
def uaq_forward(x, delta, zero_point, n_bits=8, sym=False):
    lo, hi = code_range(n_bits, sym)
    x_int = torch.round(x / delta) + zero_point
    x_quant = torch.clamp(x_int, lo, hi)
    return (x_quant - zero_point) * delta


def uaq_codes(x, delta, zero_point, n_bits=8, sym=False):
    """T0: the integer codes q of quant_layer.py:82-87 (int64)."""
    lo, hi = code_range(n_bits, sym)
    q = torch.clamp(torch.round(x / delta) + zero_point, lo, hi)
    return q.to(torch.int64)


# ------------------------------------------------------------------------------------------------
# UniformAffineQuantizer.init_quantization_scale  (quant_layer.py:112-181)
# ------------------------------------------------------------------------------------------------
def _lp_loss_all(pred, tgt, p):
    """quant_layer.py:26-33 with reduction='all'."""
    return (pred - tgt).abs().pow(p).mean()


def uaq_init_scale(x, n_bits=8, sym=False, channel_wise=False, scale_method="max", always_zero=False):
    """Returns (delta, zero_point) exactly as the reference computes them from the first tensor seen."""
    nl = n_levels(n_bits, sym)
    if channel_wise:
        # quant_layer.py:114-136: per out-channel recursion, then reshape to (C,1,..)
        xc = x.clone().detach()
        n_ch = xc.shape[0]
        delta = torch.zeros(n_ch, dtype=x.dtype)
        zp = torch.zeros(n_ch, dtype=x.dtype)
        for c in range(n_ch):
            d, z = uaq_init_scale(xc[c], n_bits, sym, False, scale_method, always_zero)
            delta[c], zp[c] = d, z
        shape = (-1,) + (1,) * (x.dim() - 1)
        return delta.view(shape), zp.view(shape)
    if "max" in scale_method:
        # quant_layer.py:142-160
        x_min = min(x.min().item(), 0)
        x_max = max(x.max().item(), 0)
        if "scale" in scale_method:
            x_min = x_min * (n_bits + 2) / 8
            x_max = x_max * (n_bits + 2) / 8
        x_absmax = max(abs(x_min), x_max)
        if sym:
            delta = x_absmax / nl
        else:
            delta = float(x.max().item() - x.min().item()) / (nl - 1)
        if delta < 1e-8:
            delta = 1e-8
        zero_point = round(-x_min / delta) if not (sym or always_zero) else 0
        return torch.tensor(delta).type_as(x), zero_point
    if scale_method == "mse":
        # quant_layer.py:162-177 (+ quantize() :183-190)
        x_max, x_min = x.max(), x.min()
        best = 1e10
        delta, zero_point = None, None
        for i in range(80):
            new_max = x_max * (1.0 - (i * 0.01))
            new_min = x_min * (1.0 - (i * 0.01))
            d = (new_max - new_min) / (2 ** n_bits - 1) if not always_zero else new_max / (2 ** n_bits - 1)
            z = (-new_min / d).round() if not always_zero else 0
            x_q = (torch.clamp(torch.round(x / d) + z, 0, nl - 1) - z) * d
            score = _lp_loss_all(x, x_q, 2.4)
            if score < best:
                best = score
                delta, zero_point = d, z
        return delta, zero_point
    raise NotImplementedError(scale_method)


# ------------------------------------------------------------------------------------------------
# AdaRoundQuantizer  (adaptive_rounding.py)
# ------------------------------------------------------------------------------------------------
def adaround_init_alpha(w, delta, gamma=-0.1, zeta=1.1):
    """adaptive_rounding.py:66-72."""
    x_floor = torch.floor(w / delta)
    rest = (w / delta) - x_floor
    return -torch.log((zeta - gamma) / (rest - gamma) - 1)


def adaround_forward(w, delta, zero_point, alpha, levels):
    """adaptive_rounding.py:49-61 — hard rounding (soft_targets=False): floor + (alpha >= 0)."""
    x_int = torch.floor(w / delta) + (alpha >= 0).float()
    x_quant = torch.clamp(x_int + zero_point, 0, levels - 1)
    return (x_quant - zero_point) * delta


def adaround_codes(w, delta, zero_point, alpha, levels):
    x_int = torch.floor(w / delta) + (alpha >= 0).float()
    return torch.clamp(x_int + zero_point, 0, levels - 1).to(torch.int64)


def nearest_codes(w, delta, zero_point, levels):
    """UniformAffineQuantizer on a weight (asymmetric, channel-wise): quant_layer.py:82-87."""
    return torch.clamp(torch.round(w / delta) + zero_point, 0, levels - 1).to(torch.int64)


# ------------------------------------------------------------------------------------------------
# QuantModule.forward  (quant_layer.py:248-279)
# ------------------------------------------------------------------------------------------------
def quant_module_forward(x, weight, bias, kind, fwd_kwargs, wq, aq, split=0, use_wq=True, use_aq=True):
    """T1 forward of one QuantModule.

    wq / aq: lists (one entry, or two when split != 0) of dicts
       wq: {delta, zero_point, alpha (or None -> nearest rounding), n_levels}
       aq: {delta, zero_point, n_bits, sym}
    kind: 'conv2d' | 'conv1d' | 'linear'.
    """
    if use_aq:
        if split != 0:
            x0 = uaq_forward(x[:, :split], aq[0]["delta"], aq[0]["zero_point"], aq[0]["n_bits"], aq[0]["sym"])
            x1 = uaq_forward(x[:, split:], aq[1]["delta"], aq[1]["zero_point"], aq[1]["n_bits"], aq[1]["sym"])
            x = torch.cat([x0, x1], dim=1)
        else:
            x = uaq_forward(x, aq[0]["delta"], aq[0]["zero_point"], aq[0]["n_bits"], aq[0]["sym"])
    if use_wq:
        def fq(w, q):
            if q.get("alpha") is not None:
                return adaround_forward(w, q["delta"], q["zero_point"], q["alpha"], q["n_levels"])
            return (torch.clamp(torch.round(w / q["delta"]) + q["zero_point"], 0, q["n_levels"] - 1) - q["zero_point"]) * q["delta"]
        if split != 0:
            weight = torch.cat([fq(weight[:, :split], wq[0]), fq(weight[:, split:], wq[1])], dim=1)
        else:
            weight = fq(weight, wq[0])
    fn = {"conv2d": F.conv2d, "conv1d": F.conv1d, "linear": F.linear}[kind]
    return fn(x, weight, bias, **fwd_kwargs)


# ------------------------------------------------------------------------------------------------
# T0: exact integer contraction
# ------------------------------------------------------------------------------------------------
def int_conv_exact(xq, zx, wq, zw, kind, fwd_kwargs):
    """sum_k (xq - zx) * (wq - zw) with exact accumulators.

    xq: int64 activation codes, NCHW (conv2d) / [B,C,T] (conv1d) / [...,K] (linear); zx: int scalar
    (conv zero padding happens in the dequantised domain, i.e. pads hold integer value 0 after the
    subtraction — quant_layer.py:256-276).  wq: int64 weight codes (OIHW / OI), zw: per-out-channel.
    |acc| stays far below 2**53, so fp64 convolution of integer-valued tensors is exact.
    """
    xa = (xq - zx).to(torch.float64)
    zw = torch.as_tensor(zw, dtype=torch.int64).view((-1,) + (1,) * (wq.dim() - 1))
    wa = (wq - zw).to(torch.float64)
    fn = {"conv2d": F.conv2d, "conv1d": F.conv1d, "linear": F.linear}[kind]
    acc = fn(xa, wa, None, **fwd_kwargs)
    out = acc.round().to(torch.int64)
    assert (acc - out.to(torch.float64)).abs().max().item() == 0.0
    return out


# ------------------------------------------------------------------------------------------------
# glue ops
# ------------------------------------------------------------------------------------------------
def silu(x):
    """ddim/models/diffusion.py:27-29 / nn.SiLU."""
    return x * torch.sigmoid(x)


def group_norm(x, weight, bias, groups=32, eps=1e-6):
    """GroupNorm32 (ldm util.py:214-216) / Normalize (ddim diffusion.py:32-33)."""
    return F.group_norm(x.float(), groups, weight, bias, eps)


def geglu(h):
    """ldm/modules/attention.py:42-44."""
    x, gate = h.chunk(2, dim=-1)
    return x * F.gelu(gate)


def timestep_embedding_ldm(timesteps, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:151-171 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def timestep_embedding_ddim(timesteps, dim):
    """ddim/models/diffusion.py:6-24 (sin first, half-1 denominator)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    e = timesteps.float()[:, None] * e[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1, 0, 0))
    return e


# ------------------------------------------------------------------------------------------------
# attention (fake-quant tier): one function, three callers
# ------------------------------------------------------------------------------------------------
def attention_fq(q, k, v, scale, aq_q, aq_k, aq_v, aq_w, use_aq=True, pre_scale=1.0):
    """q:[N,T,d] k,v:[N,S,d].  Follows cross_attn_forward (quant_block.py:198-219):
    sim = einsum(quant(q), quant(k)) * scale ; softmax ; einsum(quant_w(attn), quant_v(v)).
    pre_scale != 1 reproduces QuantQKMatMul (quant_block.py:125-126: the *scaled* q,k are quantised
    and no post-scale is applied) when called with scale=1.
    """
    if pre_scale != 1.0:
        q = q * pre_scale
        k = k * pre_scale
    if use_aq:
        q = uaq_forward(q, **aq_q)
        k = uaq_forward(k, **aq_k)
    sim = torch.einsum("bid,bjd->bij", q, k) * scale
    attn = sim.softmax(dim=-1)
    if use_aq:
        attn = uaq_forward(attn, **aq_w)
        v = uaq_forward(v, **aq_v)
    return torch.einsum("bij,bjd->bid", attn, v)


def attention_int(q, k, v, scale, aq_q, aq_k, aq_v, aq_w, pre_scale=1.0):
    """T0/T1 hybrid used to check the fused kernel: integer codes and exact integer contractions,
    fp64 softmax.  Returns (out fp64 [N,T,d], P codes int64 [N,T,S])."""
    if pre_scale != 1.0:
        q = q * pre_scale
        k = k * pre_scale
    qc = uaq_codes(q, aq_q["delta"], aq_q["zero_point"], aq_q["n_bits"], aq_q["sym"]) - int(aq_q["zero_point"])
    kc = uaq_codes(k, aq_k["delta"], aq_k["zero_point"], aq_k["n_bits"], aq_k["sym"]) - int(aq_k["zero_point"])
    vc = uaq_codes(v, aq_v["delta"], aq_v["zero_point"], aq_v["n_bits"], aq_v["sym"]) - int(aq_v["zero_point"])
    s_int = torch.einsum("bid,bjd->bij", qc.double(), kc.double())
    sim = s_int * (float(aq_q["delta"]) * float(aq_k["delta"]) * scale)
    p = sim.softmax(dim=-1)
    pc = uaq_codes(p.float(), aq_w["delta"], aq_w["zero_point"], aq_w["n_bits"], aq_w["sym"]) - int(aq_w["zero_point"])
    o_int = torch.einsum("bij,bjd->bid", pc.double(), vc.double())
    return o_int * (float(aq_w["delta"]) * float(aq_v["delta"])), pc


def np_int_matmul(a, b):
    """Exact int64 matmul helper for tiny cases."""
    return np.matmul(np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64))
