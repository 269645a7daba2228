"""Oracle: functional CPU restatement of the reference's quantised UNet forward.

TEST INFRASTRUCTURE — see oracle/__init__.py.  `QuantCkpt` reads a reference-format checkpoint
(state-dict) and evaluates QuantModule / block forwards exactly as the reference's
fp32 simulation does (same ATen calls in the same order, so results are bit-identical to the
reference on the same machine); `cifar_forward` / `ldm_forward` walk the two UNet families.

Pinned by tests/test_oracle_golden.py against outputs of the real reference (tests/golden/).
"""
import math

import torch
import torch.nn.functional as F

from . import quant_ref as R


class QuantCkpt:
    """View of a reference-format checkpoint + the quantisation state (use_wq, use_aq)."""

    def __init__(self, sd, w_bits, a_bits=8, a_sym=False, sm_abit=8, use_wq=True, use_aq=True, prefix="model."):
        self.sd, self.prefix = sd, prefix
        self.w_bits, self.a_bits, self.a_sym, self.sm_abit = w_bits, a_bits, a_sym, sm_abit
        self.use_wq, self.use_aq = use_wq, use_aq
        self.trace = None            # set to a list to record (name, tensor) at block boundaries
        self.blocks = None           # set to a list to record (kind, module path, inputs dict, output) of every block
                                     # (teacher-forced per-block parity: tests/test_block_parity.py)
        self.sublayers = None        # set to a list to record (kind, module path, inputs dict, output) of the three sub-layers
                                     # of every transformer block (attn1 / attn2 / ff incl. their residual adds)

    def note(self, name, t):
        if self.trace is not None:
            self.trace.append((name, t.detach().clone()))
        return t

    def block(self, kind, name, out, **inputs):
        """Record one block evaluation: `name` is the module path inside the UNet (state-dict prefix)."""
        if self.blocks is not None:
            keep = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
            self.blocks.append((kind, name, keep, out.detach().clone()))
        return out

    def sub(self, kind, name, out, **inputs):
        """Record one transformer sub-layer evaluation (input rows BEFORE its LayerNorm, output AFTER the residual add)."""
        if self.sublayers is not None:
            keep = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
            self.sublayers.append((kind, name, keep, out.detach().clone()))
        return out

    def get(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd

    @staticmethod
    def _zp(v):
        # after resume the reference holds act zero points as Python ints (qdiff/utils.py:452-457)
        if torch.is_tensor(v) and v.dim() == 0:
            return int(v.item())
        return v

    def act_q(self, name, n_bits=None, sym=None):
        return dict(delta=self.get(name + ".delta"), zero_point=self._zp(self.get(name + ".zero_point")),
                    n_bits=self.a_bits if n_bits is None else n_bits, sym=self.a_sym if sym is None else sym)

    def weight_q(self, name):
        alpha = self.get(name + ".alpha") if self.has(name + ".alpha") else None
        return dict(delta=self.get(name + ".delta"), zero_point=self.get(name + ".zero_point"), alpha=alpha,
                    n_levels=2 ** self.w_bits)

    # ---- QuantModule.forward (quant_layer.py:248-279) -------------------------------------------
    def module(self, name, x, kind, split=0, **kw):
        w = self.get(name + ".weight")
        b = self.get(name + ".bias") if self.has(name + ".bias") else None
        # the UNet walks below pass `split` on every call, which equals the reference's sticky
        # behaviour (quant_layer.py:249-254: once set, a module stays split at the same channel)
        wq = aq = None
        if self.use_wq:
            wq = [self.weight_q(name + ".weight_quantizer")] + ([self.weight_q(name + ".weight_quantizer_0")] if split else [])
        if self.use_aq:
            aq = [self.act_q(name + ".act_quantizer")] + ([self.act_q(name + ".act_quantizer_0")] if split else [])
        return R.quant_module_forward(x, w, b, kind, kw, wq, aq, split=split, use_wq=self.use_wq, use_aq=self.use_aq)

    def conv(self, name, x, stride=1, padding=0, split=0):
        return self.module(name, x, "conv2d", split=split, stride=(stride, stride), padding=(padding, padding),
                           dilation=(1, 1), groups=1)

    def conv1d(self, name, x):
        return self.module(name, x, "conv1d", stride=(1,), padding=(0,), dilation=(1,), groups=1)

    def linear(self, name, x):
        return self.module(name, x, "linear")

    def gn(self, name, x, eps):
        return F.group_norm(x.float(), 32, self.get(name + ".weight"), self.get(name + ".bias"), eps).type(x.dtype)

    def ln(self, name, x):
        w = self.get(name + ".weight")
        return F.layer_norm(x, (w.shape[0],), w, self.get(name + ".bias"), 1e-5)


class QuantCkpt64(QuantCkpt):
    """Exact-arithmetic tier: the same fake-quant network evaluated in fp64 (all weights, scales and
    activations double).  The reference's fp32 simulation is itself ~1e-7-noisy at every layer, and a
    quantised network amplifies that noise at round() ties; this tier tells rounding noise (inherent,
    present in the reference too) from real defects: an exact-integer engine must sit much closer to
    this tier than the reference's own fp32 run does."""

    def __init__(self, sd, *a, **k):
        sd = {key: (v.double() if torch.is_tensor(v) and torch.is_floating_point(v) else v) for key, v in sd.items()}
        super().__init__(sd, *a, **k)

    def module(self, name, x, kind, split=0, **kw):
        return super().module(name, x.double(), kind, split=split, **kw)

    def gn(self, name, x, eps):
        return F.group_norm(x.double(), 32, self.get(name + ".weight"), self.get(name + ".bias"), eps)


# =================================================================================================
# CIFAR DDIM UNet  (ddim/models/diffusion.py:199-360 with qdiff/quant_block.py:286-386)
# =================================================================================================
def _swish(x):
    return x * torch.sigmoid(x)


def _cifar_resblock(Q, p, x, temb, cin, cout, split=0):
    """QuantResnetBlock.forward (quant_block.py:307-330)."""
    h = Q.conv(p + ".conv1", _swish(Q.gn(p + ".norm1", x, 1e-6)), 1, 1)
    h = h + Q.linear(p + ".temb_proj", _swish(temb))[:, :, None, None]
    h = Q.conv(p + ".conv2", _swish(Q.gn(p + ".norm2", h, 1e-6)), 1, 1)      # dropout: eval mode
    x0 = x
    if cin != cout:
        x = Q.conv(p + ".nin_shortcut", x, 1, 0, split=split)
    return Q.block("cifar_res", p, x + h, x=x0, emb=temb, split=split)


def _cifar_attn(Q, p, x):
    """QuantAttnBlock.forward (quant_block.py:354-386)."""
    hn = Q.gn(p + ".norm", x, 1e-6)
    q, k, v = Q.conv(p + ".q", hn), Q.conv(p + ".k", hn), Q.conv(p + ".v", hn)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    if Q.use_aq:
        q = R.uaq_forward(q, **Q.act_q(p + ".act_quantizer_q"))
        k = R.uaq_forward(k, **Q.act_q(p + ".act_quantizer_k"))
    w_ = torch.bmm(q, k)
    w_ = w_ * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    w_ = w_.permute(0, 2, 1)
    if Q.use_aq:
        v = R.uaq_forward(v, **Q.act_q(p + ".act_quantizer_v"))
        w_ = R.uaq_forward(w_, **Q.act_q(p + ".act_quantizer_w", n_bits=Q.sm_abit))
    h_ = torch.bmm(v, w_).reshape(b, c, h, w)
    return Q.block("cifar_attn", p, x + Q.conv(p + ".proj_out", h_), x=x)


def cifar_forward(Q, cfg, x, t, split_shortcut=True):
    """cfg: dict(ch, ch_mult, num_res_blocks, attn_resolutions, resolution)."""
    ch, mult, nrb = cfg["ch"], list(cfg["ch_mult"]), cfg["num_res_blocks"]
    nres = len(mult)
    temb = R.timestep_embedding_ddim(t, ch)
    temb = Q.linear("temb.dense.1", _swish(Q.linear("temb.dense.0", temb)))
    widths = [ch * m for m in mult]
    feeds = [ch] + widths[:-1]
    res = cfg["resolution"]
    hs = [Q.conv("conv_in", x, 1, 1)]
    cur = ch
    for lvl in range(nres):
        cur = feeds[lvl]
        for j in range(nrb):
            h = _cifar_resblock(Q, f"down.{lvl}.block.{j}", hs[-1], temb, cur, widths[lvl])
            cur = widths[lvl]
            if res in cfg["attn_resolutions"]:
                h = _cifar_attn(Q, f"down.{lvl}.attn.{j}", h)
            hs.append(h)
        if lvl != nres - 1:
            # Downsample: asymmetric zero pad then stride-2 conv, padding 0 (diffusion.py:69-71)
            hs.append(Q.conv(f"down.{lvl}.downsample.conv", F.pad(hs[-1], (0, 1, 0, 1), mode="constant", value=0), 2, 0))
            res //= 2
    h = hs[-1]
    h = _cifar_resblock(Q, "mid.block_1", h, temb, cur, cur)
    h = _cifar_attn(Q, "mid.attn_1", h)
    h = _cifar_resblock(Q, "mid.block_2", h, temb, cur, cur)
    for lvl in reversed(range(nres)):
        for j in range(nrb + 1):
            skip = hs.pop()
            split = h.size(1) if split_shortcut else 0
            cin = h.size(1) + skip.size(1)
            h = _cifar_resblock(Q, f"up.{lvl}.block.{j}", torch.cat([h, skip], dim=1), temb, cin, widths[lvl], split=split)
            if res in cfg["attn_resolutions"]:
                h = _cifar_attn(Q, f"up.{lvl}.attn.{j}", h)
        if lvl != 0:
            h = Q.conv(f"up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"), 1, 1)
            res *= 2
    return Q.conv("conv_out", _swish(Q.gn("norm_out", h, 1e-6)), 1, 1)


# =================================================================================================
# LDM / SD UNet  (openaimodel.py:447-782, attention.py, quant_block.py:44-282)
# =================================================================================================
def _ldm_resblock(Q, p, x, emb, cin, cout, split=0, updown=None, scale_shift=False):
    """QuantResBlock._forward (quant_block.py:83-111).  updown: None | "up" | "down" — the resblock_updown variant resamples
    h (after norm + SiLU) and x before the first convolution (:84-90; Upsample / Downsample without convolution,
    openaimodel.py:108-119,139-160); scale_shift: use_scale_shift_norm (:99-103)."""
    x_in = x
    h = F.silu(Q.gn(p + ".in_layers.0", x, 1e-5))
    if updown == "up":
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    elif updown == "down":
        h, x = F.avg_pool2d(h, kernel_size=2, stride=2), F.avg_pool2d(x, kernel_size=2, stride=2)
    h = Q.conv(p + ".in_layers.2", h, 1, 1)
    e = Q.linear(p + ".emb_layers.1", F.silu(emb)).type(h.dtype)[..., None, None]
    if scale_shift:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = Q.gn(p + ".out_layers.0", h, 1e-5) * (1 + scale) + shift
        h = Q.conv(p + ".out_layers.3", F.silu(h), 1, 1)
    else:
        h = h + e
        h = Q.conv(p + ".out_layers.3", F.silu(Q.gn(p + ".out_layers.0", h, 1e-5)), 1, 1)
    if cin == cout:
        return Q.block("ldm_res", p, x + h, x=x_in, emb=emb, split=0)
    return Q.block("ldm_res", p, Q.conv(p + ".skip_connection", x, 1, 0, split=split) + h, x=x_in, emb=emb, split=split)


def _cross_attn(Q, p, x, context, heads):
    """cross_attn_forward (quant_block.py:190-221)."""
    q = Q.linear(p + ".to_q", x)
    context = x if context is None else context
    k, v = Q.linear(p + ".to_k", context), Q.linear(p + ".to_v", context)

    def split_heads(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)
    q, k, v = split_heads(q), split_heads(k), split_heads(v)
    scale = q.shape[-1] ** -0.5
    if Q.use_aq:
        q = R.uaq_forward(q, **Q.act_q(p + ".act_quantizer_q"))
        k = R.uaq_forward(k, **Q.act_q(p + ".act_quantizer_k"))
    sim = torch.einsum('b i d, b j d -> b i j', q, k) * scale
    attn = sim.softmax(dim=-1)
    if Q.use_aq:
        attn = R.uaq_forward(attn, **Q.act_q(p + ".act_quantizer_w", n_bits=Q.sm_abit))
        v = R.uaq_forward(v, **Q.act_q(p + ".act_quantizer_v"))
    out = torch.einsum('b i j, b j d -> b i d', attn, v)
    bh, n, d = out.shape
    out = out.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, heads * d)
    return Q.linear(p + ".to_out.0", out)


def _transformer_block(Q, p, x, context, heads):
    """QuantBasicTransformerBlock._forward (quant_block.py:263-271) with GEGLU FF (attention.py:37-63)."""
    x = Q.sub("attn1", p, _cross_attn(Q, p + ".attn1", Q.ln(p + ".norm1", x), None, heads) + x, x=x, heads=heads)
    x = Q.sub("attn2", p, _cross_attn(Q, p + ".attn2", Q.ln(p + ".norm2", x), context, heads) + x, x=x, context=context, heads=heads)
    h = R.geglu(Q.linear(p + ".ff.net.0.proj", Q.ln(p + ".norm3", x)))
    return Q.sub("ff", p, Q.linear(p + ".ff.net.2", h) + x, x=x)


def _spatial_transformer(Q, p, x, context, heads):
    """SpatialTransformer.forward (attention.py:276-287), depth 1."""
    b, c, h, w = x.shape
    t = Q.conv(p + ".proj_in", Q.gn(p + ".norm", x, 1e-6))
    t = t.permute(0, 2, 3, 1).reshape(b, h * w, t.shape[1])
    t = _transformer_block(Q, p + ".transformer_blocks.0", t, context, heads)
    t = t.reshape(b, h, w, t.shape[-1]).permute(0, 3, 1, 2)
    return Q.block("sd_transformer", p, Q.conv(p + ".proj_out", t) + x, x=x, context=context)


def _attention_block(Q, p, x, heads):
    """AttentionBlock._forward + QKVAttentionLegacy + QuantQKMatMul / QuantSMVMatMul
    (openaimodel.py:321-327, 384-406; quant_block.py:123-157)."""
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = Q.conv1d(p + ".qkv", Q.gn(p + ".norm", xf, 1e-5))
    bs, width, length = qkv.shape
    ch = width // (3 * heads)
    q, k, v = qkv.reshape(bs * heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if Q.use_aq:
        qq = R.uaq_forward(q * scale, **Q.act_q(p + ".attention.qkv_matmul.act_quantizer_q"))
        kk = R.uaq_forward(k * scale, **Q.act_q(p + ".attention.qkv_matmul.act_quantizer_k"))
    else:
        qq, kk = q * scale, k * scale
    weight = torch.einsum("bct,bcs->bts", qq, kk)
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    if Q.use_aq:
        weight = R.uaq_forward(weight, **Q.act_q(p + ".attention.smv_matmul.act_quantizer_w", n_bits=Q.sm_abit, sym=False))
        v = R.uaq_forward(v, **Q.act_q(p + ".attention.smv_matmul.act_quantizer_v"))
    a = torch.einsum("bts,bcs->bct", weight, v).reshape(bs, -1, length)
    h = Q.conv1d(p + ".proj_out", a)
    return Q.block("ldm_attn", p, (xf + h).reshape(b, c, *spatial), x=x)


def ldm_forward(Q, cfg, x, t, context=None, split=True):
    """cfg: the UNetModel kwargs (model_channels, channel_mult, num_res_blocks, attention_resolutions,
    num_heads | num_head_channels, use_spatial_transformer, legacy)."""
    mc, mult, nrb = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    st = bool(cfg.get("use_spatial_transformer", False))
    ss, rud = bool(cfg.get("use_scale_shift_norm", False)), bool(cfg.get("resblock_updown", False))
    if rud and split:
        raise AttributeError("resblock_updown with the split shortcut: the reference raises here (quant_block.py:75, the "
                             "resampling ResBlock's skip connection is an Identity without `.split`)")
    nhc, nh = cfg.get("num_head_channels", -1), cfg.get("num_heads", -1)

    def heads_at(ch):
        return nh if nhc == -1 else ch // nhc

    def attn(p, h, ch):
        if st:
            return _spatial_transformer(Q, p, h, context, heads_at(ch))
        return _attention_block(Q, p, h, heads_at(ch))

    emb = Q.note("time_embed", Q.linear("time_embed.2", F.silu(Q.linear("time_embed.0", R.timestep_embedding_ldm(t, mc)))))
    Q.block("ldm_time_embed", "time_embed", emb, t=t)
    hs = []
    h = Q.note("input_blocks.0", Q.conv("input_blocks.0.0", x.float(), 1, 1))
    Q.block("conv", "input_blocks.0.0", h, x=x.float())
    hs.append(h)
    ch, ds, idx = mc, 1, 1
    chans = [mc]
    for level, m in enumerate(mult):
        for _ in range(nrb):
            h = _ldm_resblock(Q, f"input_blocks.{idx}.0", h, emb, ch, m * mc, scale_shift=ss)
            ch = m * mc
            if ds in cfg["attention_resolutions"]:
                Q.note(f"input_blocks.{idx}.0", h)
                h = attn(f"input_blocks.{idx}.1", h, ch)
            Q.note(f"input_blocks.{idx}", h)
            hs.append(h)
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            if rud:
                h = Q.note(f"input_blocks.{idx}", _ldm_resblock(Q, f"input_blocks.{idx}.0", h, emb, ch, ch, updown="down", scale_shift=ss))
            else:
                h = Q.note(f"input_blocks.{idx}", Q.block("conv", f"input_blocks.{idx}.0.op", Q.conv(f"input_blocks.{idx}.0.op", h, 2, 1), x=h))
            hs.append(h)
            chans.append(ch)
            idx += 1
            ds *= 2
    h = Q.note("middle_block.0", _ldm_resblock(Q, "middle_block.0", h, emb, ch, ch, scale_shift=ss))
    h = Q.note("middle_block.1", attn("middle_block.1", h, ch))
    h = Q.note("middle_block", _ldm_resblock(Q, "middle_block.2", h, emb, ch, ch, scale_shift=ss))
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            skip = hs.pop()
            sp = h.shape[1] if split else 0
            cin = h.shape[1] + skip.shape[1]
            h = _ldm_resblock(Q, f"output_blocks.{idx}.0", torch.cat([h, skip], dim=1), emb, cin, mc * m, split=sp, scale_shift=ss)
            ch = mc * m
            j = 1
            if ds in cfg["attention_resolutions"]:
                h = attn(f"output_blocks.{idx}.{j}", h, ch)
                j += 1
            if level and i == nrb and rud:
                h = _ldm_resblock(Q, f"output_blocks.{idx}.{j}", h, emb, ch, ch, updown="up", scale_shift=ss)
                ds //= 2
            elif level and i == nrb:
                h = Q.block("ldm_upsample", f"output_blocks.{idx}.{j}", Q.conv(f"output_blocks.{idx}.{j}.conv", F.interpolate(h, scale_factor=2, mode="nearest"), 1, 1), x=h)
                ds //= 2
            Q.note(f"output_blocks.{idx}", h)
            idx += 1
    return Q.block("ldm_head", "out", Q.conv("out.2", F.silu(Q.gn("out.0", h, 1e-5)), 1, 1), x=h)
