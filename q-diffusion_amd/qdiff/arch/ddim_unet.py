"""DDPM-style pixel-space UNet (CIFAR-10 config) — same architecture, module names and state-dict
keys as the reference's ddim/models/diffusion.py (`Model`, :199-360), written for this engine:
activations are kept channels-last so every Conv2d the quantiser wraps sees K-contiguous rows.

The fp32 forward here is the *unquantised* model; qdiff.QuantModel swaps Conv/Linear for
QuantModule and ResnetBlock/AttnBlock for their fused quantised counterparts (quant_block.py).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


_FREQ_CACHE = {}


def get_timestep_embedding(timesteps, embedding_dim):
    """sin|cos table with the fairseq/tensor2tensor (half-1) denominator (reference :6-24)."""
    assert timesteps.dim() == 1
    half = embedding_dim // 2
    key = (embedding_dim, timesteps.device)
    freqs = _FREQ_CACHE.get(key)
    if freqs is None:
        # host-computed like the reference, uploaded once: no H2D copy in the per-step forward
        rate = math.log(10000) / (half - 1)
        freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -rate).to(timesteps.device)
        _FREQ_CACHE[key] = freqs
    ang = timesteps.float()[:, None] * freqs[None, :]
    emb = torch.cat([ang.sin(), ang.cos()], dim=1)
    if embedding_dim % 2:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def nonlinearity(x):
    return x * torch.sigmoid(x)  # swish (reference :27-29)


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    """nearest x2 (+3x3 conv).  reference :36-52"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        if self.with_conv:
            from .. import engine, quant_block as qb
            conv = self.conv
            if (isinstance(conv, qb.QuantModule) and qb._int_mode(conv) and conv.split == 0
                    and conv.act_quantizer.inited and not conv.act_quantizer.running_stat and x.dim() == 4):
                # quantisation commutes with nearest-neighbour replication, and the replication itself is folded into the
                # convolution's im2col gather (qd_conv_desc.upsample2x): quantise the SMALL map, convolve its up-sampling
                b, c, h, w = x.shape
                plan = conv.conv_plan()
                if engine.upsample_fold_ok(plan, 2 * h, 2 * w):
                    rows = qb._nhwc_rows(x)
                    xq = engine.quantize_rows(rows, plan, 1, c, b * h * w, (0, 1, rows.stride(0)))
                    out = conv.forward_codes(xq, b, 2 * h, 2 * w, gn_stats=True, slot=out_slot, upsample2x=True)
                    return qb._rows_to_nchw(out, b, 2 * h, 2 * w)
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if self.with_conv and out_slot is not None and getattr(self.conv, "qd_takes_out_slot", False):
            return self.conv(x, out_slot=out_slot)
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    """stride-2 3x3 conv on a (0,1,0,1)-padded input, or 2x2 average pool.  reference :55-74"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        if not self.with_conv:
            return F.avg_pool2d(x, kernel_size=2, stride=2)
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        if out_slot is not None and getattr(self.conv, "qd_takes_out_slot", False):
            return self.conv(x, out_slot=out_slot)
        return self.conv(x)


class ResnetBlock(nn.Module):
    """GN-swish-conv3x3, +temb projection, GN-swish-dropout-conv3x3, 1x1 (or 3x3) shortcut.
    reference :77-141; `split` is forwarded to the 1x1 shortcut only (:136-139)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None, split=0):
        if temb is None:
            x, temb = x
        h = self.conv1(nonlinearity(self.norm1(x)))
        h = h + self.temb_proj(nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(nonlinearity(self.norm2(h))))
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                x = self.conv_shortcut(x)
            else:
                x = self.nin_shortcut(x, split) if split != 0 else self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    """single-head spatial self-attention with 1x1 q/k/v/proj convs.  reference :144-196"""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        hn = self.norm(x)
        b, c, h, w = x.shape
        q = self.q(hn).reshape(b, c, h * w).permute(0, 2, 1)
        k = self.k(hn).reshape(b, c, h * w)
        v = self.v(hn).reshape(b, c, h * w)
        attn = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
        out = torch.bmm(v, attn.permute(0, 2, 1)).reshape(b, c, h, w)
        return x + self.proj_out(out)


def cifar10_config(split_shortcut=False):
    """The hyper-parameters of the reference's configs/cifar10.yml as the namespace `Model` expects."""
    return SimpleNamespace(
        model=SimpleNamespace(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2,
                              attn_resolutions=[16], dropout=0.1, var_type="fixedlarge", ema_rate=0.9999, ema=True,
                              resamp_with_conv=True),
        data=SimpleNamespace(image_size=32, channels=3),
        diffusion=SimpleNamespace(beta_schedule="linear", beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000),
        sampling=SimpleNamespace(batch_size=64),
        split_shortcut=split_shortcut)


class Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        mc = config.model
        ch, mult = mc.ch, tuple(mc.ch_mult)
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions, self.num_res_blocks = len(mult), mc.num_res_blocks
        self.resolution, self.in_channels = config.data.image_size, mc.in_channels
        if mc.type == 'bayesian':
            self.logvar = nn.Parameter(torch.zeros(config.diffusion.num_diffusion_timesteps))

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(mc.in_channels, ch, kernel_size=3, stride=1, padding=1)

        def res(cin, cout):
            return ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=self.temb_ch, dropout=mc.dropout)

        res_now = self.resolution
        widths = [ch * m for m in mult]
        feeds = [ch] + widths[:-1]                      # input width of every level
        self.down = nn.ModuleList()
        cur = ch
        for lvl, (cin, cout) in enumerate(zip(feeds, widths)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cur = cin
            for _ in range(self.num_res_blocks):
                stage.block.append(res(cur, cout))
                cur = cout
                if res_now in mc.attn_resolutions:
                    stage.attn.append(AttnBlock(cur))
            if lvl != self.num_resolutions - 1:
                stage.downsample = Downsample(cur, mc.resamp_with_conv)
                res_now //= 2
            self.down.append(stage)

        self.mid = nn.Module()
        self.mid.block_1 = res(cur, cur)
        self.mid.attn_1 = AttnBlock(cur)
        self.mid.block_2 = res(cur, cur)

        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cout = widths[lvl]
            for j in range(self.num_res_blocks + 1):
                skip = feeds[lvl] if j == self.num_res_blocks else widths[lvl]
                stage.block.append(res(cur + skip, cout))
                cur = cout
                if res_now in mc.attn_resolutions:
                    stage.attn.append(AttnBlock(cur))
            if lvl != 0:
                stage.upsample = Upsample(cur, mc.resamp_with_conv)
                res_now *= 2
            self.up.insert(0, stage)

        self.norm_out = Normalize(cur)
        self.conv_out = nn.Conv2d(cur, mc.out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, x, t=None, context=None):
        if t is None:
            x, t = x
        assert x.shape[2] == x.shape[3] == self.resolution
        if x.is_cuda or not torch.is_grad_enabled():       # see arch/ldm_unet.py UNetModel.forward
            x = x.contiguous(memory_format=torch.channels_last)
        temb = get_timestep_embedding(t, self.ch)
        from ..quant_block import time_mlp
        from ..quant_layer import QuantModule
        if isinstance(self.temb.dense[0], QuantModule):
            temb = time_mlp(self.temb.dense[0], self.temb.dense[1], temb, act=nonlinearity)   # two K6 launches on the integer path
        else:
            temb = self.temb.dense[1](nonlinearity(self.temb.dense[0](temb)))

        # planned skip concatenations (see arch/ldm_unet.py UNetModel.forward and engine.CatSlot): producer k of the
        # encoder writes side 1 of slot k, the decoder-side tensor that meets it writes side 0
        from .. import engine, quant_block as qb
        plan = self.__dict__.get("_cat_plan")
        nsk = 1 + self.num_resolutions * self.num_res_blocks + (self.num_resolutions - 1)
        slots = ([engine.CatSlot(*plan[k]) for k in range(nsk)]
                 if (plan is not None and len(plan) == nsk and qb.CAT_SLOTS and not torch.is_grad_enabled()) else None)
        seen = [None] * nsk

        def run(mod, *args, slot=None, **kw):
            if slot is not None and getattr(mod, "qd_takes_out_slot", False):
                return mod(*args, out_slot=slot, **kw)
            return mod(*args, **kw)

        def enc_slot():
            return slots[len(skips)].side(1) if slots else None

        def dec_slot():
            """side 0 of the slot whose skip tensor is popped next (None when nothing is left to concatenate)"""
            return slots[len(skips) - 1].side(0) if (slots and skips) else None

        skips = []
        skips.append(run(self.conv_in, x, slot=enc_slot()))
        for lvl, stage in enumerate(self.down):
            for j in range(self.num_res_blocks):
                if len(stage.attn) > 0:
                    h = run(stage.attn[j], stage.block[j](skips[-1], temb), slot=enc_slot())
                else:
                    h = run(stage.block[j], skips[-1], temb, slot=enc_slot())
                skips.append(h)
            if lvl != self.num_resolutions - 1:
                skips.append(run(stage.downsample, skips[-1], slot=enc_slot()))

        h = run(self.mid.block_2, self.mid.attn_1(self.mid.block_1(skips[-1], temb)), temb, slot=dec_slot())

        use_split = bool(getattr(self.config, "split_shortcut", False))
        for lvl in reversed(range(self.num_resolutions)):
            stage = self.up[lvl]
            for j in range(self.num_res_blocks + 1):
                k = len(skips) - 1
                skip = skips.pop()
                seen[k] = (h.size(1), skip.size(1))
                cat = qb.cat_channels(h, skip)
                # the tensor concatenated next comes from the upsample (end of a stage above level 0), else from the
                # attention block, else from the residual block: only that producer is given the slot
                ends_stage = j == self.num_res_blocks and lvl != 0
                has_attn = len(stage.attn) > 0
                kw = {"split": h.size(1)} if use_split else {}                # reference :340-346
                h = run(stage.block[j], cat, temb, slot=None if (has_attn or ends_stage) else dec_slot(), **kw)
                if has_attn:
                    h = run(stage.attn[j], h, slot=None if ends_stage else dec_slot())
            if lvl != 0:
                h = run(stage.upsample, h, slot=dec_slot())
        self.__dict__["_cat_plan"] = seen

        # output head (reference ddim/models/diffusion.py:346-348: norm_out -> swish -> conv_out): on the integer path the
        # normalisation emits conv_out's int8 rows in one pass (statistics from the last block's epilogue), as UNetModel._out does
        conv = self.conv_out
        if (isinstance(conv, QuantModule) and qb._int_mode(conv) and conv.split == 0 and conv.kind == 'conv2d'
                and conv.act_quantizer.inited and h.dim() == 4 and h.shape[1] % 16 == 0):
            B, C, H, W = h.shape
            rows = qb._nhwc_rows(h)
            xq = qb._gn_silu_to(conv, rows, B, H * W, C, self.norm_out)
            return qb._rows_to_nchw(conv.forward_codes(xq, B, H, W), B, H, W)
        if h.dtype != torch.float32:
            h = h.float()                                   # fp16 activation stream: the torch output head runs on fp32
        return self.conv_out(nonlinearity(self.norm_out(h)))
