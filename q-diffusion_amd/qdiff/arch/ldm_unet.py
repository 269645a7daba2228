"""Latent-diffusion / Stable-Diffusion UNet — same architecture, module names and state-dict keys as
the reference's ldm/modules/diffusionmodules/openaimodel.py (`UNetModel`, :447-782) and
ldm/modules/attention.py (`SpatialTransformer`, `BasicTransformerBlock`, `CrossAttention`, GEGLU
feed-forward), written for this engine (2-D only, channels-last activations).

Unquantised definitions; qdiff.QuantModel rewrites them (quant_model.py / quant_block.py).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# small pieces (reference ldm/modules/diffusionmodules/util.py)
# ------------------------------------------------------------------------------------------------
_FREQ_CACHE = {}


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """cos|sin sinusoid table (reference util.py:151-171)."""
    if repeat_only:
        return timesteps[:, None].expand(-1, dim)
    half = dim // 2
    key = (dim, max_period, timesteps.device)
    freqs = _FREQ_CACHE.get(key)
    if freqs is None:
        # computed on the host exactly as the reference does, uploaded once (keeps the per-step
        # forward free of host->device copies, so it can be captured in a HIP graph)
        freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
        _FREQ_CACHE[key] = freqs
    ang = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([ang.cos(), ang.sin()], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    """GroupNorm computed in fp32 whatever the activation dtype (reference util.py:214-216)."""

    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def normalization(channels):
    return GroupNorm32(32, channels)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def checkpoint(func, inputs, params, flag):
    """Gradient checkpointing hook of the reference (util.py:102-148).  Inference-transparent: the
    engine never needs the backward pass, so the function is simply evaluated."""
    return func(*inputs)


# ------------------------------------------------------------------------------------------------
# transformer pieces (reference ldm/modules/attention.py)
# ------------------------------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        val, gate = self.proj(x).chunk(2, dim=-1)
        return val * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def forward(self, x):
        return self.net(x)


class CrossQKMatMul(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, q, k):
        return torch.einsum('b i d, b j d -> b i j', q, k) * self.scale


class CrossSMVMatMul(nn.Module):
    def forward(self, attn, v):
        return torch.einsum('b i j, b j d -> b i d', attn, v)


def _split_heads(t, h):
    b, n, c = t.shape
    return t.view(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)


def _merge_heads(t, h):
    bh, n, d = t.shape
    return t.view(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)


class CrossAttention(nn.Module):
    """Multi-head attention over `context` (self-attention when context is None).
    reference attention.py:152-198"""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.qk_matmul = CrossQKMatMul(self.scale)
        self.smv_matmul = CrossSMVMatMul()
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def forward(self, x, context=None, mask=None):
        h = self.heads
        context = x if context is None else context
        q, k, v = (_split_heads(t, h) for t in (self.to_q(x), self.to_k(context), self.to_v(context)))
        sim = self.qk_matmul(q, k)
        if mask is not None:
            mask = mask.reshape(mask.shape[0], -1)[:, None, :].repeat_interleave(h, dim=0)
            sim.masked_fill_(~mask, -torch.finfo(sim.dtype).max)
        out = self.smv_matmul(sim.softmax(dim=-1), v)
        return self.to_out(_merge_heads(out, h))


class BasicTransformerBlock(nn.Module):
    """LN-selfattn, LN-crossattn, LN-GEGLU-FF, each with a residual.  reference attention.py:222-241"""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """GN, 1x1 proj_in, tokens through transformer blocks, 1x1 proj_out, residual.
    reference attention.py:244-287"""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim) for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0))

    qd_takes_out_slot = True

    def forward(self, x, context=None, out_slot=None):
        """out_slot (engine-internal, optional): engine.CatSlot side that receives the output on the integer path."""
        from .. import quant_block as qb            # lazy: quant_block imports this module's classes
        b, c, h, w = x.shape
        if qb._int_mode(self.proj_in, self.proj_out) and not (self.proj_in.split or self.proj_out.split):
            # quantised: GroupNorm emits proj_in's int8 rows directly (no SiLU here), the 1x1 projections are
            # row GEMMs on the channels-last stream and the `+ x` rides in proj_out's epilogue.
            rows = qb._nhwc_rows(x)
            xq = qb._gn_silu_to(self.proj_in, rows, b, h * w, c, self.norm, silu=False)
            o = self.proj_in.forward_codes(xq, b, h, w)
            t = o.view(b, h * w, -1)
            last = len(self.transformer_blocks) - 1
            for i, blk in enumerate(self.transformer_blocks):
                if i == last and self.proj_out.act_quantizer.inited and isinstance(blk, qb.QuantBasicTransformerBlock):
                    t = blk(t, context, out_plan=self.proj_out.conv_plan())     # may hand back proj_out's int8 rows
                else:
                    t = blk(t, context)
            if t.dtype == torch.int8:
                out = self.proj_out.forward_codes(t, 1, 1, b * h * w, residual=rows, gn_stats=True, slot=out_slot)
            else:
                out = qb._linear_rows(self.proj_out, t.reshape(b * h * w, t.shape[-1]), residual=rows, gn_stats=True, slot=out_slot)
            return qb._rows_to_nchw(out, b, h, w)
        t = self.proj_in(self.norm(x))
        t = t.permute(0, 2, 3, 1).reshape(b, h * w, t.shape[1])
        for blk in self.transformer_blocks:
            t = blk(t, context)
        t = t.reshape(b, h, w, t.shape[-1]).permute(0, 3, 1, 2)
        return self.proj_out(t) + x


# ------------------------------------------------------------------------------------------------
# UNet building blocks (reference openaimodel.py)
# ------------------------------------------------------------------------------------------------
class TimestepBlock(nn.Module):
    """Marker: forward(x, emb, split=0)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Routes emb / context / split to the children that take them (reference openaimodel.py:74-88)."""

    def forward(self, x, emb, context=None, split=0, out_slot=None):
        """out_slot (engine-internal, optional): where the LAST layer should put its output — one side of a planned skip
        concatenation (engine.CatSlot).  Only layers that declare `qd_takes_out_slot` are told; the others (and every
        non-integer state) allocate as usual and the concatenation copies."""
        last = len(self) - 1
        for i, layer in enumerate(self):
            kw = {"out_slot": out_slot} if (i == last and out_slot is not None and getattr(layer, "qd_takes_out_slot", False)) else {}
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb, split=split, **kw)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context, **kw)
            else:
                x = layer(x, **kw)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if use_conv:
            self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=padding)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        assert x.shape[1] == self.channels
        if self.use_conv:
            from .. import quant_block as qb
            conv = self.conv
            if qb._int_mode(conv) and conv.split == 0 and conv.act_quantizer.inited and not conv.act_quantizer.running_stat:
                # quantisation commutes with nearest-neighbour replication: quantise the small map, replicate the int8
                # rows (4x fewer bytes than replicating fp32 and quantising the large map), then the integer conv
                from .. import engine
                b, c, h, w = x.shape
                rows = qb._nhwc_rows(x)
                plan = conv.conv_plan()
                xq = engine.quantize_rows(rows, plan, 1, c, b * h * w, (0, 1, rows.stride(0)))
                if engine.upsample_fold_ok(plan, 2 * h, 2 * w):
                    # the replication is folded into the convolution's im2col gather: the kernel reads the small map
                    out = conv.forward_codes(xq, b, 2 * h, 2 * w, gn_stats=True, slot=out_slot, upsample2x=True)
                    return qb._rows_to_nchw(out, b, 2 * h, 2 * w)
                up = xq.view(b, h, 1, w, 1, -1).expand(b, h, 2, w, 2, xq.shape[1]).reshape(b * 4 * h * w, xq.shape[1])
                out = conv.forward_codes(up, b, 2 * h, 2 * w, gn_stats=True, slot=out_slot)
                return qb._rows_to_nchw(out, b, 2 * h, 2 * w)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if use_conv:
            self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        assert x.shape[1] == self.channels
        if out_slot is not None and getattr(self.op, "qd_takes_out_slot", False):
            return self.op(x, out_slot=out_slot)
        return self.op(x)


class ResBlock(TimestepBlock):
    """reference openaimodel.py:163-278"""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert dims == 2
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm = use_conv, use_checkpoint, use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False), Upsample(channels, False)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False), Downsample(channels, False)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)

    def forward(self, x, emb, split=0):
        if self.updown:
            h = self.in_layers[:-1](x)
            h, x = self.h_upd(h), self.x_upd(x)
            h = self.in_layers[-1](h)
        else:
            h = self.in_layers(x)
        e = self.emb_layers(emb).type(h.dtype)[:, :, None, None]
        if self.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = self.out_layers[1:](self.out_layers[0](h) * (1 + scale) + shift)
        else:
            h = self.out_layers(h + e)
        if split > 0:
            return self.skip_connection(x, split=split) + h
        return self.skip_connection(x) + h


class QKMatMul(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = None

    def forward(self, q, k):
        return torch.einsum("bct,bcs->bts", q * self.scale, k * self.scale)


class SMVMatMul(nn.Module):
    def forward(self, weight, v):
        return torch.einsum("bts,bcs->bct", weight, v)


class QKVAttentionLegacy(nn.Module):
    """heads are split before q/k/v (reference openaimodel.py:373-406)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads
        self.qkv_matmul = QKMatMul()
        self.smv_matmul = SMVMatMul()

    def forward(self, qkv):
        bs, width, length = qkv.shape
        assert width % (3 * self.n_heads) == 0
        ch = width // (3 * self.n_heads)
        q, k, v = qkv.reshape(bs * self.n_heads, ch * 3, length).split(ch, dim=1)
        scale = 1 / math.sqrt(math.sqrt(ch))
        if self.qkv_matmul.scale is None:
            self.qkv_matmul.scale = scale
        assert self.qkv_matmul.scale == scale
        weight = self.qkv_matmul(q, k)
        weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
        return self.smv_matmul(weight, v).reshape(bs, -1, length)


class QKVAttention(nn.Module):
    """q/k/v are split before heads (reference openaimodel.py:413-444)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        bs, width, length = qkv.shape
        assert width % (3 * self.n_heads) == 0
        ch = width // (3 * self.n_heads)
        q, k, v = qkv.chunk(3, dim=1)
        scale = 1 / math.sqrt(math.sqrt(ch))
        weight = torch.einsum("bct,bcs->bts", (q * scale).view(bs * self.n_heads, ch, length),
                              (k * scale).view(bs * self.n_heads, ch, length))
        weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
        a = torch.einsum("bts,bcs->bct", weight, v.reshape(bs * self.n_heads, ch, length))
        return a.reshape(bs, -1, length)


class AttentionBlock(nn.Module):
    """GN, conv1d qkv, multi-head attention over flattened positions, conv1d proj, residual.
    reference openaimodel.py:281-327"""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.use_checkpoint = use_checkpoint
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = QKVAttention(self.num_heads) if use_new_attention_order else QKVAttentionLegacy(self.num_heads)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))

    def forward(self, x):
        b, c, *spatial = x.shape
        xf = x.reshape(b, c, -1)
        h = self.proj_out(self.attention(self.qkv(self.norm(xf))))
        return (xf + h).reshape(b, c, *spatial)


class UNetModel(nn.Module):
    """Constructor keywords as the reference (openaimodel.py:478-503).  2-D, no codebook head."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        assert dims == 2 and n_embed is None and not use_fp16, "this engine builds the 2-D fp32-residual UNet only"
        if use_spatial_transformer:
            assert context_dim is not None
        if context_dim is not None:
            assert use_spatial_transformer
            if not isinstance(context_dim, int):
                context_dim = list(context_dim)
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        assert num_heads != -1 or num_head_channels != -1
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions, self.dropout, self.channel_mult = attention_resolutions, dropout, channel_mult
        self.conv_resample, self.num_classes, self.use_checkpoint = conv_resample, num_classes, use_checkpoint
        self.dtype = torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.predict_codebook_ids = False
        self.split = False

        emb_dim = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, emb_dim), nn.SiLU(), nn.Linear(emb_dim, emb_dim))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, emb_dim)

        def res(cin, cout, **kw):
            return ResBlock(cin, emb_dim, dropout, out_channels=cout, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(ch, heads_arg):
            """attention layer at width ch (reference :567-586 head arithmetic, incl. `legacy`)."""
            if num_head_channels == -1:
                heads, dim_head = num_heads, ch // num_heads
            else:
                heads, dim_head = ch // num_head_channels, num_head_channels
            if legacy:
                dim_head = ch // heads if use_spatial_transformer else num_head_channels
            if use_spatial_transformer:
                return SpatialTransformer(ch, heads, dim_head, depth=transformer_depth, context_dim=context_dim)
            return AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=heads_arg if num_head_channels == -1 else heads,
                                  num_head_channels=dim_head, use_new_attention_order=use_new_attention_order)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        skip_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                down = res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, out_channels=ch)
                self.input_blocks.append(TimestepEmbedSequential(down))
                skip_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(res(ch, None), attn(ch, num_heads), res(ch, None))

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown else Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(normalization(ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        assert (y is not None) == (self.num_classes is not None)
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        if self.num_classes is not None:
            emb = emb + self.label_emb(y)
        h = x.type(self.dtype)
        if h.is_cuda or not torch.is_grad_enabled():
            # channels-last is the integer engine's layout; the host autograd path keeps NCHW (calibration on a CPU: the
            # CPU GroupNorm backward of this PyTorch build crashes on channels-last inputs)
            h = h.contiguous(memory_format=torch.channels_last)
        from .. import engine, quant_block as qb
        # Skip concatenations (reference :772-777) are planned: the channel counts of both halves of every
        # `th.cat([h, hs.pop()], dim=1)` are recorded by the first evaluation; from then on the producers of both halves
        # write into the two column ranges of one buffer (engine.CatSlot) and cat_channels returns a view.
        plan = self.__dict__.get("_cat_plan")
        n_in = len(self.input_blocks)
        slots = ([engine.CatSlot(*plan[i]) for i in range(n_in)]
                 if (plan is not None and len(plan) == n_in and qb.CAT_SLOTS and not torch.is_grad_enabled()) else None)
        seen = [None] * n_in
        skips = []
        for i, blk in enumerate(self.input_blocks):
            h = blk(h, emb, context, out_slot=slots[i].side(1) if slots else None)
            skips.append(h)
        h = self.middle_block(h, emb, context, out_slot=slots[-1].side(0) if slots else None)
        for blk in self.output_blocks:
            split = h.shape[1] if self.split else 0          # reference :772-777
            i = len(skips) - 1
            skip = skips.pop()
            seen[i] = (h.shape[1], skip.shape[1])
            h = blk(qb.cat_channels(h, skip), emb, context, split=split, out_slot=slots[i - 1].side(0) if (slots and i > 0) else None)
        self.__dict__["_cat_plan"] = seen
        return self._out(h.type(x.dtype))

    def _out(self, h):
        """GroupNorm -> SiLU -> conv (reference :778-781); on the integer path the normalisation emits the conv's int8 rows."""
        from .. import quant_block as qb
        conv = self.out[-1]
        if (len(self.out) == 3 and isinstance(self.out[0], nn.GroupNorm) and isinstance(self.out[1], nn.SiLU) and h.dim() == 4
                and qb._int_mode(conv) and conv.split == 0 and conv.kind == 'conv2d' and conv.act_quantizer.inited
                and h.shape[1] % 16 == 0):
            B, C, H, W = h.shape
            rows = qb._nhwc_rows(h)
            xq = qb._gn_silu_to(conv, rows, B, H * W, C, self.out[0])
            return qb._rows_to_nchw(conv.forward_codes(xq, B, H, W), B, H, W)
        return self.out(h)


def sd_v1_config():
    """configs/stable-diffusion/v1-inference.yaml:29-44 (unet_config.params)."""
    return dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)


def lsun_churches_config():
    """models/ldm/lsun_churches256/config.yaml:32-53 (unet_config.params): LDM-8 on 4 x 32 x 32 latents."""
    return dict(image_size=32, in_channels=4, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
                num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True, resblock_updown=True)


def lsun_beds_config():
    """models/ldm/lsun_beds256/config.yaml:17-34 (unet_config.params)."""
    return dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=[8, 4, 2],
                num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)
