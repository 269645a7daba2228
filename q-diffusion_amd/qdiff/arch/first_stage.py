"""First-stage decoders: latent -> image, the step that follows the denoising loop in the
reference's LDM / Stable-Diffusion scripts (`model.decode_first_stage(samples)`: scripts/txt2img.py:…, ldm/models/
diffusion/ddpm.py:710-770 -> ldm/models/autoencoder.py:274-283 `VQModelInterface.decode`, :330-333
`AutoencoderKL.decode` -> ldm/modules/diffusionmodules/model.py:465-572 `Decoder`).

The reference does not quantise this network (q-diffusion quantises the UNet only), so the arithmetic here is the
reference's fp32 arithmetic: same operations in the same order, same module tree, therefore the same state-dict keys
(`decoder.*`, `post_quant_conv.*`, `quantize.embedding.weight`) — a `first_stage_model.*` slice of an LDM / SD checkpoint
loads directly (`load_first_stage_state_dict`).  What is MI355X-specific is how it is run: channels-last activations (the
layout MIOpen's NHWC convolutions want), an optional fused attention for the single 4096-token mid block, optional
autocast (the reference's txt2img default, scripts/txt2img.py:231-236), and batches sized for 288 GB of HBM — the
decoder's largest activation is 512 x 512 x 128 fp32 = 134 MB per image, so whole sampler batches decode in one call.

The vector quantiser of the VQ-f4 model comes from a dependency that is NOT part of the reference tree
(taming-transformers @ master, `taming.modules.vqvae.quantize.VectorQuantizer2`, environment.yml:42); its published
inference rule is restated in `VectorQuantizer.forward`: nearest codebook entry under the expanded squared distance
|z|^2 + |e|^2 - 2 z.e, evaluated on `b c h w -> (b h w) c` rows.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def nonlinearity(x):
    """swish, written as the reference writes it (model.py:34-36): x * sigmoid(x)"""
    return x * torch.sigmoid(x)


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    """nearest x2 (+ 3x3 conv), model.py:44-60"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class ResnetBlock(nn.Module):
    """GN-swish-conv3x3 twice with a 1x1 (or 3x3) shortcut; no timestep embedding in the first stage (temb_channels=0).
    model.py:85-150"""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        h = self.conv1(nonlinearity(self.norm1(x)))
        if temb is not None:
            h = h + self.temb_proj(nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(nonlinearity(self.norm2(h))))
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    """single-head spatial self-attention (model.py:153-205).  `fused=True` evaluates softmax(q k^T / sqrt(c)) v with
    torch's fused attention instead of materialising the [hw, hw] map (4096 x 4096 per image at the SD mid block); the
    default is the reference's bmm -> softmax -> bmm sequence."""

    def __init__(self, in_channels, fused=False):
        super().__init__()
        self.in_channels, self.fused = in_channels, fused
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        if self.fused:
            qt, kt, vt = (t.reshape(b, c, h * w).permute(0, 2, 1).unsqueeze(1) for t in (q, k, v))     # [b, 1, hw, c]
            o = F.scaled_dot_product_attention(qt, kt, vt, scale=int(c) ** (-0.5))
            h_ = o.squeeze(1).permute(0, 2, 1).reshape(b, c, h, w)
        else:
            q = q.reshape(b, c, h * w).permute(0, 2, 1)
            k = k.reshape(b, c, h * w)
            w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
            w_ = F.softmax(w_, dim=2)
            v = v.reshape(b, c, h * w)
            h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
        return x + self.proj_out(h_)


class Decoder(nn.Module):
    """model.py:465-572 (attn_type "vanilla" / "none"; `tanh_out`, `give_pre_end` as there)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, attn_type="vanilla", fused_attention=False,
                 **ignorekwargs):
        super().__init__()
        assert attn_type in ("vanilla", "none"), "linear attention is not used by any first-stage config of the reference"
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)

        def attn(c):
            return AttnBlock(c, fused=fused_attention) if attn_type == "vanilla" else nn.Identity()

        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = attn(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, att = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    att.append(attn(block_in))
            up = nn.Module()
            up.block, up.attn = block, att
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)                        # prepend: same indices (and state-dict keys) as the reference
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        self.last_z_shape = z.shape
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, None)), None)
        for i_level in reversed(range(self.num_resolutions)):
            stage = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = stage.block[i_block](h, None)
                if len(stage.attn) > 0:
                    h = stage.attn[i_block](h)
            if i_level != 0:
                h = stage.upsample(h)
        if self.give_pre_end:
            return h
        h = self.conv_out(nonlinearity(self.norm_out(h)))
        return torch.tanh(h) if self.tanh_out else h


class VectorQuantizer(nn.Module):
    """Inference half of taming's VectorQuantizer2 (see the module docstring): z [b, c, h, w] -> the nearest codebook
    entries, same shape.  `embedding.weight` is [n_e, e_dim] (state-dict key `quantize.embedding.weight`)."""

    def __init__(self, n_e, e_dim):
        super().__init__()
        self.n_e, self.e_dim = n_e, e_dim
        self.embedding = nn.Embedding(n_e, e_dim)

    def indices(self, z):
        zf = z.permute(0, 2, 3, 1).reshape(-1, self.e_dim)
        e = self.embedding.weight
        d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) - 2 * torch.einsum('bd,dn->bn', zf, e.t())
        return torch.argmin(d, dim=1)

    def get_codebook_entry(self, indices, shape=None):
        z_q = self.embedding(indices)
        if shape is not None:                             # (batch, height, width, channel)
            z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
        return z_q

    def forward(self, z):
        b, c, h, w = z.shape
        return self.get_codebook_entry(self.indices(z), (b, h, w, c))


class _FirstStage(nn.Module):
    def __init__(self, ddconfig, embed_dim, fused_attention=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.decoder = Decoder(**ddconfig, fused_attention=fused_attention)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)

    def load_first_stage_state_dict(self, sd, prefix="first_stage_model."):
        """Load the decode-side tensors out of a full LDM / SD checkpoint (or an autoencoder checkpoint with prefix="")."""
        own = self.state_dict()
        picked = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix) and k[len(prefix):] in own}
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError(f"first-stage checkpoint lacks {len(missing)} decode-side tensors, e.g. {missing[:3]}")
        self.load_state_dict(picked, strict=True)
        return self


class AutoencoderKLDecoder(_FirstStage):
    """Decode half of ldm.models.autoencoder.AutoencoderKL (:285-333): post_quant_conv -> Decoder."""

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    forward = decode


class VQModelDecoder(_FirstStage):
    """Decode half of ldm.models.autoencoder.VQModelInterface (:264-283): codebook lookup -> post_quant_conv -> Decoder."""

    def __init__(self, ddconfig, embed_dim, n_embed, fused_attention=False):
        super().__init__(ddconfig, embed_dim, fused_attention)
        self.n_embed = n_embed
        self.quantize = VectorQuantizer(n_embed, embed_dim)

    def decode(self, h, force_not_quantize=False):
        quant = h if force_not_quantize else self.quantize(h)
        return self.decoder(self.post_quant_conv(quant))

    forward = decode


def sd_v1_first_stage():
    """configs/stable-diffusion/v1-inference.yaml:46-67 (KL-f8); LatentDiffusion.scale_factor = 0.18215 (:17)."""
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    return AutoencoderKLDecoder(dd, embed_dim=4), 0.18215


def lsun_churches_first_stage(scale_factor=1.0):
    """models/ldm/lsun_churches256/config.yaml:32-53 (KL-f8, the same decoder shape as SD's).  `scale_by_std: true` (:17): the
    checkpoint carries `scale_factor` as a buffer (ddpm.py:460-463) — pass its value."""
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    return AutoencoderKLDecoder(dd, embed_dim=4), scale_factor


def lsun_beds_first_stage():
    """models/ldm/lsun_beds256/config.yaml:35-55 (VQ-f4); scale_factor 1.0."""
    dd = dict(double_z=False, z_channels=3, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    return VQModelDecoder(dd, embed_dim=3, n_embed=8192), 1.0


def largest_activation_bytes(dec, h, w):
    """fp32 bytes per image of the largest tensor `Decoder.forward` materialises for an h x w latent: not the output-
    resolution stream (ch x H x W) but the nearest-2x copy in front of the LAST upsampling convolution, which still has the
    channel count of the level below (SD KL-f8: 256 x 512 x 512 = twice the 128-channel stream)."""
    c = dec.conv_in.out_channels
    best = c * h * w
    for i_level in reversed(range(dec.num_resolutions)):
        c = max(c, dec.up[i_level].block[0].out_channels)
        best = max(best, c * h * w)
        c = dec.up[i_level].block[0].out_channels
        if i_level != 0:
            h, w = 2 * h, 2 * w
            best = max(best, c * h * w)
    return 4 * best


@torch.no_grad()
def decode_first_stage(first_stage, z, scale_factor=1.0, force_not_quantize=False, autocast_dtype=None, to_uint8=False,
                       max_activation_bytes=1 << 30, engine=None):
    """`LatentDiffusion.decode_first_stage` (ddpm.py:710-770, the un-tiled branch): images = decode(z / scale_factor).
    On the GPU the latents go channels-last (MIOpen NHWC convolutions); `autocast_dtype` reproduces the reference scripts'
    `precision=autocast` mode; `to_uint8` applies the scripts' clamp((x + 1) / 2, 0, 1) * 255 post-processing
    (txt2img.py: `torch.clamp((x_samples + 1.0) / 2.0, min=0.0, max=1.0)`) on the device.
    The batch is decoded in chunks whose largest activation (`largest_activation_bytes`) stays below
    `max_activation_bytes`: library convolutions index with 32-bit byte offsets, and 64 LDM-4 images put exactly 2^31 bytes
    into one tensor (measured: a GPU memory fault inside the convolution; the reference decodes its small script batches).
    Until the end of round 3 the estimate was the output-resolution stream (ch x H_out x W_out), half the true maximum: 8 SD
    latents put exactly 2^31 bytes into the upsampled 256 x 512 x 512 tensor, which the library survived or not depending
    on the allocator state of the process (`bench.py --decode` faulted once in four runs).
    `engine="hip"`: the `Decoder` runs on this package's 16-bit-float MFMA convolution / GroupNorm kernels
    (qdiff.first_stage_hip; GPU only, raises without the library) instead of the library convolutions, with fp16 operands —
    the reference scripts' precision (txt2img.py:231-236) — or, `engine="hip_bf16"`, with round 3's bf16 operands;
    `autocast_dtype` is then ignored."""
    per_image = largest_activation_bytes(first_stage.decoder, z.shape[2], z.shape[3])
    chunk = max(1, int(max_activation_bytes // max(per_image, 1)))
    if z.shape[0] > chunk:
        return torch.cat([decode_first_stage(first_stage, z[i:i + chunk], scale_factor, force_not_quantize, autocast_dtype, to_uint8,
                                             max_activation_bytes, engine) for i in range(0, z.shape[0], chunk)], dim=0)
    z = (1.0 / scale_factor) * z
    if z.is_cuda:
        z = z.contiguous(memory_format=torch.channels_last)
    kw = dict(force_not_quantize=force_not_quantize) if isinstance(first_stage, VQModelDecoder) else {}
    if engine in ("hip", "hip_fp16", "hip_bf16"):
        from ..first_stage_hip import hip_decoder
        quant = z if (not isinstance(first_stage, VQModelDecoder) or force_not_quantize) else first_stage.quantize(z)
        dt = {"hip": None, "hip_fp16": torch.float16, "hip_bf16": torch.bfloat16}[engine]
        x = hip_decoder(first_stage.decoder, dt)(first_stage.post_quant_conv(quant).float())
    elif engine is not None:
        raise ValueError(f"decode_first_stage: unknown engine {engine!r}")
    elif autocast_dtype is not None and z.is_cuda:
        with torch.autocast("cuda", dtype=autocast_dtype):
            x = first_stage.decode(z, **kw)
    else:
        x = first_stage.decode(z, **kw)
    if to_uint8:
        x = (torch.clamp((x.float() + 1.0) / 2.0, min=0.0, max=1.0) * 255.0).round().to(torch.uint8)
    return x
