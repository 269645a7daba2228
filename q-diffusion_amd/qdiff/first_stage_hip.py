"""First-stage decoder on the hand-written 16-bit-float MFMA kernels.

Operand type (round 4): IEEE fp16 by default — the precision the reference decodes at (scripts/txt2img.py:231-236 run the
model under fp16 autocast; 11 significant bits against bf16's 8) — on v_mfma_f32_32x32x16_f16, same rate and bytes as the
bf16 instruction; `QDIFF_DECODER_DTYPE=bf16` (or HipDecoder(dtype=torch.bfloat16)) keeps round 3's bf16 operands.  Below,
"bf16" in the entry-point names is historical: they take either type.


`HipDecoder(decoder)` evaluates a `qdiff.arch.first_stage.Decoder` (the reference's ldm/modules/diffusionmodules/model.py:465-572
`Decoder.forward`) with every convolution on `qd_conv2d_bf16` — the bf16 mode of the implicit-GEMM kernel the quantised UNet
runs on (csrc/igemm_dma.hip: v_mfma_f32_32x32x16_bf16, LDS-DMA ring) — and every GroupNorm (+ swish) on
`qd_groupnorm_silu_bf16`.  Data layout in HBM:

* the residual stream h is fp32 channels-last rows [B*H*W][C] (what a residual block adds to is never rounded);
* every convolution INPUT is bf16 rows written by the GroupNorm that precedes it (model.py:121-137: norm -> swish -> conv) —
  one read of the fp32 stream, half the bytes written, and the first-level statistics of that read come with the tensor
  (the epilogue of the convolution that produced it wrote them: qd_conv_desc.gn_part), so a GroupNorm is one finalise + one
  apply launch;
* `Upsample` (model.py:48-63) never materialises the 4x map: the convolution gathers from the half-resolution rows
  (`upsample2x`);
* the single-head 4096-token mid-block attention (model.py:144-196) takes q | k | v from ONE 1x1 convolution with bf16 output
  and runs torch's scaled_dot_product_attention on them (plumbing; 1 % of the decoder's arithmetic).

Weights are rounded to bf16 once (`qd_pack_weights_bf16`, tile order); accumulation, bias, residual adds, GroupNorm
statistics and swish are fp32.  The result therefore differs from the fp32 reference by bf16 operand rounding only;
tests/test_first_stage_hip.py states the bound (and runs this wiring on CPU against tests/abi_emulator.py).  There is no
fallback: without the library or a GPU the first launch wrapper raises.
"""
import os

import torch
import torch.nn.functional as F

from . import engine, hip
import logging

logger = logging.getLogger(__name__)

# USE_GRAPH: replay the ~150 launches of a decode as ONE HIP graph per (latent shape, weights) instead of issuing them one by
# one.  Off: measured no gain (3.97 ms vs 3.99 ms per image, round 4 call 6: the decode is GPU-bound, its launches are tens of
# microseconds to milliseconds long) and it pins a private memory pool per shape.  The replay stays as the switch of its
# equality test (tests/test_first_stage_hip.py); no environment variable selects it.
USE_GRAPH = False
DEFAULT_DTYPE = torch.bfloat16 if os.environ.get("QDIFF_DECODER_DTYPE", "fp16").lower() in ("bf16", "bfloat16") else torch.float16


class _Conv:
    __slots__ = ("wt", "bias", "cin_pad", "cout", "k", "pad", "stamp")

    def __init__(self, weight, bias, device, dtype):
        w = weight.detach().to(device=device, dtype=torch.float32)
        if w.dim() == 2:
            w = w[:, :, None, None]
        self.cout, cin, self.k = w.shape[0], w.shape[1], w.shape[2]
        if w.shape[2] != w.shape[3] or self.k not in (1, 3):
            raise hip.HipEngineError(f"first-stage convolution {tuple(w.shape)}: only 1x1 / 3x3")
        self.pad = self.k // 2
        self.cin_pad = hip.pad8(cin)
        self.wt = hip.pack_weights_bf16(w, dtype)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()


class HipDecoder:
    """Runs `decoder` (qdiff.arch.first_stage.Decoder, eval mode) on the 16-bit-float MFMA kernels.  Packed weights are built on
    first use per device and re-built when the module's parameters were re-assigned or modified in place since
    (`load_state_dict` copies in place: the version counter of the parameter moves)."""

    def __init__(self, decoder, dtype=None):
        if decoder.give_pre_end or decoder.tanh_out:
            raise hip.HipEngineError("HipDecoder: give_pre_end / tanh_out decoders are not used by the reference's configs")
        self.dec = decoder
        self.dtype = DEFAULT_DTYPE if dtype is None else dtype
        if self.dtype not in (torch.float16, torch.bfloat16):
            raise hip.HipEngineError("HipDecoder: operand type must be float16 or bfloat16")
        self._packs = {}
        self._graphs = {}

    def invalidate(self):
        self._packs.clear()
        self._graphs.clear()

    # ---- weights ----
    @staticmethod
    def _stamp(*params):
        return tuple((id(p), engine.tensor_version(p), p.data_ptr()) for p in params if p is not None)

    def _conv(self, dev, key, mod):
        p = self._packs.get((dev, key))
        stamp = self._stamp(mod.weight, mod.bias)
        if p is None or p.stamp != stamp:
            p = self._packs[(dev, key)] = _Conv(mod.weight, mod.bias, dev, self.dtype)
            p.stamp = stamp
        return p

    def _qkv(self, dev, key, attn):
        p = self._packs.get((dev, key))
        stamp = self._stamp(attn.q.weight, attn.k.weight, attn.v.weight, attn.q.bias, attn.k.bias, attn.v.bias)
        if p is None or p.stamp != stamp:
            w = torch.cat([attn.q.weight, attn.k.weight, attn.v.weight], dim=0)
            b = torch.cat([attn.q.bias, attn.k.bias, attn.v.bias], dim=0)
            p = self._packs[(dev, key)] = _Conv(w, b, dev, self.dtype)
            p.stamp = stamp
        return p

    # ---- launches ----
    @staticmethod
    def _part(B, S, C, dev):
        """first-level GroupNorm statistics written by a convolution's epilogue (128-row chunks inside one sample)"""
        return torch.empty((B, S // 128, C, 2), dtype=torch.float32, device=dev) if S % 128 == 0 else None

    def _run_conv(self, c, x, B, H, W, out_dtype=torch.float32, residual=None, stats=True, upsample2x=False):
        M = B * H * W
        out = torch.empty((M, c.cout), dtype=out_dtype, device=x.device)
        part = self._part(B, H * W, c.cout, x.device) if stats else None
        hip.conv2d_bf16(x, c.wt, c.bias, out, B, H, W, c.cin_pad, c.cout, k=c.k, pad=c.pad, residual=residual, gn_part=part,
                        upsample2x=upsample2x)
        return out, part

    def _norm(self, norm, x, part, B, S, silu):
        C = norm.num_channels
        out = torch.empty((B * S, C), dtype=self.dtype, device=x.device)
        ws = torch.empty(hip.groupnorm_ws_bytes(B, C, S), dtype=torch.uint8, device=x.device)
        hip.groupnorm_silu_bf16(x, B, S, C, norm.num_groups, norm.eps, norm.weight.detach().float(), norm.bias.detach().float(),
                                silu, out, ws, part=part)
        return out

    def _resblock(self, key, blk, x, part, B, H, W):
        dev, S = x.device, H * W
        if blk.use_conv_shortcut and blk.in_channels != blk.out_channels:
            raise hip.HipEngineError("HipDecoder: 3x3 conv_shortcut residual blocks are not used by the reference's decoders")
        a = self._norm(blk.norm1, x, part, B, S, True)
        h1, p1 = self._run_conv(self._conv(dev, key + ".conv1", blk.conv1), a, B, H, W)
        b = self._norm(blk.norm2, h1, p1, B, S, True)
        if blk.in_channels != blk.out_channels:
            x, _ = self._run_conv(self._conv(dev, key + ".nin", blk.nin_shortcut), x.to(self.dtype), B, H, W, stats=False)
        return self._run_conv(self._conv(dev, key + ".conv2", blk.conv2), b, B, H, W, residual=x)

    def _attn(self, key, att, x, part, B, H, W):
        dev, S, C = x.device, H * W, att.in_channels
        hn = self._norm(att.norm, x, part, B, S, False)
        qkv, _ = self._run_conv(self._qkv(dev, key + ".qkv", att), hn, B, H, W, out_dtype=self.dtype, stats=False)
        q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B, 1, S, C) for i in range(3))
        o = F.scaled_dot_product_attention(q, k, v, scale=int(C) ** (-0.5)).reshape(B * S, C).contiguous()
        return self._run_conv(self._conv(dev, key + ".proj", att.proj_out), o, B, H, W, residual=x)

    @torch.no_grad()
    def __call__(self, z):
        """z: fp32 [B, z_channels, h, w] on the GPU (after post_quant_conv) -> fp32 [B, out_ch, H, W].
        With USE_GRAPH the walk is captured once per (latent shape, state of the weights) and replayed as one HIP
        graph (reference: one `Decoder.forward` per batch, model.py:538-572); bit-identical, measured no faster."""
        if not (USE_GRAPH and z.is_cuda) or torch.cuda.is_current_stream_capturing():
            return self._walk(z)
        key = (z.device, tuple(z.shape), z.dtype, self._stamp(*self.dec.parameters()))
        g = self._graphs.get(key)
        if g is None:
            zs = z.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._walk(zs)                  # packs, allocator, library attention heuristics: outside the capture
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._walk(zs)
            if len(self._graphs) >= 4:          # a few batch shapes at most (full chunks + a ragged last one)
                self._graphs.clear()
            g = self._graphs[key] = (graph, zs, out)
        graph, zs, out = g
        zs.copy_(z)
        graph.replay()
        return out.clone()                      # the static output belongs to the graph: the next replay overwrites it

    def _walk(self, z):
        """z: fp32 [B, z_channels, h, w] on the GPU (after post_quant_conv) -> fp32 [B, out_ch, H, W] (NCHW view of NHWC rows)"""
        d, dev = self.dec, z.device             # (host tensors: the first launch wrapper raises — there is no host path)
        B, zc, H, W = z.shape
        from .arch.first_stage import largest_activation_bytes
        if B * largest_activation_bytes(d, H, W) >= 1 << 32:      # fp32 bytes of the largest tensor of the walk (the last up-sampling level)
            raise HipBatchTooLarge("HipDecoder: batch too large for 32-bit row offsets; decode in chunks (decode_first_stage does)")
        cin = self._conv(dev, "conv_in", d.conv_in)
        x0 = torch.zeros((B * H * W, cin.cin_pad), dtype=self.dtype, device=dev)
        x0[:, :zc] = z.permute(0, 2, 3, 1).reshape(B * H * W, zc)
        h, part = self._run_conv(cin, x0, B, H, W)
        h, part = self._resblock("mid.block_1", d.mid.block_1, h, part, B, H, W)
        a1 = d.mid.attn_1
        if all(hasattr(a1, n) for n in ("q", "k", "v", "proj_out", "norm")):      # AttnBlock (attn_type "vanilla")
            h, part = self._attn("mid.attn_1", a1, h, part, B, H, W)
        elif not isinstance(a1, torch.nn.Identity):                                # attn_type "none" is nn.Identity; anything else
            raise hip.HipEngineError(f"HipDecoder: mid-block attention {type(a1).__name__} has no kernel here "
                                     "(only AttnBlock / Identity, reference model.py:139-150)")
        h, part = self._resblock("mid.block_2", d.mid.block_2, h, part, B, H, W)
        for i_level in reversed(range(d.num_resolutions)):
            stage = d.up[i_level]
            for i_block in range(d.num_res_blocks + 1):
                h, part = self._resblock(f"up.{i_level}.block.{i_block}", stage.block[i_block], h, part, B, H, W)
                if len(stage.attn) > 0:
                    h, part = self._attn(f"up.{i_level}.attn.{i_block}", stage.attn[i_block], h, part, B, H, W)
            if i_level != 0:
                if stage.upsample.with_conv:
                    H, W = 2 * H, 2 * W
                    h, part = self._run_conv(self._conv(dev, f"up.{i_level}.upsample", stage.upsample.conv), h.to(self.dtype),
                                             B, H, W, upsample2x=True)
                else:
                    C = h.shape[1]
                    h = h.view(B, H, 1, W, 1, C).expand(B, H, 2, W, 2, C).reshape(B * 4 * H * W, C)
                    H, W, part = 2 * H, 2 * W, None
        a = self._norm(d.norm_out, h, part, B, H * W, True)
        out, _ = self._run_conv(self._conv(dev, "conv_out", d.conv_out), a, B, H, W, stats=False)
        return out.view(B, H, W, -1).permute(0, 3, 1, 2)


def hip_decoder(decoder, dtype=None):
    """the HipDecoder of a Decoder module for an operand type (kept on the module, so packed weights are built once)"""
    dtype = DEFAULT_DTYPE if dtype is None else dtype
    cache = decoder.__dict__.setdefault("_hip_decoder", {})
    hd = cache.get(dtype)
    if hd is None:
        hd = cache[dtype] = HipDecoder(decoder, dtype)
    return hd


# ------------------------------------------------------------------------------------------------
# the reference's own Decoder class (drop-in use inside the q-diffusion source tree)
# ------------------------------------------------------------------------------------------------
class HipBatchTooLarge(hip.HipEngineError):
    """A size limit of ONE call (32-bit row offsets), not a property of the decoder: callers chunk or fall back for that call."""


# QDIFF_ADOPT_DECODER: "autocast" (default) — the reference's `Decoder.forward` is served by this package's fp16-operand MFMA
# kernels only while autocast is active, i.e. where the reference itself already decodes in fp16 (the scripts' default
# `--precision autocast`, txt2img.py:231-236); an fp32 decode (`--precision full`, the FID runs of sample_diffusion_ldm.py)
# stays the reference's own fp32 library path, bit for bit.  "always": adopt fp32 decodes too (1e-3 of range from fp32, 7x
# faster).  "0": never.
ADOPT_DECODER = os.environ.get("QDIFF_ADOPT_DECODER", "autocast").lower()
if ADOPT_DECODER in ("1", "on", "true"):
    ADOPT_DECODER = "autocast"
_ADOPT_LOGGED = [False]


def _on_device(z):
    return bool(z.is_cuda)


def _autocast_gpu_dtype():
    """torch >= 2.4 spells it get_autocast_dtype('cuda'); the builds the reference's environment pins only have the old name."""
    get = getattr(torch, 'get_autocast_dtype', None)
    return get('cuda') if get is not None else torch.get_autocast_gpu_dtype()


def adopt_reference_decoder():
    """Unmodified reference scripts never call this package's first-stage code: `model.decode_first_stage(samples)`
    (ldm/models/diffusion/ddpm.py:710-770) ends in `AutoencoderKL.decode` -> `self.decoder(z)` (autoencoder.py:330-333), an
    instance of the REFERENCE's `Decoder` (ldm/modules/diffusionmodules/model.py:465-572), which lives outside the UNet that
    `QuantModel` wraps.  When that class is importable its `forward` is therefore bound — on the CLASS, once — to a dispatcher:
    GPU tensor, no autograd, eval mode, the plain configuration (no `give_pre_end` / `tanh_out`) -> `HipDecoder` on this very
    module (same attribute names as this package's mirror; fp16 operands, the precision the scripts' autocast runs the
    decoder at, txt2img.py:231-236); anything else — CPU tensors, training, an unsupported layer — runs the reference's own
    forward, untouched.  Called by QuantModel when it wraps a model (qdiff/quant_model.py: _adopt_reference_modules).
    Returns the class, or None when the reference's `ldm` package is not importable / adoption is switched off."""
    if ADOPT_DECODER in ("0", "off", "false", "no"):
        return None
    try:
        from ldm.modules.diffusionmodules import model as ref_model
    except Exception:  # noqa: BLE001 - optional dependency
        return None
    cls = getattr(ref_model, "Decoder", None)
    if cls is None or cls.__dict__.get("_qd_hip_forward"):
        return cls
    ref_forward = cls.forward

    def forward(self, z):
        if (torch.is_tensor(z) and z.dim() == 4 and _on_device(z) and not torch.is_grad_enabled() and not self.training
                and not getattr(self, "give_pre_end", False) and not getattr(self, "tanh_out", False)
                and not self.__dict__.get("_qd_hip_unsupported") and (ADOPT_DECODER == "always" or (z.is_cuda and torch.is_autocast_enabled()))
                and hip.available()):
            try:
                out = hip_decoder(self)(z.float())
            except HipBatchTooLarge:
                return ref_forward(self, z)                           # this call only: the next (smaller) batch takes the kernels again
            except hip.HipEngineError:
                self.__dict__["_qd_hip_unsupported"] = True           # a layer this engine does not implement: the reference's own code
                return ref_forward(self, z)
            if not _ADOPT_LOGGED[0]:
                _ADOPT_LOGGED[0] = True
                logger.info("first-stage Decoder.forward runs on qdiff's fp16-operand MFMA kernels (QDIFF_ADOPT_DECODER=%s)", ADOPT_DECODER)
            return out.to(_autocast_gpu_dtype()) if (z.is_cuda and torch.is_autocast_enabled()) else out
        return ref_forward(self, z)

    cls.forward = forward
    cls._qd_hip_forward = True
    cls._qd_ref_forward = ref_forward
    return cls
