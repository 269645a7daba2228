"""Host side of the integer engine: freezes quantiser state into packed weights and per-channel
epilogue constants ("plans"), and issues the HIP kernels of libqdiff_hip.so through qdiff.hip.

Everything a kernel needs at run time lives in device tensors built here with a handful of torch
ops when quantiser state changes; the steady-state forward never reads a device value back to the
host, so a whole UNet evaluation can be captured in a HIP graph (qdiff/graph.py).

Integer conventions: DESIGN.md §3 (stored byte a' = code - off, "true zero" z' = zp - off, ...).
"""
import math

import os

import torch

from . import hip
from .hip import Grid, pad16, pad32

# True: every (True, True) module / block runs the reference's fp32 fake-quant simulation instead of the integer
# kernels.  Never set by the product path; bench.py uses it to time the simulation on the GPU (the "fake-quant on
# GPU" denominator of SURVEY.md §8d) and tests use it to compare EMA range tracking with the fused path.
SIMULATE = False


class simulation:
    """`with engine.simulation():` — every (True, True) module runs the reference's fp32 fake-quant arithmetic inside the
    block (calibration is DEFINED on that arithmetic: qdiff/recon.py, qdiff/calibrate.py; bench.py times it as the GPU
    fake-quant denominator)."""

    def __enter__(self):
        global SIMULATE
        self.prev, SIMULATE = SIMULATE, True
        return self

    def __exit__(self, *exc):
        global SIMULATE
        SIMULATE = self.prev
        return False


# ------------------------------------------------------------------------------------------------
# quantiser views
# ------------------------------------------------------------------------------------------------
def act_grid(n_bits, sym):
    """Integer grid of an activation quantiser (reference quant_layer.py:54,84-87)."""
    if n_bits > 8:
        raise hip.HipEngineError(f"{n_bits}-bit activations do not fit the int8 MFMA operand")
    if sym:
        nl = 2 ** (n_bits - 1) - 1
        return Grid(-nl - 1, nl, 0)
    return Grid(0, 2 ** n_bits - 1, 128 if n_bits == 8 else 0)


def _scalar_tensor(v, device):
    if torch.is_tensor(v):
        return v.detach().to(device=device, dtype=torch.float32).reshape(())
    return torch.tensor(float(v), dtype=torch.float32, device=device)


def qparams_of(quantizer, device):
    """Kernel-side parameters of a per-tensor quantiser: the device float[4] {delta, zero_point, rinv, fast} of
    hip.make_qparams (no host sync)."""
    d = _scalar_tensor(quantizer.delta, device)
    z = _scalar_tensor(quantizer.zero_point, device)
    return hip.make_qparams(d, z)


def check_act_zero_point(quantizer, grid):
    """The kernels keep activations as bytes a' = code - off and the "true zero" as z' = zp - off: zp must lie on the
    grid's own code range [qmin, qmax] or z' does not fit int8 and the quantiser kernels would wrap it silently
    (the reference handles any zp in floating point: quant_layer.py:82-88).  One host read per plan build; skipped
    while a HIP graph is being captured (plans are built during the warm-up evaluations)."""
    z = quantizer.zero_point
    if torch.is_tensor(z):
        if z.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        z = float(z.detach().reshape(-1)[0].item())
    if not (grid.qmin <= float(z) <= grid.qmax):
        raise hip.HipEngineError(f"activation zero point {z} lies outside the integer grid [{grid.qmin}, {grid.qmax}]: its stored "
                                 "byte zp - off does not fit int8 (set qdiff.engine.SIMULATE = True for the fp32 simulation)")


def tensor_version(t):
    """In-place version counter of a tensor, or None for an inference tensor (created under torch.inference_mode(): it has no
    counter — `t._version` raises — and may change in place without a trace, so nothing may rely on identity + version for it;
    callers compare such tensors by value or treat them as unseen)."""
    try:
        return t._version
    except RuntimeError:
        return None


def quantizer_key(q):
    """Cheap identity of a quantiser's state: changes whenever delta / zero_point / alpha are
    re-assigned or modified in place (resume_cali_model does both: reference utils.py:397-457)."""
    if q is None:
        return None

    def one(v):
        if torch.is_tensor(v):
            return (id(v), tensor_version(v), v.data_ptr())
        return v
    return (id(q), one(getattr(q, "delta", None)), one(getattr(q, "zero_point", None)),
            one(getattr(q, "alpha", None)), getattr(q, "n_bits", None), getattr(q, "sym", None))


# ------------------------------------------------------------------------------------------------
# weight packing (K2)
# ------------------------------------------------------------------------------------------------
class WeightPack:
    __slots__ = ("wq", "ldk", "wbits", "mode", "segs", "Cout", "taps", "Cin", "tiled", "row_perm")


def _w_levels(q):
    return int(q.n_levels)


def geglu_row_perm(F, device):
    """Row order that interleaves 32-wide tiles of the GEGLU projection's value rows [0,F) and gate
    rows [F,2F) — (v0, g0, v1, g1, ...) — so that one lane of the contraction kernel holds the value and
    the gate of the same output feature (QD_EPI_GEGLU_I8 epilogue)."""
    assert F % 32 == 0
    t = torch.arange(F, device=device).view(F // 32, 1, 32)
    return torch.cat([t, t + F], dim=1).reshape(-1)


def pack_module_weights(weight, quantizers, split, row_perm=None):
    """weight: fp32 [Cout, Cin, *k] on the GPU; quantizers: [wq] or [wq, wq_0] (UniformAffine- or
    AdaRound-like objects with delta/zero_point[/alpha]/n_levels).  row_perm: optional output-row
    permutation — or selection: fewer indices than rows (QuantModule.head_plans) — applied to the weight and to every
    per-row quantity.  Returns a WeightPack."""
    dev = weight.device
    w = weight.detach().float().contiguous()
    if row_perm is not None:
        w = w.index_select(0, row_perm).contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= int(s)
    bounds = [(0, Cin)] if split == 0 else [(0, split), (split, Cin)]
    if len(bounds) != len(quantizers):
        raise hip.HipEngineError("split / quantiser count mismatch")
    levels = _w_levels(quantizers[0])
    zps = [q.zero_point.detach().float().reshape(-1).to(dev) if torch.is_tensor(q.zero_point)
           else torch.full((weight.shape[0],), float(q.zero_point), device=dev) for q in quantizers]
    if row_perm is not None:
        zps = [z.index_select(0, row_perm) for z in zps]
    zall = torch.cat(zps)
    zmin, zmax = int(zall.min().item()), int(zall.max().item())  # one-time host read at pack time
    if any(getattr(q, "sym", False) for q in quantizers):
        # a symmetric weight grid is [-n_levels-1, n_levels] (reference quant_layer.py:84-85); the packers implement the
        # asymmetric [0, 2^b - 1] grid every reference configuration uses (weight_quant_params never set `symmetric`)
        raise hip.HipEngineError("symmetric weight quantisers are not supported by the integer engine (the packers clamp "
                                 "to [0, n_levels-1]); set qdiff.engine.SIMULATE = True for the fp32 simulation")
    if levels > 256 or zmin < -128 or zmax > 255:      # the epilogue's zw * Asum product must fit int32
        raise hip.HipEngineError(f"weight quantiser with {levels} levels / zero points in [{zmin}, {zmax}] does not fit the int8 MFMA operand")
    # MFMA-tile-ordered operands (csrc/igemm_dma.hip): raw nibbles W when the grid fits 4 bits (mode 4), else bytes W-128
    # (mode 8); the zero point goes to the epilogue either way (zw resp. zw-128 times the activation row sums)
    mode = 4 if levels <= 16 else 8
    pk = WeightPack()
    pk.Cout, pk.taps, pk.Cin = Cout, taps, Cin
    pk.tiled = True
    pk.mode, pk.wbits = mode, (4 if mode == 4 else 8)
    kofs, segs = 0, []
    for (c0, c1) in bounds:
        clen = c1 - c0
        segs.append(dict(c0w=c0, clen=clen, clen_pad=pad16(clen), kofs=kofs))
        kofs += pad16(clen)
    pk.ldk = max(pad32(kofs), 32)
    # [kstep][n/32][1 KB | 2 KB], kstep = (segment, tap, 64-channel step)
    ntiles, kstep = (Cout + 31) // 32, 0
    for sg in segs:
        sg["kstep0"] = kstep
        kstep += taps * ((sg["clen_pad"] + 63) // 64)
    pk.wq = torch.zeros(kstep * ntiles * (1024 if mode == 4 else 2048), dtype=torch.uint8, device=dev)
    for sg, q, z in zip(segs, quantizers, zps):
        delta = q.delta.detach().float().reshape(-1).to(dev).contiguous()
        if delta.numel() != weight.shape[0]:
            raise hip.HipEngineError("weight quantiser must be channel-wise (per output channel)")
        alpha = getattr(q, "alpha", None)
        if alpha is not None:
            if getattr(q, "soft_targets", False):
                raise hip.HipEngineError("AdaRound soft targets are a calibration-time mode; the integer engine packs hard rounding only")
            alpha = alpha.detach().to(device=dev, dtype=torch.float32).contiguous()
        if row_perm is not None:
            delta = delta.index_select(0, row_perm).contiguous()
            alpha = alpha.index_select(0, row_perm).contiguous() if alpha is not None else None
        wsum = torch.zeros(Cout, dtype=torch.int32, device=dev)
        (hip.pack_weights_t4 if mode == 4 else hip.pack_weights_t8)(
            w, alpha, delta, z.contiguous(), Cout, Cin, taps, sg["c0w"], sg["clen"], levels,
            pk.wq, sg["kstep0"], (Cout + 31) // 32, wsum)
        sg["wsum"] = wsum
        sg["delta_w"] = delta
        # epilogue-side weight zero point: the stored operand is the raw nibble W (zw = zp) or the byte W-128 (zw = zp-128)
        sg["zw"] = (z.to(torch.int32) - (128 if mode == 8 else 0)).contiguous()
        sg["wzp"] = None
    pk.segs = segs
    pk.row_perm = row_perm
    return pk


_PACK_SEG_TENSORS = ("wsum", "delta_w", "zw", "wzp")
_PACK_SEG_INTS = ("c0w", "clen", "clen_pad", "kofs", "kstep0")


def pack_to_dict(pk, to_cpu=True):
    """WeightPack -> plain dict of (CPU) tensors / ints (what a packed checkpoint stores per layer)."""
    cpu = (lambda t: None if t is None else t.detach().cpu()) if to_cpu else (lambda t: None if t is None else t.detach())
    segs = []
    for sg in pk.segs:
        e = {k: int(sg[k]) for k in _PACK_SEG_INTS if k in sg}
        e.update({k: cpu(sg.get(k)) for k in _PACK_SEG_TENSORS})
        segs.append(e)
    return dict(wq=cpu(pk.wq), ldk=pk.ldk, wbits=pk.wbits, mode=pk.mode, Cout=pk.Cout, taps=pk.taps, Cin=pk.Cin,
                tiled=bool(pk.tiled), row_perm=cpu(pk.row_perm), segs=segs)


def pack_from_dict(d, device):
    """Inverse of pack_to_dict: the tensors go to `device`, nothing is re-quantised."""
    dev = lambda t: None if t is None else t.to(device)
    pk = WeightPack()
    pk.wq, pk.ldk, pk.wbits, pk.mode = dev(d["wq"]), int(d["ldk"]), int(d["wbits"]), int(d["mode"])
    pk.Cout, pk.taps, pk.Cin, pk.tiled, pk.row_perm = int(d["Cout"]), int(d["taps"]), int(d["Cin"]), bool(d["tiled"]), dev(d["row_perm"])
    pk.segs = []
    for e in d["segs"]:
        sg = {k: int(e[k]) for k in _PACK_SEG_INTS if k in e}
        sg.update({k: dev(e.get(k)) for k in _PACK_SEG_TENSORS})
        pk.segs.append(sg)
    return pk


def pack_select_tiles(pk, rows):
    """Sub-pack of the output rows `rows` (index tensor, whole 32-row tiles at tile-aligned positions) of a tile-ordered
    pack: the [kstep][n/32][1 KB | 2 KB] operand is gathered tile by tile, the per-row constants row by row — nothing is
    re-quantised, so this also works on the frozen pack of a packed checkpoint whose fp32 weights are gone."""
    n = int(rows.numel())
    if not pk.tiled or pk.row_perm is not None or n % 32 != 0:
        return None
    r = rows.view(-1, 32)
    if bool((r[:, 0] % 32 != 0).any()) or not torch.equal(r, r[:, :1] + torch.arange(32, device=rows.device)):
        return None
    tile_bytes = 1024 if pk.mode == 4 else 2048
    ntiles = (pk.Cout + 31) // 32
    sub = WeightPack()
    sub.ldk, sub.wbits, sub.mode, sub.taps, sub.Cin, sub.tiled = pk.ldk, pk.wbits, pk.mode, pk.taps, pk.Cin, True
    sub.Cout, sub.row_perm = n, rows
    sub.wq = pk.wq.view(-1, ntiles, tile_bytes).index_select(1, (r[:, 0] // 32).to(pk.wq.device)).reshape(-1).contiguous()
    sub.segs = []
    for sg in pk.segs:
        e = {k: sg[k] for k in _PACK_SEG_INTS if k in sg}
        e.update({k: (None if sg.get(k) is None else sg[k].index_select(0, rows.to(sg[k].device)).contiguous()) for k in _PACK_SEG_TENSORS})
        sub.segs.append(e)
    return sub


# ------------------------------------------------------------------------------------------------
# conv / linear plans (K3/K4)
# ------------------------------------------------------------------------------------------------
class ConvPlan:
    """Frozen launch state of one QuantModule in (weight_quant, act_quant) = (True, True)."""
    __slots__ = ("pack", "kh", "kw", "stride", "pad", "bias", "segs", "grids", "qparams", "Cout", "Cin", "ldx")


def build_conv_plan(pack, act_quantizers, kh, kw, stride, pad, bias):
    dev = pack.wq.device
    plan = ConvPlan()
    plan.pack, plan.kh, plan.kw, plan.stride, plan.pad = pack, kh, kw, stride, pad
    plan.bias = bias.detach().float().contiguous() if bias is not None else None
    if plan.bias is not None and pack.row_perm is not None:
        plan.bias = plan.bias.index_select(0, pack.row_perm).contiguous()
    plan.Cout, plan.Cin = pack.Cout, pack.Cin
    plan.segs, plan.grids, plan.qparams = [], [], []
    for sg, aq in zip(pack.segs, act_quantizers):
        grid = act_grid(aq.n_bits, aq.sym)
        check_act_zero_point(aq, grid)
        qp = qparams_of(aq, dev)
        K = pack.taps * sg["clen_pad"]
        d = dict(c0=sg["kofs"], clen=sg["clen_pad"], kofs=sg["kofs"], scale=(qp[0] * sg["delta_w"]).contiguous(),
                 zw=sg["zw"], wzp=sg["wzp"], zc=None, zfill=None, fill16=None, kstep0=sg.get("kstep0", 0))
        if not aq.sym:
            zprime = (qp[1] - grid.off).round().to(torch.int32)          # device scalar z'
            d["zc"] = (zprime * sg["wsum"]).to(torch.int32).contiguous()
            d["zfill"] = torch.stack([zprime, zprime * K]).to(torch.int32).contiguous()
            d["fill16"] = zprime.to(torch.int8).repeat(16).contiguous()  # source of out-of-image taps (DMA loader)
        plan.segs.append(d)
        plan.grids.append(grid)
        plan.qparams.append(qp)
    # activation rows are laid out segment after segment, each padded to 16 bytes: same offsets as the
    # packed weight rows, so c0 == kofs.
    plan.ldx = pad16(sum(s["clen_pad"] for s in pack.segs))
    return plan


def quantize_rows(x, plan, B, C, S, strides, out=None):
    """Quantise a logical [B][C][S] float tensor into the plan's int8 row layout [B*S][ldx]."""
    if out is None:
        out = torch.empty((B * S, plan.ldx), dtype=torch.int8, device=x.device)
    for sg, d, grid, qp in zip(plan.pack.segs, plan.segs, plan.grids, plan.qparams):
        hip.quantize_act(x, B, C, S, strides, qp, grid, out, plan.ldx, c0=sg["c0w"], clen=sg["clen"], oc0=d["c0"])
    return out


def conv_out_hw(H, W, plan, pad_br=None):
    """Output size for symmetric padding `plan.pad` (or explicit bottom/right padding)."""
    pb = plan.pad if pad_br is None else pad_br
    Ho = (H + plan.pad + pb - plan.kh) // plan.stride + 1
    Wo = (W + plan.pad + pb - plan.kw) // plan.stride + 1
    return Ho, Wo


# Storage type of the activations BETWEEN kernels (the residual stream: convolution / projection outputs, skip buffers).
# fp32 is the parity-first default (the reference's arithmetic); fp16 is the reference scripts' own default precision
# (`--precision autocast`, scripts/txt2img.py:231-236) and halves the traffic of the HBM-bound launches.  Every kernel computes
# in fp32 / exact integers either way; only the stored tensor changes.  QDIFF_STREAM=fp16, or engine.set_stream_dtype().
STREAM_DTYPE = torch.float16 if os.environ.get("QDIFF_STREAM", "fp32").lower() in ("fp16", "half", "float16") else torch.float32


# The stream type in force for the evaluation being issued: STREAM_DTYPE once the whole model runs on the integer path (every
# activation quantiser initialised, no range tracking), fp32 before — the floating-point compositions that initialise the
# quantisers are torch code on fp32 parameters.  Set by QuantModel's forward pre-hook; code that drives kernels directly
# (tests, micro-benchmarks) gets STREAM_DTYPE.
_EFFECTIVE = [None]

# Bumped by every change of a quantisation switch or quantiser state below QuantModel's own entry points
# (QuantModule / BaseQuantBlock.set_quant_state, set_running_stat, a quantiser dropping its initialisation): QuantModel re-validates
# its cached "every layer is on the integer path" verdict when the counter has moved.
STATE_GENERATION = [0]


def bump_state():
    STATE_GENERATION[0] += 1


def stream_dtype():
    return _EFFECTIVE[0] if _EFFECTIVE[0] is not None else STREAM_DTYPE


def set_stream_dtype(dtype):
    global STREAM_DTYPE
    if dtype not in (torch.float32, torch.float16):
        raise ValueError("activation stream dtype must be torch.float32 or torch.float16")
    STREAM_DTYPE = dtype
    _EFFECTIVE[0] = None


class CatSlot:
    """Planned destination of one skip concatenation `th.cat([h, hs.pop()], dim=1)` (openaimodel.py:776, ddim
    diffusion.py:340): the kernel that produces the decoder-side `h` (side 0) and the kernel that produced the encoder-side
    skip tensor (side 1) write their fp32 rows — and the first-level GroupNorm statistics of those rows — straight into
    column ranges of ONE [M][C0 + C1] buffer (`ldo` / `gn_ld` of qd_conv2d_i8), every consumer reads its half through the
    row stride, and the concatenation itself is a view.  A producer that does not take a slot simply ignores it; the
    concatenation then falls back to the copy (quant_block.cat_channels checks adjacency in memory, not this object)."""

    def __init__(self, c_left, c_right):
        self.c = (int(c_left), int(c_right))
        self.buf = None
        self.pbuf = None

    def side(self, i):
        return _SlotSide(self, i)

    def rows(self, i, M, C, device):
        if C != self.c[i]:
            return None
        if self.buf is None:
            self.buf = torch.empty((M, self.c[0] + self.c[1]), dtype=stream_dtype(), device=device)
        if self.buf.shape[0] != M or self.buf.device != device or self.buf.dtype != stream_dtype():
            return None
        c0 = 0 if i == 0 else self.c[0]
        return self.buf[:, c0:c0 + C]

    def part(self, i, B, nchunk, C, device):
        if C != self.c[i]:
            return None
        if self.pbuf is None:
            self.pbuf = torch.empty((B * nchunk, self.c[0] + self.c[1], 2), dtype=torch.float32, device=device)
        if self.pbuf.shape[0] != B * nchunk or self.pbuf.device != device:
            return None
        c0 = 0 if i == 0 else self.c[0]
        return self.pbuf.view(B, nchunk, self.c[0] + self.c[1], 2)[:, :, c0:c0 + C]


class _SlotSide:
    __slots__ = ("slot", "i")

    def __init__(self, slot, i):
        self.slot, self.i = slot, i

    def rows(self, M, C, device):
        return self.slot.rows(self.i, M, C, device)

    def part(self, B, nchunk, C, device):
        return self.slot.part(self.i, B, nchunk, C, device)


def upsample_fold_ok(plan, H, W):
    """qd_conv2d_i8 can run `plan` on the nearest-2x up-sampling of a half-resolution map (qd_conv_desc.upsample2x: the
    replication is folded into the im2col source address of the gather kernel); H x W is the UP-SAMPLED size."""
    return bool(plan.pack.tiled and plan.kh * plan.kw > 1 and plan.stride == 1 and H % 2 == 0 and W % 2 == 0)


def conv_forward(plan, xq, B, H, W, Ho=None, Wo=None, out=None, rowbias=None, residual=None, acc_out=None,
                 out_dtype=None, pad_tl=None, splitk=None, gn_stats=False, slot=None, upsample2x=False):
    """Run K3/K4 on quantised rows xq [B*H*W][ldx].  Returns out [B*Ho*Wo][Cout] (row-major).
    splitk=False forbids the split-K schedule (tests compare it with the default, which lets the library decide).
    gn_stats=True: when the layer is eligible (tile-ordered int4, fp32 out, Ho*Wo % 128 == 0, not a split-K layer) the
    kernel also writes the first level of GroupNorm statistics of its output; they are attached to the returned tensor
    as `out.qd_gn_part` ([B][Ho*Wo/128][Cout][2]) for groupnorm_silu_quant to pick up.
    slot: optional CatSlot side — the output (and its statistics) land in that column range of the concatenation buffer."""
    if Ho is None:
        Ho, Wo = conv_out_hw(H, W, plan)
    M = B * Ho * Wo
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else stream_dtype()
    if residual is not None and acc_out is None and residual.dtype != out_dtype:
        residual = residual.to(out_dtype)                  # the kernel reads the residual in the output's type
    if out is None and acc_out is None:
        if slot is not None and out_dtype == stream_dtype():
            out = slot.rows(M, plan.Cout, xq.device)
        if out is None:
            slot = None
            out = torch.empty((M, plan.Cout), dtype=out_dtype, device=xq.device)
    else:
        slot = None
    pad = plan.pad if pad_tl is None else pad_tl
    call = hip.ConvCall(x=xq, w=plan.pack.wq, out=out, bias=plan.bias, rowbias=rowbias,
                        residual=residual, ldx=plan.ldx, ldk=plan.pack.ldk,
                        ldo=(out.stride(0) if out is not None else 0),
                        ldr=(residual.stride(0) if residual is not None else 0),
                        ld_rowbias=(rowbias.stride(0) if rowbias is not None else 0),
                        B=B, H=H, W=W, Ho=Ho, Wo=Wo, Cout=plan.Cout, kh=plan.kh, kw=plan.kw, stride=plan.stride,
                        pad_t=pad, pad_l=pad, wbits=plan.pack.wbits, w_tiled=plan.pack.tiled, segs=plan.segs,
                        splitk=splitk, upsample2x=upsample2x)
    part = None
    if (gn_stats and acc_out is None and plan.pack.tiled and out.dtype in (torch.float32, torch.float16) and (Ho * Wo) % 128 == 0
            and out.stride(1) == 1 and (splitk is False or hip.splitk_ws_bytes(call) == 0)):
        if slot is not None:
            part = slot.part(B, Ho * Wo // 128, plan.Cout, xq.device)
        if part is None:
            part = torch.empty((B, Ho * Wo // 128, plan.Cout, 2), dtype=torch.float32, device=xq.device)
        call.gn_part = part
    hip.conv2d_i8(call, acc_out=acc_out)
    if part is not None:
        out.qd_gn_part = part
    return out if acc_out is None else acc_out


def conv_forward_geglu(plan, xq, M, next_plan):
    """GEGLU projection (plan packed with geglu_row_perm) with the fused value*gelu(gate) -> quantise
    epilogue: returns the int8 rows [M][next_plan.ldx] that `next_plan` (the FF output Linear) consumes."""
    if not plan.pack.tiled or plan.pack.wbits != 4 or plan.pack.row_perm is None or len(plan.segs) != 1 or len(next_plan.segs) != 1:
        raise hip.HipEngineError("fused GEGLU epilogue needs a tile-ordered int4 projection packed with geglu_row_perm")
    out = torch.empty((M, next_plan.ldx), dtype=torch.int8, device=xq.device)
    call = hip.ConvCall(x=xq, w=plan.pack.wq, out=out, bias=plan.bias, ldx=plan.ldx, ldk=plan.pack.ldk, ldo=next_plan.ldx,
                        B=1, H=1, W=M, Ho=1, Wo=M, Cout=plan.Cout, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                        wbits=plan.pack.wbits, w_tiled=True, segs=plan.segs, epilogue=hip.EPI_GEGLU_I8,
                        oq_params=next_plan.qparams[0], oq_grid=next_plan.grids[0])
    hip.conv2d_i8(call)
    return out


# ------------------------------------------------------------------------------------------------
# producers (K5, K9)
# ------------------------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _part_view(part, B, S, C):
    """Producer statistics `part` ([Bp][n][C][2], possibly a column range of a wider buffer) as [B][S/128][C][2], or None."""
    if part is None or S % 128 != 0 or part.dim() != 4 or part.shape[2] != C or part.shape[0] * part.shape[1] * 128 != B * S:
        return None
    if tuple(part.shape[:2]) == (B, S // 128):
        return part
    try:
        # producers that ran as one [B*S]-row GEMM report [1][B*S/128]: same memory as [B][S/128] when S % 128 == 0
        return part.view(B, S // 128, C, 2)
    except RuntimeError:
        return None


def raw_quant_segs(plan, C):
    """Segment table of qd_raw_quant for `plan` reading a C-channel tensor, or None when the layout is not covered
    (segment bounds must be multiples of 16 channels and tile the C channels)."""
    segs, pos = [], 0
    for sg, d, grid, qp in zip(plan.pack.segs, plan.segs, plan.grids, plan.qparams):
        if sg["clen"] % 16 or sg["c0w"] % 16 or d["c0"] % 16 or sg["c0w"] != pos:
            return None
        segs.append(dict(c0=sg["c0w"], clen=sg["clen"], oc0=d["c0"], qparams=qp, grid=grid))
        pos += sg["clen"]
    return segs if pos == C and 1 <= len(segs) <= 2 else None


def groupnorm_silu_quant(x_rows, B, S, C, gn, silu, plan=None, want_float=False, part=None, raw_plan=None, mod=None):
    """x_rows: channels-last rows [B*S][>=C] (fp32/fp16).  Returns (int8 rows for `plan`, float rows) — and, with
    raw_plan (the ConvPlan of a 1x1 consumer of the SAME un-normalised tensor: a residual block's skip connection), a
    third value: that consumer's int8 input rows, quantised in the same pass over x.
    part: first-level statistics that came with x_rows from its producer (conv_forward(gn_stats=True)).
    mod: [B][2C] fp32 rows scale | shift — GroupNorm(x) * (1 + scale) + shift of a use_scale_shift_norm residual block
    (reference quant_block.py:99-103), folded into the normalisation's affine."""
    part = _part_view(part, B, S, C)
    dev = x_rows.device
    ws = _workspace(hip.groupnorm_ws_bytes(B, C, S), dev)
    out = torch.empty((B * S, plan.ldx), dtype=torch.int8, device=dev) if plan is not None else None
    y = torch.empty((B * S, C), dtype=torch.float32, device=dev) if want_float else None
    if plan is not None and (len(plan.segs) != 1 or plan.pack.segs[0]["clen"] != C):
        raise hip.HipEngineError("groupnorm producer feeds single-segment consumers of the same width only")
    raw, raw_out = None, None
    if raw_plan is not None:
        segs = raw_quant_segs(raw_plan, C)
        if segs is None:
            raise hip.HipEngineError("raw_plan: the consumer's segments must tile the channels in multiples of 16")
        raw_out = torch.empty((B * S, raw_plan.ldx), dtype=torch.int8, device=dev)
        raw = dict(out=raw_out, segs=segs)
    hip.groupnorm_silu_quant(x_rows, B, S, C, x_rows.stride(0), gn.num_groups, gn.eps, gn.weight, gn.bias, silu,
                             plan.qparams[0] if plan is not None else None, plan.grids[0] if plan is not None else None,
                             out, plan.ldx if plan is not None else 0, ws, yout=y, ldy=C, part=part, raw=raw, mod=mod)
    if raw_plan is not None:
        return out, y, raw_out
    return out, y


def layernorm_quant(x_rows, M, C, ln, plans):
    """LayerNorm + one quantised copy per consumer plan (to_q / to_k / to_v have distinct deltas)."""
    outs = [torch.empty((M, p.ldx), dtype=torch.int8, device=x_rows.device) for p in plans]
    ldo = plans[0].ldx
    if any(p.ldx != ldo or len(p.segs) != 1 for p in plans):
        raise hip.HipEngineError("layernorm consumers must share one row layout")
    if C % 16 or C > 1536:                                    # outside qd_layernorm_quant's widths: the library's norm, then one quantiser pass per consumer
        y = torch.nn.functional.layer_norm(x_rows[:, :C].float(), (C,), ln.weight, ln.bias, ln.eps)
        return [quantize_rows(y, p, 1, C, M, (0, 1, y.stride(0))) for p in plans]
    hip.layernorm_quant(x_rows, M, C, x_rows.stride(0), ln.eps, ln.weight, ln.bias, [p.qparams[0] for p in plans],
                        [p.grids[0] for p in plans], outs, ldo)
    return outs


def geglu_quant(h_rows, M, F, plan):
    out = torch.empty((M, plan.ldx), dtype=torch.int8, device=h_rows.device)
    hip.geglu_quant(h_rows, M, F, h_rows.stride(0), plan.qparams[0], plan.grids[0], out, plan.ldx)
    return out


# ------------------------------------------------------------------------------------------------
# attention (K7/K8)
# ------------------------------------------------------------------------------------------------
class AttnPlan:
    __slots__ = ("prm", "grids", "qparams", "wbits", "wmin", "wmax", "asym", "prescale", "scale")


def build_attn_plan(aq_q, aq_k, aq_v, aq_w, scale, prescale, device):
    """aq_*: the four activation quantisers of an attention block (quant_block.py:240-252,345-351).
    scale multiplies the integer scores (d^-1/2 for SD/CIFAR, 1 for LDM where q*s, k*s are quantised)."""
    ap = AttnPlan()
    gq, gk, gv = act_grid(aq_q.n_bits, aq_q.sym), act_grid(aq_k.n_bits, aq_k.sym), act_grid(aq_v.n_bits, aq_v.sym)
    if aq_w.n_bits not in (8, 16) and aq_w.n_bits > 8:
        raise hip.HipEngineError(f"softmax bit-width {aq_w.n_bits} unsupported (<=8 or 16)")
    if aq_w.sym:
        nl = 2 ** (aq_w.n_bits - 1) - 1
        wmin, wmax = -nl - 1, nl
    else:
        wmin, wmax = 0, 2 ** aq_w.n_bits - 1
    ap.wbits = 16 if aq_w.n_bits > 8 else 8
    ap.wmin, ap.wmax = wmin, wmax
    qq, qk, qv, qw = (qparams_of(a, device) for a in (aq_q, aq_k, aq_v, aq_w))
    prm = torch.zeros(16, dtype=torch.float32, device=device)
    prm[0] = qq[0] * qk[0] * float(scale)
    prm[1] = qq[1] - gq.off
    prm[2] = qk[1] - gk.off
    prm[3] = qw[0]
    prm[4] = qw[1]
    prm[5] = qw[0] * qv[0]
    prm[6] = qv[1] - gv.off
    ap.prm = prm
    ap.grids = (gq, gk, gv)
    ap.qparams = (qq, qk, qv)
    ap.asym = not aq_q.sym                 # q has a non-zero stored zero point -> per-key restoration term
    ap.prescale = float(prescale)
    ap.scale = float(scale)
    return ap


def attention(ap, q, k, v, B, T, S, H, d, q_strides, k_strides, v_strides, out=None):
    """q: logical [B][T][H][d] addressed by element strides (sb, st, sh, sd); k, v: [B][S][H][d].
    Returns merged-head rows out[B*T][H*d] fp32."""
    dev = q.device
    Tpad, Spad, dpad = pad32(T), pad32(S), pad32(d)
    BH = B * H
    q8 = torch.empty((BH, Tpad, dpad), dtype=torch.int8, device=dev)
    k8 = torch.empty((BH, Spad, dpad), dtype=torch.int8, device=dev)
    v8 = torch.empty((BH, dpad, Spad), dtype=torch.int8, device=dev)
    # no q/k row sums: per-query zero-point terms cancel in the softmax, the per-key term is restored by the attention
    # kernel (accumulator seeds from hip.attn_keyterm on the d < 64 kernels, constant-operand MFMAs elsewhere)
    vsum = torch.empty((BH, dpad), dtype=torch.int32, device=dev)
    gq, gk, gv = ap.grids
    hip.quantize_heads(q, B, T, H, d, q_strides, ap.prescale, ap.qparams[0], gq, False, q8, None, Tpad, dpad)
    hip.quantize_heads(k, B, S, H, d, k_strides, ap.prescale, ap.qparams[1], gk, False, k8, None, Spad, dpad)
    hip.quantize_heads(v, B, S, H, d, v_strides, 1.0, ap.qparams[2], gv, True, v8, vsum, Spad, dpad)
    if out is None:
        out = torch.empty((B * T, H * d), dtype=torch.float32, device=dev)
    hip.attn_i8(q8, k8, v8, vsum, BH, H, T, S, d, Tpad, Spad, dpad, ap.prm, ap.wbits, ap.wmin, ap.wmax, ap.asym,
                out, out.stride(0))
    return out


_HEAD_BUFS = {}


def head_buffers(device, BH, T, S, d):
    """(q8 [BH][Tpad][dpad], k8 [BH][Spad][dpad], v8^T [BH][dpad][Spad], vsum [BH][dpad]): the int8 operands of
    qd_attn_i8, zero-initialised ONCE and shared by every attention block of that LOGICAL shape on the device
    (stream-ordered reuse: one evaluation runs on one stream; the projection epilogues never write pad bytes, so the
    pads stay zero — two blocks that only agree on the padded sizes, e.g. d = 40 and d = 48, get separate buffers — and
    a captured HIP graph keeps pointing at stable addresses, which is why the key does not include the stream: the
    warm-up evaluations before a capture run on a side stream and must create the buffers the capture then reuses)."""
    Tpad, Spad, dpad = pad32(T), pad32(S), pad32(d)
    key = (device, BH, T, S, d)
    bufs = _HEAD_BUFS.get(key)
    if bufs is None:
        bufs = (torch.zeros((BH, Tpad, dpad), dtype=torch.int8, device=device),
                torch.zeros((BH, Spad, dpad), dtype=torch.int8, device=device),
                torch.zeros((BH, dpad, Spad), dtype=torch.int8, device=device),
                torch.zeros((BH, dpad), dtype=torch.int32, device=device))
        _HEAD_BUFS[key] = bufs
    return bufs


# V^T column sums (hd_sum of QD_EPI_HEADS_T_I8) are ACCUMULATED by the projection epilogue, so they must be zero when it
# starts.  One memset per attention block per evaluation (16 tiny launches in SD) becomes one: every block owns a slice of
# a per-device arena that QuantModel's forward pre-hook zeroes once per UNet evaluation (begin_evaluation); a block whose
# slice is not known to be clean — called outside a QuantModel evaluation, or twice in one — zeroes it itself.
_VSUM = {}
_VSUM_ARENA_INTS = 1 << 20
_VSUM_OWNER = [None]          # the QuantModel whose evaluation is being issued (set by begin_evaluation)


def vsum_slice(owner, device, shape):
    """int32 tensor of `shape` for `owner` (any hashable), carved out of the arena of (device, model under evaluation).
    One arena per MODEL: two QuantModels evaluating on different streams of one device (a capture warm-up next to a live
    model) never zero or mark clean each other's in-flight column sums."""
    mkey = (device, _VSUM_OWNER[0])
    st = _VSUM.get(mkey)
    if st is None:
        st = _VSUM[mkey] = dict(arena=torch.zeros(_VSUM_ARENA_INTS, dtype=torch.int32, device=device), used=0, views={}, clean=set())
    key = (owner, tuple(shape))
    v = st["views"].get(key)
    if v is None:
        n = 1
        for e in shape:
            n *= e
        n_pad = (n + 63) // 64 * 64
        if st["used"] + n_pad > st["arena"].numel():
            v = torch.zeros(shape, dtype=torch.int32, device=device)         # arena exhausted: a buffer of its own
        else:
            v = st["arena"][st["used"]:st["used"] + n].view(shape)
            st["used"] += n_pad
            st["clean"].add(id(v))                                            # the arena starts zeroed
        st["views"][key] = v
    return v


def begin_evaluation(model_key=None):
    """Start of a UNet evaluation (QuantModel forward pre-hook): one memset for all attention blocks' V^T column sums of
    THIS model (`model_key`: id of the QuantModel; None = the anonymous arena of code that drives blocks directly)."""
    _VSUM_OWNER[0] = model_key
    for (dev, mk), st in _VSUM.items():
        if mk == model_key and st["used"]:
            st["arena"][:st["used"]].zero_()
            st["clean"] = {id(v) for v in st["views"].values() if v.untyped_storage().data_ptr() == st["arena"].untyped_storage().data_ptr()}


def release_model(model_key):
    """Drop the V-sum arena(s) of a QuantModel that is gone (weakref.finalize of the model): its 4 MB device arena and the
    bookkeeping that a recycled id() would otherwise inherit."""
    for k in [k for k in _VSUM if k[1] == model_key]:
        del _VSUM[k]
    if _VSUM_OWNER[0] == model_key:
        _VSUM_OWNER[0] = None


def _vsum_prepare(vsum):
    st = _VSUM.get((vsum.device, _VSUM_OWNER[0]))
    if st is not None and id(vsum) in st["clean"]:
        st["clean"].discard(id(vsum))
        return
    vsum.zero_()


def heads_fusable(plan, T, H):
    """The projection `plan` can write its output directly as attention operand bytes (QD_EPI_HEADS_*)."""
    # int4 weights on every tile width; int8 weights (CIFAR W8A8) on the 128-wide tile, i.e. more than 64 output channels
    return bool(plan.pack.tiled and (plan.pack.wbits == 4 or (plan.pack.wbits == 8 and plan.Cout > 64)) and len(plan.segs) == 1
                and T % 128 == 0 and plan.Cout % H == 0 and (plan.Cout // H) % 4 == 0)


def project_heads(plan, xq, B, T, H, ap, which, out8, vsum=None):
    """q (which=0) / k (1) / v (2) projection of B*T token rows `xq`, quantised with the attention block's
    own act quantiser inside the GEMM epilogue and stored in the operand layout of the attention kernel."""
    hip.conv2d_i8(_heads_call(plan, xq, B, T, H, ap, which, out8, vsum))


def project_heads_group(members, B, T, H, ap, vsum=None):
    """The q / k / v projections of one attention block — members: [(plan, xq, which, out8), ...], all heads_fusable, on the
    same B*T rows — as ONE launch where the library can group them (hip.conv2d_i8_group), else one launch each: same bytes."""
    hip.conv2d_i8_group([_heads_call(plan, xq, B, T, H, ap, which, out8, vsum) for plan, xq, which, out8 in members])


def _heads_call(plan, xq, B, T, H, ap, which, out8, vsum):
    d = plan.Cout // H
    if which == 2:
        _vsum_prepare(vsum)
    return hip.ConvCall(x=xq, w=plan.pack.wq, out=out8, bias=plan.bias, ldx=plan.ldx, ldk=plan.pack.ldk, ldo=0,
                        B=B, H=1, W=T, Ho=1, Wo=T, Cout=plan.Cout, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                        wbits=plan.pack.wbits, w_tiled=True, segs=plan.segs,
                        epilogue=hip.EPI_HEADS_T_I8 if which == 2 else hip.EPI_HEADS_I8,
                        oq_params=ap.qparams[which], oq_grid=ap.grids[which],
                        heads=dict(H=H, d=d, T=T, Tpad=pad32(T), dpad=pad32(d), sum=vsum,
                                   prescale=ap.prescale if which < 2 else 1.0))


def rows_i8_fusable(plan, next_plan, T):
    """`plan`'s output (+ residual) can be written as `next_plan`'s int8 input rows by the GEMM epilogue."""
    return bool(plan.pack.tiled and plan.pack.wbits == 4 and len(plan.segs) == 1 and len(next_plan.segs) == 1 and T % 128 == 0
                and plan.Cout % 32 == 0 and next_plan.ldx == plan.Cout)


def linear_to_rows_i8(plan, xq, B, T, next_plan, residual=None):
    """Linear on B*T token rows whose only consumer is `next_plan`: out = quantise_next(I*scale + bias + residual), int8
    rows [B*T][next_plan.ldx] (QD_EPI_HEADS_I8 with one "head" as wide as the layer: plain row-major bytes)."""
    out8 = torch.empty((B * T, next_plan.ldx), dtype=torch.int8, device=xq.device)
    call = hip.ConvCall(x=xq, w=plan.pack.wq, out=out8, bias=plan.bias, residual=residual, ldx=plan.ldx, ldk=plan.pack.ldk, ldo=0,
                        ldr=(residual.stride(0) if residual is not None else 0),
                        B=B, H=1, W=T, Ho=1, Wo=T, Cout=plan.Cout, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                        wbits=plan.pack.wbits, w_tiled=True, segs=plan.segs, epilogue=hip.EPI_HEADS_I8,
                        oq_params=next_plan.qparams[0], oq_grid=next_plan.grids[0],
                        heads=dict(H=1, d=plan.Cout, T=T, Tpad=T, dpad=plan.Cout, prescale=1.0, sum=None))
    hip.conv2d_i8(call)
    return out8


def heads_from_float(ap, which, x, B, T, H, d, strides, out8, vsum=None):
    """Same operand bytes from an fp32 projection output (the unfused route: ragged token counts, context k/v)."""
    hip.quantize_heads(x, B, T, H, d, strides, ap.prescale if which < 2 else 1.0, ap.qparams[which], ap.grids[which],
                       which == 2, out8, vsum if which == 2 else None, pad32(T), pad32(d))


def attention_codes(ap, q8, k8, v8, vsum, B, T, S, H, d, out=None, out_plan=None, kterm=None):
    """Fused quantised attention on prepared operand bytes; returns merged-head rows out[B*T][H*d] fp32 —
    or, with out_plan (the ConvPlan of the Linear that consumes the output, one segment, input width H*d), that
    Linear's int8 input rows [B*T][out_plan.ldx], quantised in the attention epilogue.
    kterm: the key-term table of THIS k8 (hip.attn_keyterm) when the caller keeps one (static keys); else hip.attn_i8
    builds it per call where the head dim takes one."""
    if out_plan is not None:
        if len(out_plan.segs) != 1 or out_plan.ldx != H * d:
            raise hip.HipEngineError("attention_codes: out_plan must take exactly the H*d merged-head features")
        out8 = torch.empty((B * T, out_plan.ldx), dtype=torch.int8, device=q8.device)
        hip.attn_i8(q8, k8, v8, vsum, B * H, H, T, S, d, pad32(T), pad32(S), pad32(d), ap.prm, ap.wbits, ap.wmin, ap.wmax,
                    ap.asym, None, 0, out8=out8, oq_params=out_plan.qparams[0], oq_grid=out_plan.grids[0], kterm=kterm)
        return out8
    if out is None:
        out = torch.empty((B * T, H * d), dtype=torch.float32, device=q8.device)
    hip.attn_i8(q8, k8, v8, vsum, B * H, H, T, S, d, pad32(T), pad32(S), pad32(d), ap.prm, ap.wbits, ap.wmin, ap.wmax,
                ap.asym, out, out.stride(0), kterm=kterm)
    return out


# ------------------------------------------------------------------------------------------------
# standalone attention matmuls (QuantQKMatMul / QuantSMVMatMul used on their own)
# ------------------------------------------------------------------------------------------------
def _softmax_grid(aq_w):
    if aq_w.n_bits not in (8, 16) and aq_w.n_bits > 8:
        raise hip.HipEngineError(f"softmax bit-width {aq_w.n_bits} unsupported (<=8 or 16)")
    if aq_w.sym:
        nl = 2 ** (aq_w.n_bits - 1) - 1
        return (16 if aq_w.n_bits > 8 else 8), -nl - 1, nl
    return (16 if aq_w.n_bits > 8 else 8), 0, 2 ** aq_w.n_bits - 1


def qk_matmul_int(aq_q, aq_k, q, k, scale):
    """QuantQKMatMul.forward on the integer engine (reference quant_block.py:123-134): q [BH][c][T], k [BH][c][S]
    fp32 ("bct"); q*scale and k*scale are quantised, the contraction is exact int32, the T x S scores are returned
    as fp32 [BH][T][S] (the API materialises them: the caller applies the softmax)."""
    BH, c, T = q.shape
    S = k.shape[2]
    dev = q.device
    gq, gk = act_grid(aq_q.n_bits, aq_q.sym), act_grid(aq_k.n_bits, aq_k.sym)
    pq, pk = qparams_of(aq_q, dev), qparams_of(aq_k, dev)
    prm = torch.zeros(8, dtype=torch.float32, device=dev)
    prm[0], prm[1], prm[2] = pq[0] * pk[0], pq[1] - gq.off, pk[1] - gk.off
    Tpad, Spad, dpad = pad32(T), pad32(S), pad32(c)
    q8 = torch.empty((BH, Tpad, dpad), dtype=torch.int8, device=dev)
    k8 = torch.empty((BH, Spad, dpad), dtype=torch.int8, device=dev)
    hip.quantize_heads(q, BH, T, 1, c, (q.stride(0), q.stride(2), 0, q.stride(1)), float(scale), pq, gq, False, q8, None, Tpad, dpad)
    hip.quantize_heads(k, BH, S, 1, c, (k.stride(0), k.stride(2), 0, k.stride(1)), float(scale), pk, gk, False, k8, None, Spad, dpad)
    out = torch.empty((BH, T, S), dtype=torch.float32, device=dev)
    hip.bmm_qk_i8(q8, k8, BH, T, S, c, Tpad, Spad, dpad, prm, out)
    return out


def smv_matmul_int(aq_w, aq_v, weight, v):
    """QuantSMVMatMul.forward on the integer engine (reference quant_block.py:152-157): weight [BH][T][S] fp32
    probabilities, v [BH][c][S]; returns [BH][c][T] fp32."""
    BH, T, S = weight.shape
    c = v.shape[1]
    dev = weight.device
    gv = act_grid(aq_v.n_bits, aq_v.sym)
    pv, pw = qparams_of(aq_v, dev), qparams_of(aq_w, dev)
    wbits, wmin, wmax = _softmax_grid(aq_w)
    prm = torch.zeros(8, dtype=torch.float32, device=dev)
    prm[3], prm[4], prm[5], prm[6] = pw[0], pw[1], pw[0] * pv[0], pv[1] - gv.off
    Spad, dpad = pad32(S), pad32(c)
    v8 = torch.empty((BH, dpad, Spad), dtype=torch.int8, device=dev)
    vsum = torch.empty((BH, dpad), dtype=torch.int32, device=dev)
    hip.quantize_heads(v, BH, S, 1, c, (v.stride(0), v.stride(2), 0, v.stride(1)), 1.0, pv, gv, True, v8, vsum, Spad, dpad)
    if weight.stride(2) != 1:
        weight = weight.contiguous()
    out = torch.empty((BH, c, T), dtype=torch.float32, device=dev)
    hip.bmm_pv_i8(weight.float(), v8, vsum, BH, T, S, c, Spad, dpad, prm, wbits, wmin, wmax, out)
    return out


def sinusoid(timesteps, dim, flavour):
    """Timestep sinusoid table (K6 front half; tiny, kept in torch).  flavour 'ldm': cos|sin with
    /half (ldm util.py:151-171); 'ddim': sin|cos with /(half-1) (ddim diffusion.py:6-24)."""
    half = dim // 2
    ar = torch.arange(half, dtype=torch.float32, device=timesteps.device)
    if flavour == "ldm":
        freqs = torch.exp(-math.log(10000) * ar / half)
        args = timesteps[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    else:
        freqs = torch.exp(ar * -(math.log(10000) / (half - 1)))
        args = timesteps.float()[:, None] * freqs[None, :]
        emb = torch.cat([torch.sin(args), torch.cos(args)], dim=1)
    if dim % 2:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb
