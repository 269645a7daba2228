"""ctypes binding of libqdiff_hip.so (include/qdiff_hip.h) for torch tensors.

This module is the only place where device pointers cross from PyTorch (used for HBM allocation,
streams and torch.distributed only) into the HIP engine.  There is no fallback: if the shared
library is missing or a tensor is not on the GPU, calls raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# QDIFF_HIP_LIB: an alternative build of the same library (q-diffusion_amd/build.py --variant ...), for A/B measurements
LIB_PATH = os.environ.get("QDIFF_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libqdiff_hip.so")

F32, F16, BF16 = 0, 1, 2
_DT = {torch.float32: F32, torch.float16: F16}


class HipEngineError(RuntimeError):
    pass


class ConvSeg(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int32), ("clen", ctypes.c_int32), ("kofs", ctypes.c_int32), ("kstep0", ctypes.c_int32),
                ("wzp", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("zc", ctypes.c_void_p), ("zw", ctypes.c_void_p), ("zfill", ctypes.c_void_p),
                ("fill16", ctypes.c_void_p)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("rowbias", ctypes.c_void_p), ("residual", ctypes.c_void_p),
                ("ldx", ctypes.c_int64), ("ldk", ctypes.c_int64), ("ldo", ctypes.c_int64), ("ldr", ctypes.c_int64),
                ("ld_rowbias", ctypes.c_int64),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Ho", ctypes.c_int32),
                ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad_t", ctypes.c_int32),
                ("pad_l", ctypes.c_int32),
                ("wbits", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("nseg", ctypes.c_int32),
                ("w_tiled", ctypes.c_int32), ("epilogue", ctypes.c_int32),
                ("seg", ConvSeg * 2),
                ("oq_params", ctypes.c_void_p), ("oq_min", ctypes.c_int32), ("oq_max", ctypes.c_int32),
                ("oq_off", ctypes.c_int32), ("_pad2", ctypes.c_int32),
                ("splitk_ws", ctypes.c_void_p), ("splitk_ws_bytes", ctypes.c_int64),
                ("hd_H", ctypes.c_int32), ("hd_d", ctypes.c_int32), ("hd_T", ctypes.c_int32), ("hd_Tpad", ctypes.c_int32),
                ("hd_dpad", ctypes.c_int32), ("oq_prescale", ctypes.c_float), ("hd_sum", ctypes.c_void_p),
                ("gn_part", ctypes.c_void_p), ("gn_ld", ctypes.c_int64), ("upsample2x", ctypes.c_int32), ("_pad3", ctypes.c_int32)]


class RawSeg(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int32), ("clen", ctypes.c_int32), ("oc0", ctypes.c_int32),
                ("qmin", ctypes.c_int32), ("qmax", ctypes.c_int32), ("off", ctypes.c_int32), ("qparams", ctypes.c_void_p)]


class RawQuant(ctypes.Structure):
    _fields_ = [("out", ctypes.c_void_p), ("ldo", ctypes.c_int64), ("nseg", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("seg", RawSeg * 2)]


EPI_LINEAR, EPI_GEGLU_I8, EPI_HEADS_I8, EPI_HEADS_T_I8 = 0, 1, 2, 3


EXPORTS = ["qd_abi_version", "qd_last_error", "qd_device_ok", "qd_box_probe", "qd_make_qparams", "qd_quantize_act", "qd_pack_weights", "qd_pack_weights_t4",
           "qd_pack_weights_t8",
           "qd_conv2d_i8", "qd_conv_config", "qd_conv2d_i8_group", "qd_conv2d_i8_splitk_ws_bytes",
           "qd_conv2d_i8_acc", "qd_groupnorm_ws_bytes", "qd_groupnorm_silu_quant", "qd_groupnorm_mod_silu_quant", "qd_layernorm_quant",
           "qd_geglu_quant", "qd_quantize_heads", "qd_attn_i8", "qd_attn_keyterm", "qd_attn_uses_keyterm", "qd_attn_config", "qd_attn_ws_bytes", "qd_bmm_qk_i8", "qd_bmm_pv_i8", "qd_temb_mlp",
           "qd_fakequant_blocks", "qd_fakequant_fwd", "qd_fakequant_bwd",
           "qd_conv2d_bf16", "qd_pack_weights_bf16_bytes", "qd_pack_weights_bf16", "qd_groupnorm_silu_bf16",
           "qd_pack_weights_h16", "qd_groupnorm_silu_h16"]

_lib = None


def load():
    """Load the shared library (once).  Raises HipEngineError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipEngineError(f"{LIB_PATH} not found: build it with `python q-diffusion_amd/build.py` "
                             "(or __graft_entry__.build()); the quantised path has no fallback")
    lib = ctypes.CDLL(LIB_PATH)
    lib.qd_last_error.restype = ctypes.c_char_p
    lib.qd_groupnorm_ws_bytes.restype = ctypes.c_int64
    lib.qd_groupnorm_ws_bytes.argtypes = [ctypes.c_int64] * 3
    i64, i32, vp, f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float
    lib.qd_make_qparams.argtypes = [vp, vp, vp, vp]
    lib.qd_quantize_act.argtypes = [vp, i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, vp, i32, i32, i32, vp, i64, i32, vp]
    lib.qd_pack_weights.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp, vp]
    lib.qd_pack_weights_t4.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]
    lib.qd_pack_weights_t8.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp]
    lib.qd_conv2d_i8.argtypes = [ctypes.POINTER(ConvDesc), vp]
    lib.qd_conv_config.argtypes = [i32]
    lib.qd_conv_config.restype = None
    lib.qd_conv2d_i8_group.argtypes = [ctypes.POINTER(ctypes.POINTER(ConvDesc)), i32, vp]
    lib.qd_conv2d_i8_acc.argtypes = [ctypes.POINTER(ConvDesc), vp, vp]
    lib.qd_conv2d_i8_splitk_ws_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.qd_conv2d_i8_splitk_ws_bytes.restype = ctypes.c_int64
    lib.qd_groupnorm_silu_quant.argtypes = [vp, i32, i64, i64, i32, i64, i32, f32, vp, vp, i32, vp, i32, i32, i32, vp,
                                            i64, vp, i64, vp, vp, i32, i64, ctypes.POINTER(RawQuant), vp]
    lib.qd_groupnorm_mod_silu_quant.argtypes = [vp, i32, i64, i64, i32, i64, i32, f32, vp, vp, i32, vp, i32, i32, i32, vp,
                                                i64, vp, i64, vp, vp, i32, i64, vp, i64, vp]
    lib.qd_layernorm_quant.argtypes = [vp, i32, i64, i32, i64, f32, vp, vp, i32, ctypes.POINTER(vp),
                                       ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                       ctypes.POINTER(vp), i64, vp]
    lib.qd_geglu_quant.argtypes = [vp, i32, i64, i32, i64, vp, i32, i32, i32, vp, i64, vp]
    lib.qd_quantize_heads.argtypes = [vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, f32, vp, i32, i32, i32, i32,
                                      vp, vp, i32, i32, vp]
    lib.qd_attn_i8.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp,
                               i64, vp, i64, vp, i32, i32, i32, vp, i64, vp]
    lib.qd_attn_keyterm.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.qd_attn_uses_keyterm.argtypes = [i32, i32, i32]
    lib.qd_attn_config.argtypes = [i32, i32, i32, i32]
    lib.qd_attn_config.restype = None
    lib.qd_temb_mlp.argtypes = [vp, i64, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, i64, vp]
    lib.qd_bmm_qk_i8.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i64, i64, vp]
    lib.qd_bmm_pv_i8.argtypes = [vp, i64, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i64, i64, vp]
    lib.qd_fakequant_blocks.restype = ctypes.c_int64
    lib.qd_fakequant_blocks.argtypes = [i64]
    lib.qd_fakequant_fwd.argtypes = [vp, i64, vp, vp, i32, i32, vp, vp]
    lib.qd_fakequant_bwd.argtypes = [vp, vp, i64, vp, vp, i32, i32, vp, vp, vp]
    lib.qd_conv2d_bf16.argtypes = [ctypes.POINTER(ConvDesc), vp]
    lib.qd_pack_weights_bf16_bytes.restype = ctypes.c_int64
    lib.qd_pack_weights_bf16_bytes.argtypes = [i32, i32, i32]
    lib.qd_pack_weights_bf16.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.qd_groupnorm_silu_bf16.argtypes = [vp, i64, i64, i32, i64, i32, f32, vp, vp, i32, vp, i64, vp, vp, i32, i64, vp]
    lib.qd_pack_weights_h16.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.qd_groupnorm_silu_h16.argtypes = [vp, i64, i64, i32, i64, i32, f32, vp, vp, i32, i32, vp, i64, vp, vp, i32, i64, vp]
    lib.qd_box_probe.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.qd_attn_ws_bytes.argtypes = [i32, i32, i32, i32]
    lib.qd_attn_ws_bytes.restype = ctypes.c_int64
    if lib.qd_abi_version() != 20:
        raise HipEngineError("libqdiff_hip.so ABI version mismatch")
    _lib = lib
    return lib


def available():
    try:
        return bool(load().qd_device_ok())
    except (HipEngineError, OSError):
        return False


def _check(rc, what):
    if rc != 0:
        raise HipEngineError(f"{what} failed ({rc}): {load().qd_last_error().decode()}")


def _ptr(t, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipEngineError(f"{name} must live in GPU memory (got device={t.device}); the integer engine has no host path")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dtype(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise HipEngineError(f"unsupported dtype {t.dtype}")


class Grid:
    """Integer grid of a quantiser: codes in [qmin, qmax], stored as code - off."""
    __slots__ = ("qmin", "qmax", "off")

    def __init__(self, qmin, qmax, off):
        self.qmin, self.qmax, self.off = int(qmin), int(qmax), int(off)


def pad16(n):
    return (n + 15) // 16 * 16


def pad32(n):
    return (n + 31) // 32 * 32


def make_qparams(delta, zero_point):
    """{delta, zero_point} (0-dim / 1-element float32 device tensors) -> the float[4] the kernels read
    ({delta, zero_point, rinv, fast}: include/qdiff_hip.h, qd_make_qparams).  No host synchronisation."""
    d = delta.detach().reshape(1).float().contiguous()
    z = zero_point.detach().reshape(1).float().contiguous()
    out = torch.empty(4, dtype=torch.float32, device=d.device)
    _check(load().qd_make_qparams(_ptr(d, "delta"), _ptr(z, "zero_point"), _ptr(out), _stream()), "qd_make_qparams")
    return out


def _qp(t):
    """Kernel-side quantiser parameters: float[4]; a legacy {delta, zero_point} pair is completed on the fly."""
    if t is None or t.numel() == 4:
        return t
    if t.numel() != 2:
        raise HipEngineError("qparams must be {delta, zero_point} or the float[4] of make_qparams")
    return make_qparams(t[0], t[1])


def quantize_act(x, B, C, S, strides, qparams, grid, out, ldo, c0=0, clen=None, oc0=0):
    """x: any float tensor addressed as logical [B][C][S] by element strides (sb, sc, ss)."""
    clen = C - c0 if clen is None else clen
    sb, sc, ss = strides
    _check(load().qd_quantize_act(_ptr(x, "x"), _dtype(x), B, C, S, sb, sc, ss, c0, clen, pad16(clen),
                                  _ptr(_qp(qparams), "qparams"), grid.qmin, grid.qmax, grid.off, _ptr(out, "out"), ldo, oc0,
                                  _stream()), "qd_quantize_act")


def pack_weights(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, mode, wq, ldk, kofs, wsum, codes=None):
    _check(load().qd_pack_weights(_ptr(w), _ptr(alpha), _ptr(delta), _ptr(zp), Cout, Cin_total, taps, c0, clen,
                                  pad16(clen), n_levels, mode, _ptr(wq), ldk, kofs, _ptr(wsum), _ptr(codes), _stream()),
           "qd_pack_weights")


def pack_weights_t4(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, wt, kstep0, ntiles, wsum):
    _check(load().qd_pack_weights_t4(_ptr(w), _ptr(alpha), _ptr(delta), _ptr(zp), Cout, Cin_total, taps, c0, clen,
                                     pad16(clen), n_levels, _ptr(wt), kstep0, ntiles, _ptr(wsum), _stream()),
           "qd_pack_weights_t4")


def pack_weights_t8(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, wt, kstep0, ntiles, wsum):
    _check(load().qd_pack_weights_t8(_ptr(w), _ptr(alpha), _ptr(delta), _ptr(zp), Cout, Cin_total, taps, c0, clen,
                                     pad16(clen), n_levels, _ptr(wt), kstep0, ntiles, _ptr(wsum), _stream()),
           "qd_pack_weights_t8")


class ConvCall:
    """Python-side description of one qd_conv2d_i8 launch (tensors, not pointers)."""
    __slots__ = ("x", "w", "out", "bias", "rowbias", "residual", "ldx", "ldk", "ldo", "ldr", "ld_rowbias",
                 "B", "H", "W", "Ho", "Wo", "Cout", "kh", "kw", "stride", "pad_t", "pad_l", "wbits", "w_tiled", "segs",
                 "epilogue", "oq_params", "oq_grid", "splitk", "heads", "gn_part", "upsample2x", "_keep")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


_SPLITK_WS = {}          # (device, stream) -> list of scratch buffers, newest last; old ones stay alive for captured graphs


def _splitk_scratch(device, nbytes):
    """Stream-ordered scratch shared by every split-K launch on one stream of `device` (grows, never shrinks or moves: a
    HIP graph captured earlier keeps pointing at the buffer it was captured with; launches on another stream — the
    context K/V branch of qdiff.quant_block.ContextKV — get their own)."""
    bufs = _SPLITK_WS.setdefault((device, _stream()), [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device))
    return bufs[-1]


def splitk_ws_bytes(c):
    """Scratch bytes the library would use to contract ConvCall `c` split-K (0 = launched unsplit)."""
    return int(load().qd_conv2d_i8_splitk_ws_bytes(ctypes.byref(_conv_desc(c))))


def conv2d_i8(c, acc_out=None):
    """c: ConvCall.  segs: list of dicts {c0, clen, kofs, wzp, scale, zc, zw, zfill} (tensors or None)."""
    d = _conv_desc(c)
    if acc_out is None:
        if c.w_tiled and c.splitk is not False:
            need = int(load().qd_conv2d_i8_splitk_ws_bytes(ctypes.byref(d)))
            if need:
                ws = _splitk_scratch(c.x.device, need)
                d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        _check(load().qd_conv2d_i8(ctypes.byref(d), _stream()), "qd_conv2d_i8")
    else:
        _check(load().qd_conv2d_i8_acc(ctypes.byref(d), _ptr(acc_out), _stream()), "qd_conv2d_i8_acc")


def conv_config(kgroups=-1):
    load().qd_conv_config(int(kgroups))


def conv2d_i8_group(calls):
    """Up to three ConvCalls of one shape with head-layout epilogues (the q / k / v projections of an attention block) as ONE
    launch (qd_conv2d_i8_group); the library falls back to single launches for members that do not qualify."""
    descs = [_conv_desc(c) for c in calls]
    arr = (ctypes.POINTER(ConvDesc) * len(descs))(*[ctypes.pointer(d) for d in descs])
    _check(load().qd_conv2d_i8_group(arr, len(descs), _stream()), "qd_conv2d_i8_group")


def _conv_desc(c):
    d = ConvDesc()
    d.x, d.w = _ptr(c.x, "x"), _ptr(c.w, "w")
    d.out = _ptr(c.out, "out")
    d.bias, d.rowbias, d.residual = _ptr(c.bias, "bias"), _ptr(c.rowbias, "rowbias"), _ptr(c.residual, "residual")
    d.ldx, d.ldk, d.ldo = c.ldx, c.ldk, c.ldo or 0
    d.ldr, d.ld_rowbias = c.ldr or 0, c.ld_rowbias or 0
    d.B, d.H, d.W, d.Ho, d.Wo, d.Cout = c.B, c.H, c.W, c.Ho, c.Wo, c.Cout
    d.kh, d.kw, d.stride, d.pad_t, d.pad_l = c.kh, c.kw, c.stride, c.pad_t, c.pad_l
    d.wbits = c.wbits
    d.w_tiled = 1 if c.w_tiled else 0
    d.epilogue = c.epilogue or EPI_LINEAR
    if d.epilogue != EPI_LINEAR:
        c.oq_params = _qp(c.oq_params)              # kept on the call object: the pointer must outlive the launch
        d.oq_params = _ptr(c.oq_params, "oq_params")
        d.oq_min, d.oq_max, d.oq_off = c.oq_grid.qmin, c.oq_grid.qmax, c.oq_grid.off
        # the output is int8; with a residual (QD_EPI_HEADS_I8) out_dtype names the RESIDUAL's type (fp32 / fp16 stream)
        d.out_dtype = _dtype(c.residual) if c.residual is not None else F32
        if d.epilogue in (EPI_HEADS_I8, EPI_HEADS_T_I8):
            hd = c.heads                            # dict(H, d, T, Tpad, dpad, prescale, sum)
            d.hd_H, d.hd_d, d.hd_T, d.hd_Tpad, d.hd_dpad = hd["H"], hd["d"], hd["T"], hd["Tpad"], hd["dpad"]
            d.oq_prescale = float(hd["prescale"])
            d.hd_sum = _ptr(hd.get("sum"), "hd_sum")
    else:
        d.out_dtype = _dtype(c.out) if c.out is not None else F32
    d.gn_part = _ptr(c.gn_part, "gn_part")
    if c.gn_part is not None:
        d.gn_ld = part_ld(c.gn_part)
    d.upsample2x = 1 if c.upsample2x else 0
    d.nseg = len(c.segs)
    for i, s in enumerate(c.segs):
        g = d.seg[i]
        g.c0, g.clen, g.kofs, g.kstep0 = s["c0"], s["clen"], s["kofs"], s.get("kstep0", 0)
        g.wzp, g.fill16 = _ptr(s.get("wzp"), "wzp"), _ptr(s.get("fill16"), "fill16")
        g.scale, g.zc, g.zw, g.zfill = _ptr(s["scale"], "scale"), _ptr(s.get("zc")), _ptr(s.get("zw")), _ptr(s.get("zfill"))
    return d


# ---- first-stage decoder: bf16 mode of the convolution kernel (include/qdiff_hip.h, "First-stage decoder") ----

def pad8(n):
    return (n + 7) // 8 * 8


def pack_weights_bf16(w, dtype=torch.bfloat16):
    """fp32 [Cout][Cin][kh][kw] (or [Cout][Cin]) -> tile-ordered bf16 (or fp16) bytes for qd_conv2d_bf16 (Cin padded to 8)."""
    if dtype not in (torch.bfloat16, torch.float16):
        raise HipEngineError("pack_weights_bf16: operand type must be bfloat16 or float16")
    w = w.detach().float().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    cpad = pad8(Cin)
    wt = torch.zeros(int(load().qd_pack_weights_bf16_bytes(Cout, taps, cpad)), dtype=torch.uint8, device=w.device)
    _check(load().qd_pack_weights_h16(_ptr(w, "w"), Cout, Cin, taps, cpad, 17 if dtype == torch.float16 else 16, _ptr(wt), _stream()),
           "qd_pack_weights_h16")
    return wt


def conv2d_bf16(x, wt, bias, out, B, H, W, Cin_pad, Cout, k=3, pad=1, residual=None, gn_part=None, upsample2x=False):
    """x: bf16 (or fp16: the weights must have been packed with the same type) rows [B*Hin*Win][ldx] (Hin = H/2 when
    upsample2x, the kernel folds the nearest-2x copy into its gather); out: fp32 rows or rows of x's type [B*H*W][ldo];
    residual: rows of the output's type; stride-1 'same' convolution k x k."""
    if x.dtype not in (torch.bfloat16, torch.float16) or out.dtype not in (torch.float32, x.dtype):
        raise HipEngineError("conv2d_bf16: x must be bf16 / fp16, out fp32 or x's type")
    if residual is not None and residual.dtype != out.dtype:
        raise HipEngineError("conv2d_bf16: the residual has the type of the output")
    d = ConvDesc()
    d.x, d.w, d.out = _ptr(x, "x"), _ptr(wt, "w"), _ptr(out, "out")
    d.bias, d.residual = _ptr(bias, "bias"), _ptr(residual, "residual")
    d.ldx, d.ldo, d.ldr = x.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0
    d.B, d.H, d.W, d.Ho, d.Wo, d.Cout = B, H, W, H, W, Cout
    d.kh = d.kw = k
    d.stride, d.pad_t, d.pad_l = 1, pad, pad
    d.wbits, d.w_tiled, d.epilogue = (17 if x.dtype == torch.float16 else 16), 1, EPI_LINEAR
    d.out_dtype = BF16 if out.dtype == torch.bfloat16 else F16 if out.dtype == torch.float16 else F32
    d.gn_part = _ptr(gn_part, "gn_part")
    if gn_part is not None:
        d.gn_ld = part_ld(gn_part)
    d.upsample2x = 1 if upsample2x else 0
    d.nseg = 1
    d.seg[0].c0, d.seg[0].clen = 0, Cin_pad
    _check(load().qd_conv2d_bf16(ctypes.byref(d), _stream()), "qd_conv2d_bf16")


def groupnorm_silu_bf16(x, B, S, C, groups, eps, gamma, beta, silu, out, ws, part=None):
    """fp32 rows [B*S][ldx] -> GroupNorm (+ SiLU) -> bf16 / fp16 rows [B*S][ldo] (the type of `out`); part: first-level
    statistics of the producer."""
    if out.dtype not in (torch.bfloat16, torch.float16):
        raise HipEngineError("groupnorm_silu_bf16: out must be bf16 or fp16 rows")
    nchunk, pld = (part.shape[1], part_ld(part)) if part is not None else (0, 0)
    _check(load().qd_groupnorm_silu_h16(_ptr(x, "x"), B, S, C, x.stride(0), groups, float(eps), _ptr(gamma), _ptr(beta),
                                        1 if silu else 0, BF16 if out.dtype == torch.bfloat16 else F16, _ptr(out, "out"), out.stride(0),
                                        _ptr(ws, "ws"), _ptr(part), nchunk, pld, _stream()), "qd_groupnorm_silu_h16")


def groupnorm_ws_bytes(B, C, S):
    return int(load().qd_groupnorm_ws_bytes(B, C, S))


def part_ld(part):
    """Channels per chunk row of a first-level GroupNorm statistics buffer [B][nchunk][C][2] — C for a buffer of its own,
    more for a column range of a wider one (the two halves of a skip concatenation share one buffer: engine.CatSlot)."""
    B, n, C, two = part.shape
    ld2 = part.stride(1) if n > 1 else (part.stride(0) // max(n, 1) if B > 1 else 2 * C)
    if (two != 2 or part.stride(3) != 1 or part.stride(2) != 2 or ld2 % 2 or ld2 < 2 * C
            or (B > 1 and part.stride(0) != n * ld2)):
        raise HipEngineError(f"GroupNorm statistics buffer has an unsupported layout: shape {tuple(part.shape)} strides {part.stride()}")
    return ld2 // 2


def raw_quant_desc(raw):
    """raw: None or dict(out=int8 rows [M][ldo], segs=[dict(c0, clen, oc0, qparams, grid), ...]) -> (RawQuant | None, keepalive)."""
    if raw is None:
        return None, None
    r = RawQuant()
    r.out, r.ldo, r.nseg = _ptr(raw["out"], "raw out"), raw["out"].stride(0), len(raw["segs"])
    keep = []
    for i, sg in enumerate(raw["segs"]):
        qp = _qp(sg["qparams"])
        keep.append(qp)
        g = r.seg[i]
        g.c0, g.clen, g.oc0 = sg["c0"], sg["clen"], sg["oc0"]
        g.qmin, g.qmax, g.off = sg["grid"].qmin, sg["grid"].qmax, sg["grid"].off
        g.qparams = _ptr(qp, "raw qparams")
    return r, keep


def groupnorm_silu_quant(x, B, S, C, ldx, groups, eps, gamma, beta, silu, qparams, grid, out, ldo, ws, yout=None, ldy=0,
                         part=None, raw=None, mod=None):
    """part: optional [B][nchunk][C][2] fp32 first-level statistics written by the producer of x (ConvCall.gn_part);
    raw: optional second output, the un-normalised input quantised for the residual block's 1x1 skip connection
    (include/qdiff_hip.h qd_raw_quant);
    mod: optional [B][>= 2C] fp32 rows scale | shift of a use_scale_shift_norm block (qd_groupnorm_mod_silu_quant)."""
    g = grid or Grid(0, 0, 0)
    if mod is not None:
        if raw is not None:
            raise HipEngineError("groupnorm_silu_quant: a modulated norm has no raw second output")
        if mod.dtype != torch.float32 or mod.dim() != 2 or mod.shape[0] != B or mod.shape[1] < 2 * C or mod.stride(1) != 1:
            raise HipEngineError("groupnorm_silu_quant: mod must be fp32 rows [B][>= 2C]")
        _check(load().qd_groupnorm_mod_silu_quant(_ptr(x), _dtype(x), B, S, C, ldx, groups, float(eps), _ptr(gamma), _ptr(beta),
                                                  1 if silu else 0, _ptr(_qp(qparams)), g.qmin, g.qmax, g.off, _ptr(out), ldo,
                                                  _ptr(yout), ldy, _ptr(ws), _ptr(part), part.shape[1] if part is not None else 0,
                                                  part_ld(part) if part is not None else 0, _ptr(mod), mod.stride(0), _stream()),
               "qd_groupnorm_mod_silu_quant")
        return
    rq, keep = raw_quant_desc(raw)
    _check(load().qd_groupnorm_silu_quant(_ptr(x), _dtype(x), B, S, C, ldx, groups, float(eps), _ptr(gamma), _ptr(beta),
                                          1 if silu else 0, _ptr(_qp(qparams)), g.qmin, g.qmax, g.off, _ptr(out), ldo,
                                          _ptr(yout), ldy, _ptr(ws), _ptr(part), part.shape[1] if part is not None else 0,
                                          part_ld(part) if part is not None else 0,
                                          ctypes.byref(rq) if rq is not None else None, _stream()), "qd_groupnorm_silu_quant")


def layernorm_quant(x, M, C, ldx, eps, gamma, beta, qparams_list, grids, outs, ldo):
    n = len(outs)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    qparams_list = [_qp(q) for q in qparams_list]
    qp = (vp * n)(*[q.data_ptr() for q in qparams_list])
    for q in list(qparams_list) + list(outs):
        _ptr(q)
    mn = (i32 * n)(*[g.qmin for g in grids])
    mx = (i32 * n)(*[g.qmax for g in grids])
    of = (i32 * n)(*[g.off for g in grids])
    op = (vp * n)(*[o.data_ptr() for o in outs])
    _check(load().qd_layernorm_quant(_ptr(x), _dtype(x), M, C, ldx, float(eps), _ptr(gamma), _ptr(beta), n, qp, mn, mx,
                                     of, op, ldo, _stream()), "qd_layernorm_quant")


def geglu_quant(h, M, F, ldh, qparams, grid, out, ldo):
    _check(load().qd_geglu_quant(_ptr(h), _dtype(h), M, F, ldh, _ptr(_qp(qparams)), grid.qmin, grid.qmax, grid.off, _ptr(out),
                                 ldo, _stream()), "qd_geglu_quant")


def quantize_heads(x, B, T, H, d, strides, prescale, qparams, grid, transpose, out, rsum, Tpad, dpad):
    sb, st, sh, sd = strides
    _check(load().qd_quantize_heads(_ptr(x), _dtype(x), B, T, H, d, sb, st, sh, sd, float(prescale), _ptr(_qp(qparams)),
                                    grid.qmin, grid.qmax, grid.off, 1 if transpose else 0, _ptr(out), _ptr(rsum), Tpad,
                                    dpad, _stream()), "qd_quantize_heads")


def attn_uses_keyterm(d, S, q_asym):
    """The attention launcher takes a key-term table for this head dim and key count (qd_attn_keyterm)."""
    return bool(load().qd_attn_uses_keyterm(int(d), int(S), 1 if q_asym else 0))


def attn_keyterm(k, BH, Spad, dpad, prm, kterm=None):
    """kterm[bh][j] = 0x4B400000 - zq' * sum_c k[bh][j][c]: the score-accumulator seeds of qd_attn_i8 (int32 [BH][Spad])."""
    if kterm is None:
        kterm = torch.empty((BH, Spad), dtype=torch.int32, device=k.device)
    _check(load().qd_attn_keyterm(_ptr(k), BH, Spad, dpad, _ptr(prm), _ptr(kterm), _stream()), "qd_attn_keyterm")
    return kterm


def attn_config(pipe_mode=-1, xcd=-1, ktab=-1, lean=-1):
    load().qd_attn_config(int(pipe_mode), int(xcd), int(ktab), int(lean))


_ATTN_WS = {}            # (device, stream) -> scratch of the three-launch attention path (grows; old buffers stay alive for captured graphs)


def attn_workspace(device, BH, T, S, d):
    """Stream-ordered scratch for qd_attn_i8 (per-query statistics + per-block flags of the LDS-staged path), or None when the
    shape runs on a one-kernel path.  One buffer per (device, stream): calls on a stream are ordered, each consumes what it wrote."""
    need = int(load().qd_attn_ws_bytes(int(BH), int(T), int(S), int(d)))
    if need == 0:
        return None
    bufs = _ATTN_WS.setdefault((device, _stream()), [])
    if not bufs or bufs[-1].numel() < need:
        bufs.append(torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device))
    return bufs[-1]


def attn_i8(q, k, vt, vsum, BH, H, T, S, d, Tpad, Spad, dpad, prm, wbits, wmin, wmax, q_asym, out, ldo,
            out8=None, oq_params=None, oq_grid=None, kterm=None):
    """q_asym: the q quantiser has a non-zero stored zero point (the kernel restores -zq'*sum_d k' itself).
    out8 (+ oq_params, oq_grid): write the output as the int8 input rows of the consuming Linear instead of fp32.
    kterm: the table attn_keyterm built for this k operand (a caller with a static k — a prepared cross-attention context —
    passes its own); None: built here when the head dim takes one."""
    g = oq_grid
    if kterm is None and attn_uses_keyterm(d, S, q_asym):
        kterm = attn_keyterm(k, BH, Spad, dpad, prm)
    ws = attn_workspace(q.device, BH, T, S, d)
    _check(load().qd_attn_i8(_ptr(q), _ptr(k), _ptr(vt), None, _ptr(kterm), _ptr(vsum), BH, H, T, S, d, Tpad, Spad,
                             dpad, _ptr(prm), wbits, wmin, wmax, 1 if q_asym else 0, _ptr(out), ldo,
                             _ptr(out8), out8.stride(0) if out8 is not None else 0, _ptr(_qp(oq_params)),
                             g.qmin if g else 0, g.qmax if g else 0, g.off if g else 0,
                             _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "qd_attn_i8")


def bmm_qk_i8(q8, k8, BH, T, S, d, Tpad, Spad, dpad, prm, out):
    """out [BH][T][S] fp32 = cs * sum_d (q'-zq')(k'-zk'); prm = device floats {cs, zq', zk'}."""
    _check(load().qd_bmm_qk_i8(_ptr(q8), _ptr(k8), BH, T, S, d, Tpad, Spad, dpad, _ptr(prm), _ptr(out), out.stride(1),
                               out.stride(0), _stream()), "qd_bmm_qk_i8")


def bmm_pv_i8(w, v8t, vsum, BH, T, S, d, Spad, dpad, prm, wbits, wmin, wmax, out):
    """w [BH][T][S] fp32 probabilities -> quantised (prm[3..6]) -> out [BH][d][T] fp32."""
    _check(load().qd_bmm_pv_i8(_ptr(w), w.stride(1), w.stride(0), _ptr(v8t), _ptr(vsum), BH, T, S, d, Spad, dpad, _ptr(prm),
                               wbits, wmin, wmax, _ptr(out), out.stride(1), out.stride(0), _stream()), "qd_bmm_pv_i8")


def fakequant_fwd(x, delta, zero_point, qmin, qmax):
    """y = (clamp(rint(x / delta) + zp, qmin, qmax) - zp) * delta, one launch (calibration: qdiff/quant_layer.FusedFakeQuant)."""
    y = torch.empty_like(x)
    _check(load().qd_fakequant_fwd(_ptr(x, "x"), x.numel(), _ptr(delta, "delta"), _ptr(zero_point, "zero_point"), int(qmin), int(qmax),
                                   _ptr(y), _stream()), "qd_fakequant_fwd")
    return y


def fakequant_bwd(x, gy, delta, zero_point, qmin, qmax):
    """(d loss / dx, d loss / d delta) of fakequant_fwd given d loss / dy."""
    gx = torch.empty_like(x)
    part = torch.empty(int(load().qd_fakequant_blocks(x.numel())), dtype=torch.float32, device=x.device)
    _check(load().qd_fakequant_bwd(_ptr(x, "x"), _ptr(gy, "gy"), x.numel(), _ptr(delta, "delta"), _ptr(zero_point, "zero_point"),
                                   int(qmin), int(qmax), _ptr(gx), _ptr(part), _stream()), "qd_fakequant_bwd")
    return gx, part.sum()


def box_probe(device, kind, blocks, iters):
    """qd_box_probe timed with HIP events: (milliseconds, mean shader-clock ticks of the blocks' first waves)."""
    ticks = torch.zeros(blocks, dtype=torch.int64, device=device)
    sink = torch.zeros(1, dtype=torch.int32, device=device)
    _check(load().qd_box_probe(kind, blocks, 16, _ptr(ticks), _ptr(sink), _stream()), "qd_box_probe")     # warm-up (code load, clocks)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _check(load().qd_box_probe(kind, blocks, iters, _ptr(ticks), _ptr(sink), _stream()), "qd_box_probe")
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1), float(ticks.double().mean())


class _TembLayer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("zc", ctypes.c_void_p), ("zw", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("qp", ctypes.c_void_p), ("zfill", ctypes.c_void_p),
                ("Cout", ctypes.c_int32), ("out_off", ctypes.c_int32)]


_TEMB_DESC = {}          # tuple(id(plan)) -> (layer records on the device, block table, #blocks, the plans themselves)


def _temb_desc(plans, offsets, device):
    key = tuple(id(p) for p in plans) + tuple(offsets)
    hit = _TEMB_DESC.get(key)
    if hit is not None:
        return hit
    recs = (_TembLayer * len(plans))()
    blocks = []
    for i, (p, off) in enumerate(zip(plans, offsets)):
        sg = p.segs[0]
        r = recs[i]
        r.w = p.pack.wq.data_ptr() + sg["kstep0"] * ((p.Cout + 31) // 32) * (1024 if p.pack.wbits == 4 else 2048)
        r.scale, r.zc, r.zw = _ptr(sg["scale"]), _ptr(sg.get("zc")), _ptr(sg.get("zw"))
        r.bias, r.qp, r.zfill = _ptr(p.bias), _ptr(p.qparams[0]), _ptr(sg.get("zfill"))
        r.Cout, r.out_off = p.Cout, off
        blocks += [(i, n0) for n0 in range(0, p.Cout, 64)]
    raw = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(device)
    tab = torch.tensor(blocks, dtype=torch.int32).to(device).contiguous()
    if len(_TEMB_DESC) > 64:
        _TEMB_DESC.clear()
    hit = _TEMB_DESC[key] = (raw, tab, len(blocks), list(plans))     # the plans (and through them every pointer) stay alive
    return hit


def temb_mlp(x, silu, plans, offsets, out):
    """x [B][K] fp32 rows; plans: single-segment ConvPlans of Linears that share this input; out[:, off:off+Cout] per plan."""
    B, K = x.shape
    p0 = plans[0]
    if any(len(p.segs) != 1 or p.pack.wbits != p0.pack.wbits or p.pack.Cin != K or p.pack.taps != 1 or
           (p.grids[0].qmin, p.grids[0].qmax, p.grids[0].off) != (p0.grids[0].qmin, p0.grids[0].qmax, p0.grids[0].off) for p in plans):
        raise HipEngineError("temb_mlp: the Linears must share input width, weight bits and activation grid (one segment each)")
    if x.stride(1) != 1 or x.stride(0) % 4 != 0:
        x = x.contiguous()
    raw, tab, nblk, _ = _temb_desc(plans, offsets, x.device)
    g = p0.grids[0]
    rows = max(1, min(B, (44 * 1024) // K))
    for b0 in range(0, B, rows):
        xb, ob = x[b0:b0 + rows], out[b0:b0 + rows]
        _check(load().qd_temb_mlp(_ptr(xb, "x"), xb.stride(0), xb.shape[0], K, 1 if silu else 0, raw.data_ptr(), len(plans),
                                  tab.data_ptr(), nblk, p0.pack.wbits, g.qmin, g.qmax, g.off, _ptr(ob, "out"), ob.stride(0),
                                  _stream()), "qd_temb_mlp")
