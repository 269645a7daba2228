"""CPU emulation of the libqdiff_hip.so entry points, at the level of qdiff.hip's Python wrappers.

TEST INFRASTRUCTURE.  The product never uses this: on a machine without the HIP library / GPU the
`qdiff` integer path raises.  Tests that exercise *host logic* (plans, packing decisions, fused block
wiring, checkpoint resume, sharding) on CPU install these functions over `qdiff.hip.*` with
monkeypatch.  Each function follows the contract of include/qdiff_hip.h literally (packed weight
layouts, stored-byte conventions, the zero-point restoration formula), using exact integer
arithmetic (fp64 convolutions of integer-valued tensors) — so it doubles as an executable statement of
the ABI.
"""
import math

import torch
import torch.nn.functional as F


def _codes(x, qp, grid):
    d, z = qp[0], qp[1]
    return torch.clamp(torch.round(x / d) + z, grid.qmin, grid.qmax)


def make_qparams(delta, zero_point):
    d, z = delta.detach().reshape(()).float(), zero_point.detach().reshape(()).float()
    return torch.stack([d, z, 1.0 / d, torch.ones(())])


def quantize_act(x, B, C, S, strides, qparams, grid, out, ldo, c0=0, clen=None, oc0=0):
    clen = C - c0 if clen is None else clen
    sb, sc, ss = strides
    v = torch.as_strided(x, (B, C, S), (sb, sc, ss))[:, c0:c0 + clen].float()
    q = (_codes(v, qparams, grid) - grid.off).to(torch.int8)            # [B, clen, S]
    rows = out.view(-1, ldo)
    pad = (clen + 15) // 16 * 16
    rows[:, oc0:oc0 + clen] = q.permute(0, 2, 1).reshape(B * S, clen)
    rows[:, oc0 + clen:oc0 + pad] = (qparams[1] - grid.off).round().to(torch.int8)


def pack_weights(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, mode, wq, ldk, kofs, wsum, codes=None):
    wv = w.reshape(Cout, Cin_total, taps)[:, c0:c0 + clen].float()
    d, z = delta.view(-1, 1, 1), zp.view(-1, 1, 1)
    if alpha is not None:
        q = torch.floor(wv / d) + (alpha.reshape(Cout, clen, taps) >= 0).float()
    else:
        q = torch.round(wv / d)
    code = torch.clamp(q + z, 0, n_levels - 1).to(torch.int64)
    if codes is not None:
        codes.copy_(code.to(torch.int32))
    zi = z.to(torch.int64)
    pad = (clen + 15) // 16 * 16
    if mode == 4:
        nib = torch.zeros(Cout, taps, pad, dtype=torch.int64)
        nib[:, :, :clen] = code.permute(0, 2, 1)
        ch = nib.view(Cout, taps, pad // 16, 16)
        by = torch.empty(Cout, taps, pad // 16, 8, dtype=torch.int64)
        for b in range(4):
            by[..., b] = ch[..., b] | (ch[..., 4 + b] << 4)
            by[..., 4 + b] = ch[..., 8 + b] | (ch[..., 12 + b] << 4)
        rows = wq.view(Cout, taps, ldk // 2)
        rows[:, :, kofs // 2:(kofs + pad) // 2] = by.view(Cout, taps, pad // 2).to(torch.uint8)
        s = (code - zi).sum(dim=(1, 2)) - zi.view(-1) * (pad - clen) * taps
    else:
        stored = code - 128 if mode == 8 else code - zi
        rows = wq.view(torch.int8).view(Cout, taps, ldk)
        rows[:, :, kofs:kofs + pad] = 0
        rows[:, :, kofs:kofs + clen] = stored.permute(0, 2, 1).to(torch.int8)
        s = stored.sum(dim=(1, 2))
    wsum += s.to(torch.int32)


def _nibble_bytes(ch):
    """[..., 16] nibble values -> [..., 8] bytes in the ABI's nibble order."""
    by = torch.empty(ch.shape[:-1] + (8,), dtype=torch.int64)
    for b in range(4):
        by[..., b] = ch[..., b] | (ch[..., 4 + b] << 4)
        by[..., 4 + b] = ch[..., 8 + b] | (ch[..., 12 + b] << 4)
    return by


def _nibble_values(by):
    """inverse of _nibble_bytes: [..., 8] bytes -> [..., 16] nibble values."""
    lo, hi = by & 15, by >> 4
    ch = torch.empty(by.shape[:-1] + (16,), dtype=torch.int64)
    for b in range(4):
        ch[..., b], ch[..., 4 + b] = lo[..., b], hi[..., b]
        ch[..., 8 + b], ch[..., 12 + b] = lo[..., 4 + b], hi[..., 4 + b]
    return ch


def pack_weights_t4(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, wt, kstep0, ntiles, wsum):
    """qd_pack_weights_t4: wt[kstep][ntile][ksub*2+half][n%32][8 B]."""
    wv = w.reshape(Cout, Cin_total, taps)[:, c0:c0 + clen].float()
    d, z = delta.view(-1, 1, 1), zp.view(-1, 1, 1)
    q = (torch.floor(wv / d) + (alpha.reshape(Cout, clen, taps) >= 0).float()) if alpha is not None else torch.round(wv / d)
    code = torch.clamp(q + z, 0, n_levels - 1).to(torch.int64)                 # [Cout, clen, taps]
    pad = (clen + 15) // 16 * 16
    nst = (pad + 63) // 64
    full = torch.zeros(ntiles * 32, taps, nst * 64, dtype=torch.int64)
    full[:Cout, :, :clen] = code.permute(0, 2, 1)
    units = full.view(ntiles, 32, taps, nst, 4, 16)                             # [jt, nn, t, cs, kh4, 16]
    by = _nibble_bytes(units).permute(2, 3, 0, 4, 1, 5).contiguous()          # [t, cs, jt, kh4, nn, 8]
    view = wt.view(-1, ntiles, 4, 32, 8)
    view[kstep0:kstep0 + taps * nst] = by.view(taps * nst, ntiles, 4, 32, 8).to(torch.uint8)
    wsum += code.sum(dim=(1, 2)).to(torch.int32)          # raw nibble sums; zero point restored via zw in the epilogue


def pack_weights_t8(w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, wt, kstep0, ntiles, wsum):
    """qd_pack_weights_t8: wt[kstep][ntile][ksub*2+half][n%32][16 B], stored byte = W - 128."""
    wv = w.reshape(Cout, Cin_total, taps)[:, c0:c0 + clen].float()
    d, z = delta.view(-1, 1, 1), zp.view(-1, 1, 1)
    q = (torch.floor(wv / d) + (alpha.reshape(Cout, clen, taps) >= 0).float()) if alpha is not None else torch.round(wv / d)
    stored = torch.clamp(q + z, 0, n_levels - 1).to(torch.int64) - 128                  # [Cout, clen, taps]
    pad = (clen + 15) // 16 * 16
    nst = (pad + 63) // 64
    full = torch.zeros(ntiles * 32, taps, nst * 64, dtype=torch.int64)
    full[:Cout, :, :clen] = stored.permute(0, 2, 1)
    units = full.view(ntiles, 32, taps, nst, 4, 16).permute(2, 3, 0, 4, 1, 5).contiguous()   # [t, cs, jt, kh4, nn, 16]
    view = wt.view(torch.int8).view(-1, ntiles, 4, 32, 16)
    view[kstep0:kstep0 + taps * nst] = units.view(taps * nst, ntiles, 4, 32, 16).to(torch.int8)
    wsum += stored.sum(dim=(1, 2)).to(torch.int32)


def _unpack_rows(c, seg):
    """stored int64 weight bytes [Cout, taps, clen] of one segment (after the nibble unpack)."""
    taps = c.kh * c.kw
    if getattr(c, "w_tiled", False) and c.wbits == 8:
        ntiles, nst = (c.Cout + 31) // 32, (seg["clen"] + 63) // 64
        k0 = seg.get("kstep0", 0)
        blk = c.w.view(torch.int8).view(-1, ntiles, 4, 32, 16)[k0:k0 + taps * nst].to(torch.int64)   # [t*cs, jt, kh4, nn, 16]
        vals = blk.view(taps, nst, ntiles, 4, 32, 16)
        return vals.permute(2, 4, 0, 1, 3, 5).reshape(ntiles * 32, taps, nst * 64)[:c.Cout, :, :seg["clen"]]
    if getattr(c, "w_tiled", False):
        ntiles, nst = (c.Cout + 31) // 32, (seg["clen"] + 63) // 64
        k0 = seg.get("kstep0", 0)
        blk = c.w.view(-1, ntiles, 4, 32, 8)[k0:k0 + taps * nst].to(torch.int64)   # [t*cs, jt, kh4, nn, 8]
        vals = _nibble_values(blk).view(taps, nst, ntiles, 4, 32, 16)
        return vals.permute(2, 4, 0, 1, 3, 5).reshape(ntiles * 32, taps, nst * 64)[:c.Cout, :, :seg["clen"]]
    if c.wbits == 4:
        raw = c.w.view(c.Cout, taps, c.ldk // 2)[:, :, seg["kofs"] // 2:(seg["kofs"] + seg["clen"]) // 2].to(torch.int64)
        lo, hi = raw & 15, raw >> 4
        ch = torch.empty(c.Cout, taps, seg["clen"] // 16, 16, dtype=torch.int64)
        lo, hi = lo.view(c.Cout, taps, -1, 8), hi.view(c.Cout, taps, -1, 8)
        for b in range(4):
            ch[..., b], ch[..., 4 + b] = lo[..., b], hi[..., b]
            ch[..., 8 + b], ch[..., 12 + b] = lo[..., 4 + b], hi[..., 4 + b]
        wz = seg.get("wzp")
        z = wz.to(torch.int64).view(-1, 1, 1) if wz is not None else 0
        return ch.view(c.Cout, taps, seg["clen"]) - z
    return c.w.view(torch.int8).view(c.Cout, taps, c.ldk)[:, :, seg["kofs"]:seg["kofs"] + seg["clen"]].to(torch.int64)


def conv2d_i8(c, acc_out=None):
    B, H, W, Ho, Wo = c.B, c.H, c.W, c.Ho, c.Wo
    if getattr(c, "upsample2x", False):                     # x is the half-resolution map: nearest-2x replication of its rows
        x = c.x.view(B, H // 2, 1, W // 2, 1, c.ldx).expand(B, H // 2, 2, W // 2, 2, c.ldx).reshape(B, H, W, c.ldx).to(torch.int64)
    else:
        x = c.x.view(B, H, W, c.ldx).to(torch.int64)
    total = None
    for seg in c.segs:
        zf = seg.get("zfill")
        zprime, kz = (int(zf[0]), int(zf[1])) if zf is not None else (0, 0)
        xs = x[..., seg["c0"]:seg["c0"] + seg["clen"]].permute(0, 3, 1, 2).double()       # stored bytes, NCHW
        pb = max((Ho - 1) * c.stride + c.kh - c.pad_t - H, 0)
        pr = max((Wo - 1) * c.stride + c.kw - c.pad_l - W, 0)
        xs = F.pad(xs, (c.pad_l, pr, c.pad_t, pb), value=float(zprime))                    # padded taps hold z'
        ws = _unpack_rows(c, seg).view(c.Cout, c.kh, c.kw, seg["clen"]).permute(0, 3, 1, 2).double()
        acc = F.conv2d(xs, ws, stride=c.stride)[:, :, :Ho, :Wo]
        asum = F.conv2d(xs, torch.ones(1, seg["clen"], c.kh, c.kw, dtype=torch.float64), stride=c.stride)[:, :, :Ho, :Wo]
        I = acc
        if seg.get("zc") is not None:
            I = I - seg["zc"].double().view(1, -1, 1, 1)
        if seg.get("zw") is not None:
            I = I - seg["zw"].double().view(1, -1, 1, 1) * (asum - kz)
        if acc_out is not None:
            acc_out.copy_(I.permute(0, 2, 3, 1).reshape(-1, c.Cout).round().to(torch.int32))
            return
        part = I.float() * seg["scale"].view(1, -1, 1, 1)
        total = part if total is None else total + part
    if c.bias is not None:
        total = total + c.bias.view(1, -1, 1, 1)
    if c.rowbias is not None:
        total = total + c.rowbias[:, :c.Cout].view(B, c.Cout, 1, 1)
    rows = total.permute(0, 2, 3, 1).reshape(-1, c.Cout)
    if c.residual is not None:
        rows = rows + c.residual[:, :c.Cout].float()
    if getattr(c, "epilogue", 0) == 1:
        # QD_EPI_GEGLU_I8: packed rows are (value tile, gate tile) interleaved per 32
        t = rows.view(rows.shape[0], c.Cout // 64, 2, 32)
        y = (t[:, :, 0] * F.gelu(t[:, :, 1])).reshape(rows.shape[0], c.Cout // 2)
        c.out[:, :c.Cout // 2] = (_codes(y, c.oq_params, c.oq_grid) - c.oq_grid.off).to(torch.int8)
        return
    if getattr(c, "epilogue", 0) in (2, 3):
        # QD_EPI_HEADS_I8 / QD_EPI_HEADS_T_I8: the projection output as attention operand bytes
        hd = c.heads
        Bn, inner = rows.shape[0] // hd["T"], hd["H"] * hd["d"]
        rsum = torch.zeros_like(hd["sum"]) if (c.epilogue == 3 and hd.get("sum") is not None) else None
        quantize_heads(rows.contiguous(), Bn, hd["T"], hd["H"], hd["d"], (hd["T"] * inner, inner, hd["d"], 1), hd["prescale"],
                       c.oq_params, c.oq_grid, c.epilogue == 3, c.out, rsum, hd["Tpad"], hd["dpad"])
        if rsum is not None:
            hd["sum"] += rsum                                    # the kernel accumulates atomically into a zeroed buffer
        return
    c.out[:, :c.Cout] = rows.to(c.out.dtype)
    if getattr(c, "gn_part", None) is not None:                  # first level of GroupNorm statistics, 128-row chunks
        ch = c.out[:, :c.Cout].float().reshape(-1, 128, c.Cout)
        c.gn_part.copy_(torch.stack([ch.sum(1), (ch * ch).sum(1)], dim=-1).view(c.gn_part.shape))   # may be a column range (gn_ld)


def conv2d_i8_group(calls):
    """qd_conv2d_i8_group: the members one after the other (what the grouped launch is defined to equal) — through
    hip.conv2d_i8, so that a test's launch counter on that entry sees every member."""
    from qdiff import hip
    for c in calls:
        hip.conv2d_i8(c)


def groupnorm_ws_bytes(B, C, S):
    return 64


def groupnorm_silu_quant(x, B, S, C, ldx, groups, eps, gamma, beta, silu, qparams, grid, out, ldo, ws, yout=None, ldy=0,
                         part=None, raw=None, mod=None):
    v = x[:, :C].float().reshape(B, S, C).permute(0, 2, 1)
    if mod is not None:
        # qd_groupnorm_mod_silu_quant: scale | shift rows of a use_scale_shift_norm block
        assert raw is None and mod.dtype == torch.float32 and mod.shape[0] == B and mod.shape[1] >= 2 * C
        sc, sf = 1.0 + mod[:, :C].reshape(B, C, 1), mod[:, C:2 * C].reshape(B, C, 1)
    if raw is not None:
        # qd_raw_quant: the un-normalised input quantised per channel segment for the 1x1 skip connection
        for sg in raw["segs"]:
            xs = x[:, sg["c0"]:sg["c0"] + sg["clen"]].float()
            raw["out"][:, sg["oc0"]:sg["oc0"] + sg["clen"]] = (_codes(xs, sg["qparams"], sg["grid"]) - sg["grid"].off).to(torch.int8)
    if part is not None:
        # statistics from the producer's partial sums (what the kernel does with part_in), fp64 second level
        st = part.double().reshape(B, -1, groups, C // groups, 2).sum(dim=(1, 3))
        n = S * (C // groups)
        mean = st[..., 0] / n
        var = (st[..., 1] / n - mean * mean).clamp_min(0)
        rstd = (1.0 / torch.sqrt(var + eps)).float().repeat_interleave(C // groups, dim=1).view(B, C, 1)
        fmean = mean.float().repeat_interleave(C // groups, dim=1).view(B, C, 1)
        a = rstd * gamma.view(1, C, 1)
        sh = beta.view(1, C, 1) - fmean * a
        if mod is not None:                                   # the kernel folds the modulation into the affine
            a, sh = a * sc, sh * sc + sf
        y = v * a + sh
    else:
        y = F.group_norm(v, groups, gamma, beta, eps)
        if mod is not None:
            y = y * sc + sf
    if silu:
        y = y * torch.sigmoid(y)
    rows = y.permute(0, 2, 1).reshape(B * S, C)
    if yout is not None:
        yout[:, :C] = rows
    if out is not None:
        out[:, :C] = (_codes(rows, qparams, grid) - grid.off).to(torch.int8)


def layernorm_quant(x, M, C, ldx, eps, gamma, beta, qparams_list, grids, outs, ldo):
    y = F.layer_norm(x[:, :C].float(), (C,), gamma, beta, eps)
    for qp, g, o in zip(qparams_list, grids, outs):
        o[:, :C] = (_codes(y, qp, g) - g.off).to(torch.int8)


def geglu_quant(h, M, F_, ldh, qparams, grid, out, ldo):
    y = h[:, :F_].float() * F.gelu(h[:, F_:2 * F_].float())
    out[:, :F_] = (_codes(y, qparams, grid) - grid.off).to(torch.int8)


def _perm_index(Tpad):
    idx = torch.empty(Tpad, dtype=torch.long)
    for p in range(Tpad):
        tile, pp = divmod(p, 32)
        half, r = divmod(pp, 16)
        idx[p] = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half
    return idx


def quantize_heads(x, B, T, H, d, strides, prescale, qparams, grid, transpose, out, rsum, Tpad, dpad):
    v = torch.as_strided(x, (B, T, H, d), strides).float() * prescale
    q = (_codes(v, qparams, grid) - grid.off).to(torch.int64).permute(0, 2, 1, 3).reshape(B * H, T, d)
    buf = torch.zeros(B * H, Tpad, dpad, dtype=torch.int64)
    buf[:, :T, :d] = q
    if not transpose:
        out.view(buf.shape).copy_(buf.to(torch.int8))        # out may be viewed as plain rows [B*T][dpad] (H = 1)
        if rsum is not None:
            rsum.copy_(buf.sum(-1).to(torch.int32))
    else:
        out.copy_(buf.permute(0, 2, 1)[:, :, _perm_index(Tpad)].to(torch.int8))
        if rsum is not None:
            rsum.copy_(buf.sum(1).to(torch.int32))


def attn_uses_keyterm(d, S, q_asym):
    return bool(q_asym) and d < 64 and d % 32 != 0 and S >= 512


def attn_keyterm(k, BH, Spad, dpad, prm, kterm=None):
    """qd_attn_keyterm: accumulator seeds 0x4B400000 - zq' * sum_c k[bh][j][c] (int32 [BH][Spad])."""
    t = (0x4B400000 - int(float(prm[1])) * k.to(torch.int64).sum(-1)).to(torch.int32)
    if kterm is None:
        return t
    kterm.copy_(t)
    return kterm


def attn_i8(q, k, vt, vsum, BH, H, T, S, d, Tpad, Spad, dpad, prm, wbits, wmin, wmax, q_asym, out, ldo,
            out8=None, oq_params=None, oq_grid=None, kterm=None):
    if kterm is not None:        # a caller-supplied table must be the one of THIS k operand and zero point
        assert torch.equal(kterm, attn_keyterm(k, BH, Spad, dpad, prm)), "stale key-term table"
    cs, zq, zk, dw, zpw, osc, zv = (float(prm[i]) for i in range(7))
    inv = torch.empty(Spad, dtype=torch.long)
    inv[_perm_index(Spad)] = torch.arange(Spad)
    qi = q.to(torch.int64)[:, :T, :d] - int(zq)
    ki = k.to(torch.int64)[:, :S, :d] - int(zk)
    vi = vt.to(torch.int64)[:, :, inv].permute(0, 2, 1)[:, :S, :d] - int(zv)
    s = torch.einsum("bid,bjd->bij", qi.double(), ki.double()).float() * cs
    p = torch.softmax(s, dim=-1)
    u = torch.clamp(torch.round(p / dw) + zpw, wmin, wmax) - zpw
    o = torch.einsum("bij,bjd->bid", u.double(), vi.double()).float() * osc
    B = BH // H
    rows = o.view(B, H, T, d).permute(0, 2, 1, 3).reshape(B * T, H * d)
    if out8 is not None:
        out8[:, :H * d] = (_codes(rows, oq_params, oq_grid) - oq_grid.off).to(torch.int8)
    else:
        out[:, :H * d] = rows


def bmm_qk_i8(q8, k8, BH, T, S, d, Tpad, Spad, dpad, prm, out):
    cs, zq, zk = float(prm[0]), int(prm[1]), int(prm[2])
    qi = q8.to(torch.int64)[:, :T, :d] - zq
    ki = k8.to(torch.int64)[:, :S, :d] - zk
    out.copy_(torch.einsum("bid,bjd->bij", qi.double(), ki.double()).float() * cs)


def bmm_pv_i8(w, v8t, vsum, BH, T, S, d, Spad, dpad, prm, wbits, wmin, wmax, out):
    dw, zpw, osc, zv = (float(prm[i]) for i in range(3, 7))
    inv = torch.empty(Spad, dtype=torch.long)
    inv[_perm_index(Spad)] = torch.arange(Spad)
    vi = v8t.to(torch.int64)[:, :, inv][:, :d, :S] - int(zv)                    # [BH, d, S]
    u = torch.clamp(torch.round(w.float() / dw) + zpw, wmin, wmax) - zpw
    out.copy_(torch.einsum("bts,bcs->bct", u.double(), vi.double()).float() * osc)


def temb_mlp(x, silu, plans, offsets, out):
    """qd_temb_mlp: SiLU -> each Linear's own activation quantiser -> exact integer contraction -> generic epilogue."""
    from qdiff import engine
    y = F.silu(x.float()) if silu else x.float()
    B, K = y.shape
    for plan, off in zip(plans, offsets):
        xq = engine.quantize_rows(y, plan, 1, K, B, (0, 1, y.stride(0)))
        out[:, off:off + plan.Cout] = engine.conv_forward(plan, xq, 1, 1, B, 1, B, splitk=False)


def splitk_ws_bytes(c):
    """The emulation never splits K (the schedule does not change results)."""
    return 0


# ---- first-stage decoder entry points (include/qdiff_hip.h "First-stage decoder"; qdiff.hip wrappers of the same names) ----

def pack_weights_bf16(w, dtype=torch.bfloat16):
    """fp32 OIHW -> bf16 (or IEEE halves: qd_pack_weights_h16 with wbits = 17) in the tile order of qd_pack_weights_bf16: per
    (tap, 32-channel K-step, 32-output-channel tile) 2 KB as [k-half (16 ch)][lane-half (8 ch)][n % 32][8 elements]; returned as
    the uint8 buffer the kernel would read."""
    w = w.detach().float()
    if w.dim() == 2:
        w = w[:, :, None, None]
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3]
    cpad = (Cin + 7) // 8 * 8
    nst, ntl = (cpad + 31) // 32, (Cout + 31) // 32
    buf = torch.zeros(taps * nst * ntl * 1024, dtype=torch.int16)
    bits = w.reshape(Cout, Cin, taps).to(dtype).view(torch.int16)
    n, c, t = torch.meshgrid(torch.arange(Cout), torch.arange(Cin), torch.arange(taps), indexing="ij")
    off = ((t * nst + c // 32) * ntl + n // 32) * 1024 + (((c % 32) // 8) * 32 + n % 32) * 8 + c % 8
    buf[off.reshape(-1)] = bits.reshape(-1)
    return buf.view(torch.uint8)


def _unpack_weights_bf16(wt, Cout, cpad, taps, dtype=torch.bfloat16):
    nst, ntl = (cpad + 31) // 32, (Cout + 31) // 32
    buf = wt.view(torch.int16)
    n, c, t = torch.meshgrid(torch.arange(Cout), torch.arange(cpad), torch.arange(taps), indexing="ij")
    off = ((t * nst + c // 32) * ntl + n // 32) * 1024 + (((c % 32) // 8) * 32 + n % 32) * 8 + c % 8
    return buf[off.reshape(-1)].view(dtype).float().reshape(Cout, cpad, taps)


def conv2d_bf16(x, wt, bias, out, B, H, W, Cin_pad, Cout, k=3, pad=1, residual=None, gn_part=None, upsample2x=False):
    """out[m][n] = sum_k x w + bias[n] (+ residual[m][n]); exact bf16 products, fp32 accumulation (here: fp64, rounded once);
    x rows are at half resolution when upsample2x (nearest-2x folded into the gather)."""
    assert x.dtype in (torch.bfloat16, torch.float16) and out.dtype in (torch.float32, x.dtype)
    hin, win = (H // 2, W // 2) if upsample2x else (H, W)
    w = _unpack_weights_bf16(wt, Cout, Cin_pad, k * k, x.dtype).reshape(Cout, Cin_pad, k, k)
    xi = x[:, :Cin_pad].double().reshape(B, hin, win, Cin_pad).permute(0, 3, 1, 2)
    if upsample2x:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xi, w.double(), None if bias is None else bias.double(), padding=pad).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    if residual is not None:
        assert residual.dtype == out.dtype
        y = y + residual.double()
    y = y.float()
    if gn_part is not None:
        v = y.double().reshape(B, H * W // 128, 128, Cout)
        gn_part.copy_(torch.stack([v.sum(2), (v * v).sum(2)], dim=-1).float())
    out.copy_(y.to(out.dtype))


def groupnorm_silu_bf16(x, B, S, C, groups, eps, gamma, beta, silu, out, ws, part=None):
    """GroupNorm (+ swish) of fp32 rows -> bf16 rows; statistics from `part` ([B][chunks][C][2] sums / sums of squares) when
    the producer supplied them.  y = x * a + sh with a = rstd * gamma, sh = beta - mean * a (the kernel's float order)."""
    v = x[:, :C].reshape(B, S, C)
    if part is not None:
        s, q = part[..., 0].double().sum(1), part[..., 1].double().sum(1)            # [B, C]
    else:
        s, q = v.double().sum(1), (v.double() ** 2).sum(1)
    cpg = C // groups
    n = float(S * cpg)
    mean = s.reshape(B, groups, cpg).sum(-1) / n
    var = (q.reshape(B, groups, cpg).sum(-1) / n - mean * mean).clamp_min(0.0)
    rstd = (1.0 / torch.sqrt(var + eps)).float().repeat_interleave(cpg, dim=1)       # [B, C]
    fmean = mean.float().repeat_interleave(cpg, dim=1)
    a = rstd * (gamma.float() if gamma is not None else 1.0)
    sh = (beta.float() if beta is not None else 0.0) - fmean * a
    y = v.float() * a[:, None, :] + sh[:, None, :]
    if silu:
        y = y * (1.0 / (1.0 + torch.exp(-y)))
    assert out.dtype in (torch.bfloat16, torch.float16)
    out.copy_(y.reshape(B * S, C).to(out.dtype))


def install(monkeypatch):
    """Replace qdiff.hip's device entry points by the emulation (CPU tensors only)."""
    from qdiff import hip
    for name in ("make_qparams", "quantize_act", "pack_weights", "pack_weights_t4", "pack_weights_t8", "conv2d_i8", "conv2d_i8_group", "groupnorm_ws_bytes", "groupnorm_silu_quant",
                 "layernorm_quant", "geglu_quant", "quantize_heads", "attn_i8", "attn_keyterm", "attn_uses_keyterm", "splitk_ws_bytes", "bmm_qk_i8", "bmm_pv_i8", "temb_mlp",
                 "pack_weights_bf16", "conv2d_bf16", "groupnorm_silu_bf16"):
        monkeypatch.setattr(hip, name, globals()[name])
