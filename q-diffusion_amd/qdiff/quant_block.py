"""Quantised UNet blocks — counterpart of the reference's qdiff/quant_block.py (same class names,
constructor signatures, quantiser attribute names and therefore state-dict keys).

Each block has two forwards:
  * `_forward_sim`: the reference composition (norm -> act -> QuantModule ...), used whenever the
    block is not fully in (weight_quant, act_quant) = (True, True) state, under autograd
    (calibration), or while a quantiser still needs its data-dependent initialisation;
  * `_forward_int`: the MI355X path.  Activations stay channels-last; GroupNorm+SiLU (K5),
    LayerNorm (K9a) and GEGLU (K9b) are *producers* that emit the next QuantModule's int8 rows
    directly, the timestep-embedding add and the residual add ride in the conv epilogue (K3), and
    q/k/softmax/v quantisation + both attention contractions are one fused kernel (K7/K8).
"""
import logging
import os
from types import MethodType

import torch
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from . import engine
from .arch import ddim_unet, ldm_unet
from .quant_layer import QuantModule, StraightThrough, UniformAffineQuantizer

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _nhwc_rows(x):
    """[B,C,H,W] channels-last (possibly a column range of wider rows: row stride ld >= C) -> rows view [B*H*W, C] with
    stride (ld, 1); any other layout is copied to channels-last first."""
    B, C, H, W = x.shape
    ld = x.stride(3) if W > 1 else (x.stride(2) if H > 1 else (x.stride(0) if B > 1 else C))
    ok = (ld >= C and (C == 1 or x.stride(1) == 1) and (W == 1 or x.stride(3) == ld) and (H == 1 or x.stride(2) == W * ld)
          and (B == 1 or x.stride(0) == H * W * ld))
    if not ok:
        x = x.contiguous(memory_format=torch.channels_last)
        if x.stride(1) != 1 and C > 1:  # degenerate shapes where channels_last == contiguous
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = C
    rows = x.as_strided((B * H * W, C), (ld, 1), x.storage_offset())
    _carry_gn_part(x, rows)
    return rows


def _carry_gn_part(src, dst):
    """GroupNorm first-level statistics written by the kernel that produced `src` (engine.conv_forward(gn_stats=True))
    travel with the tensor as a plain attribute; views do not inherit attributes, so re-attach them by hand."""
    part = getattr(src, "qd_gn_part", None)
    if part is not None:
        dst.qd_gn_part = part
    return dst


def _rows_to_nchw(rows, B, H, W):
    return _carry_gn_part(rows, rows.view(B, H, W, rows.shape[1]).permute(0, 3, 1, 2))


def _adjacent(a, b, dim, unit):
    """`a` and `b` are views of ONE buffer that differ only in a column range along `dim` (element stride `unit`), `b`
    starting where `a` ends: their concatenation along `dim` is the view returned here (None otherwise)."""
    if (a.dtype != b.dtype or a.device != b.device or a.dim() != b.dim() or a.stride() != b.stride() or a.stride(dim) != unit
            or any(a.shape[i] != b.shape[i] for i in range(a.dim()) if i != dim)
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + a.shape[dim] * unit):
        return None
    shape = list(a.shape)
    shape[dim] += b.shape[dim]
    # the rows must be wide enough to hold both halves (true by construction for engine.CatSlot buffers)
    row = min((a.stride(i) for i in range(a.dim()) if i != dim and a.shape[i] > 1 and a.stride(i) > unit), default=None)
    if row is not None and row < shape[dim] * unit:
        return None
    return a.as_strided(shape, a.stride(), a.storage_offset())


def cat_channels(a, b):
    """torch.cat([a, b], dim=1) of two NCHW activations that keeps the producers' GroupNorm statistics.  When both
    producers wrote into the two halves of one planned buffer (engine.CatSlot) the result is a view: no copy."""
    out = _adjacent(a, b, 1, 1)
    if out is None:
        out = torch.cat([a, b], dim=1)
    pa, pb = getattr(a, "qd_gn_part", None), getattr(b, "qd_gn_part", None)
    if pa is not None and pb is not None and pa.shape[0] * pa.shape[1] == pb.shape[0] * pb.shape[1]:
        n = pa.shape[0] * pa.shape[1]
        if pa.shape[:2] != pb.shape[:2]:
            try:
                pb = pb.view(pa.shape[0], pa.shape[1], pb.shape[2], 2)
            except RuntimeError:
                pb = pb.reshape(pa.shape[0], pa.shape[1], pb.shape[2], 2)
        part = _adjacent(pa, pb, 2, 2)
        if part is None:
            part = torch.cat([pa.reshape(1, n, -1, 2), pb.reshape(1, n, -1, 2)], dim=2)
        out.qd_gn_part = part
    return out


def _tracking(m):
    """A QuantModule whose activation quantisers are in EMA range-tracking mode (QuantModel.set_running_stat(True),
    the reference's calibration loop: txt2img.py:457-468, quant_layer.py:77-78,91-110)."""
    return any(getattr(q, "running_stat", False) for q in m._act_quantizers()) if m.act_quant_mode == 'qdiff' else False


def _int_mode(*modules):
    """All given QuantModules take the integer path, autograd is off and no activation quantiser is tracking its range:
    range tracking updates delta / zero_point from every float input the quantiser sees, which only the simulation
    composition feeds it — a fused block in that mode runs `_forward_sim` exactly as the reference does."""
    return ((not torch.is_grad_enabled()) and all(isinstance(m, QuantModule) and m.int_ready() for m in modules)
            and not any(_tracking(m) for m in modules))


def _dropout_live(block):
    """The block is in train mode and holds a Dropout with p > 0: the reference then drops activations (its calibration
    helpers leave the model in train mode, utils.py:249, so the scripts initialise activation ranges under dropout), which
    only the simulation composition reproduces — the integer path is the eval-mode path."""
    if not block.training:
        return False
    live = block.__dict__.get("_qd_has_dropout")
    if live is None:
        live = block.__dict__["_qd_has_dropout"] = any(isinstance(m, nn.Dropout) and m.p > 0 for m in block.modules())
    return live


def _aq_ready(*quantizers):
    """Initialised and not in EMA range-tracking mode (a tracking quantiser must see its float input: simulation path)."""
    return all(q.inited and not q.running_stat for q in quantizers)


def _gn_silu_to(conv, rows, B, S, C, gn, silu=True, raw_plan=None, mod=None):
    """GroupNorm(+SiLU) -> int8 rows for `conv`; initialises conv's act quantiser on first use.
    raw_plan: also return the int8 rows of a 1x1 consumer of the un-normalised `rows` (the skip connection), quantised
    in the same pass.  mod: [B][2C] scale | shift rows of a use_scale_shift_norm block (reference :99-103)."""
    if not conv.act_quantizer.inited:
        y = F.group_norm(rows.view(B, S, C).permute(0, 2, 1).float(), gn.num_groups, gn.weight, gn.bias, gn.eps)
        if mod is not None:
            y = y * (1 + mod[:, :C, None]) + mod[:, C:2 * C, None]
        conv._init_act_quantizers(F.silu(y) if silu else y)
    res = engine.groupnorm_silu_quant(rows, B, S, C, gn, silu, plan=conv.conv_plan(), part=getattr(rows, "qd_gn_part", None),
                                      raw_plan=raw_plan, mod=mod)
    return (res[0], res[2]) if raw_plan is not None else res[0]


def _ln_to(consumers, rows, M, C, ln):
    """LayerNorm -> one int8 copy per consumer QuantModule."""
    if not all(m.act_quantizer.inited for m in consumers):
        y = F.layer_norm(rows.float(), (C,), ln.weight, ln.bias, ln.eps)
        for m in consumers:
            m._init_act_quantizers(y)
    return engine.layernorm_quant(rows, M, C, ln, [m.conv_plan() for m in consumers])


def _linear_rows(lin, rows, residual=None, gn_stats=False, slot=None):
    """QuantModule linear on float rows [M,K] -> [M,N] (quantise + integer GEMM)."""
    lin._init_act_quantizers(rows)
    plan = lin.conv_plan()
    M, K = rows.shape
    xq = engine.quantize_rows(rows, plan, 1, K, M, (0, rows.stride(1), rows.stride(0)))
    return engine.conv_forward(plan, xq, 1, 1, M, 1, M, residual=residual, gn_stats=gn_stats, slot=slot)




class EmbGroup:
    """The `SiLU -> Linear` timestep-embedding projections of all residual blocks of one UNet (reference
    quant_block.py:98-107 `emb_layers`, ddim diffusion.py:127 `temb_proj`): every block receives the SAME embedding
    tensor, so the first block that asks evaluates all of them with ONE launch (K6, csrc/temb_mlp.hip) and the others
    pick their slice.  Falls back to the per-block composition (returns None) whenever a projection is not on the
    integer path, is tracking its range, or still needs its data-dependent initialisation."""

    def __init__(self):
        self.members = []            # (block, QuantModule)
        self._emb, self._out, self._offs = None, None, {}

    def register(self, blk, linear):
        self.members.append((blk, linear))
        blk.__dict__["_emb_group"] = self

    def reset(self):
        """Start of a UNet evaluation (forward pre-hook of the wrapped model): nothing carries over between evaluations —
        the same tensor object may hold new values (static inputs of a captured graph, in-place updates)."""
        self._emb, self._out = None, None

    def get(self, blk, emb):
        if self._emb is not emb:
            self._emb, self._out = emb, None
            self._compute(emb)
        if self._out is None:
            return None
        off, n = self._offs[id(blk)]
        return self._out[:, off:off + n]

    def _compute(self, emb):
        lins = [l for _, l in self.members]
        if not torch.is_tensor(emb) or emb.dim() != 2 or not emb.is_floating_point() or len(lins) < 2 or emb.shape[1] % 16:
            return                                       # (qd_temb_mlp streams 16-byte chunks of K)
        if not _int_mode(*lins) or any(l.split or l.kind != 'linear' or not l.act_quantizer.inited for l in lins):
            return
        plans = [l.conv_plan() for l in lins]
        p0 = plans[0]
        g0 = (p0.grids[0].qmin, p0.grids[0].qmax, p0.grids[0].off)
        if any(len(p.segs) != 1 or p.pack.Cin != emb.shape[1] or p.pack.wbits != p0.pack.wbits
               or (p.grids[0].qmin, p.grids[0].qmax, p.grids[0].off) != g0 for p in plans):
            return
        offs, tot = [], 0
        for (blk, _), p in zip(self.members, plans):
            self._offs[id(blk)] = (tot, p.Cout)
            offs.append(tot)
            tot += (p.Cout + 63) // 64 * 64
        out = torch.empty((emb.shape[0], tot), dtype=torch.float32, device=emb.device)
        engine.hip.temb_mlp(emb.float(), True, plans, offs, out)
        self._out = out


# where the context branch forks off the main stream: "start" = the model's forward pre-hook (before the stem), "attn" = right
# before the first self-attention kernel of the first transformer block, "late" = at the first cross-attention (= its join)
_CTX_FORK = os.environ.get("QDIFF_CTX_FORK", "start")
# QDIFF_CTX_PIN=0: QuantModel.prepare_context becomes a no-op, i.e. the cross-attention K / V^T operands are recomputed by
# every evaluation as the reference does (quant_block.py:193-195) — the A/B knob of the once-per-sampling-run computation
_CTX_PIN = os.environ.get("QDIFF_CTX_PIN", "1") != "0"
# prepared contexts kept at a time (ContextKV): >= 1.  QDIFF_CTX_AUTO=0: only explicit QuantModel.prepare_context() prepares
# (default: QuantModel.forward prepares a context it has not seen — the unmodified reference samplers never announce theirs)
_CTX_PINS = max(1, int(os.environ.get("QDIFF_CTX_PINS", "2")))
_CTX_AUTO = os.environ.get("QDIFF_CTX_AUTO", "1") != "0"
# Switches of the equality tests (tests/test_host_logic.py, tests/test_engine_models.py compare both sides in one process); no
# environment variable selects them.
_FUSE_SKIP_QUANT = True     # skip-connection rows from the GroupNorm pass
QKV_HEADS = True            # the LDM AttentionBlock's qkv as three GEMMs with operand epilogues; q / k / v of the DDIM AttnBlock likewise
CAT_SLOTS = True            # planned skip-concatenation buffers (engine.CatSlot)


class ContextKV:
    """Cross-attention keys / values of ALL transformer blocks of one UNet, evaluated on a SIDE STREAM.
    to_k / to_v of attn2 (reference quant_block.py:190-221, ldm attention.py:152-198) see only the conditioning `context`
    — [B, 77, 768] for SD — never the latent: 16 blocks x (row quantiser + two skinny GEMMs on B*77 rows + head-layout
    quantisers) = ~130 tiny dependent launches per evaluation that no single one of can fill the chip (~1.3 ms of an SD
    evaluation when serialised into the main stream).  The branch (`start`) prepares the int8 attention operands of every
    block while the main stream does other work; the first cross-attention joins it.  Under HIP-graph capture the fork /
    join become parallel branches of the graph.  Results are those of the in-line path bit for bit (same kernels, same
    inputs).  Where it forks (`_CTX_FORK`, env QDIFF_CTX_FORK):
      late   in `get`, i.e. AFTER the first self-attention and immediately before the join — what round 2 shipped: the
             rocprofv3 timeline profiles/r03_sd_eval_timeline.tsv shows the 148 launches of the branch running alone for
             1.07 ms (traced) with the main stream waiting;
      start  (default) the model's forward pre-hook, before the stem: the branch runs under the stem / first residual block
             / first self-attention.  It does overlap (profiles/r03_ctx_fork_ab.md), but those are launches of exactly one
             wave of blocks (512 blocks on 256 CUs x 2): a branch kernel that holds a few slots at the wrong moment pushes
             the last blocks of the big kernel into a second wave (82 -> 141 us on one of them) — net -0.12 ms per SD
             evaluation (21.37 -> 21.25, A/B/A/B on one box), not the 0.5 ms the serial chain costs;
      attn   right before the first self-attention KERNEL of the first transformer block (4096 blocks, 1.0 ms): +0.2 ms —
             the attention kernel loses more to the stolen slots than the branch saves."""

    def __init__(self):
        self.members = []
        self._ctx, self._out, self._side, self._joined = None, None, None, True
        self._pins = []               # prepared contexts (dicts, see `pin`), least recently used first; each owns the buffers of its slot
        self._sel = None              # (entry, context object, owner) in force for the evaluation being issued
        self._slot_gen = {}           # slot -> how often its buffers were (re)written
        self.token = lambda: 0        # QuantModel installs its state token: a pin made under another token is stale
        self.chain_runs = 0           # how often the to_k / to_v chain was issued (tests: a prepared context must not add to it)
        self.value_matches = 0        # prepared contexts recognised by VALUE (a fresh tensor with the pinned bytes)

    # ---- operands prepared ONCE per sampling run (QuantModel.prepare_context, or on first sight: QuantModel.forward) --------
    # The reference recomputes k = to_k(context), v = to_v(context) in every evaluation (quant_block.py:193-195) although the
    # samplers hand it the same conditioning at every step (plms.py:184-187 concatenates the same `uncond, c` each time):
    # static input, static weights, static quantisers -> static int8 operands.  `pin` runs the chain once into the buffers of
    # a SLOT (k8, v8^T, column sums, the key-term table of the attention kernel) and keeps a device copy of the context next
    # to them; `match` recognises a later context as prepared — the same object with the same in-place version (no device
    # work), or ANY tensor holding the same bytes (one comparison kernel + one flag read back: the reference's samplers build
    # a fresh `torch.cat([uncond, c])` per step) — as long as the model's state token is the one the bytes were made under.
    # QDIFF_CTX_PINS slots (default 2: cond / uncond evaluated separately, two prompts alternating) are kept, least recently
    # used evicted; an entry LOCKED by a whole-step graph (sampling.DevicePLMS) is never evicted nor rewritten with other bytes.
    def _live(self):
        tok = self.token()
        if any(e["token"] != tok for e in self._pins):
            self._pins = [e for e in self._pins if e["token"] == tok or e["locked"]]
        return tok

    def match(self, context, by_value=True):
        """The entry prepared for `context`, or None.  by_value=False: identity + in-place version only (no device read-back:
        usable under stream capture)."""
        if not self._pins or not torch.is_tensor(context):
            return None
        tok = self._live()
        ver = engine.tensor_version(context)                       # None: an inference tensor, identity proves nothing
        cands = [e for e in reversed(self._pins) if e["token"] == tok and e["shape"] == tuple(context.shape)
                 and e["dtype"] == context.dtype and e["device"] == context.device and e["stream"] == engine.STREAM_DTYPE]
        if ver is not None:
            for e in cands:
                a = e["alias"].get(id(context))
                if a is not None and a[0]() is context and a[1] == ver:
                    return self._touch(e)
        if not by_value or not cands:
            return None
        if context.is_cuda:
            if torch.cuda.is_current_stream_capturing():
                return None
            flags = torch.stack([(context == e["copy"]).all() for e in cands]).tolist()      # ONE read-back for all slots
        else:
            flags = [torch.equal(context, e["copy"]) for e in cands]
        for e, same in zip(cands, flags):
            if same:
                self.value_matches += 1
                self._alias(e, context, ver)
                return self._touch(e)
        return None

    def _touch(self, e):
        if self._pins and self._pins[-1] is not e:
            self._pins.remove(e)
            self._pins.append(e)
        return e

    @staticmethod
    def _alias(e, context, ver):
        if ver is None:
            return
        if len(e["alias"]) > 64:
            e["alias"].clear()
        import weakref
        e["alias"][id(context)] = (weakref.ref(context), ver)

    def pin(self, context):
        """Run the chain for `context` into a slot (the slot of an entry holding the same bytes, else a free one, else the least
        recently used unlocked one) and select nothing: the next evaluation `match`es.  Returns the entry or None."""
        if not _CTX_PIN or not self._ready(context):
            return None
        tok = self._live()
        e = self.match(context)
        if e is not None:
            slot = e["slot"]                           # same bytes again: the entry (a handle lock_context may have handed out) is
        else:                                          # refreshed IN PLACE below, its lock count stays with it
            used = {p["slot"] for p in self._pins}
            free = [p for p in self._pins if not p["locked"]]
            if len(self._pins) >= _CTX_PINS and free:
                slot = free[0]["slot"]
                self._pins.remove(free[0])
            else:
                slot = next(i for i in range(len(used) + 1) if i not in used)
        out = self._work(context, tag=("pin", slot))
        for blk in self.members:
            att = blk.attn2
            k8, v8, vsum, _ = out[id(blk)]
            h = att.heads
            d = att.to_k.conv_plan().Cout // h
            ap = blk._attn_plan(att, float(att.scale), 1.0, context.device)
            kterm = None
            if engine.hip.attn_uses_keyterm(d, context.shape[1], ap.asym):
                kterm = engine.hip.attn_keyterm(k8, k8.shape[0], k8.shape[1], k8.shape[2], ap.prm, self._bufs_for(("pin-kterm", slot, id(blk)) + tuple(k8.shape[:2]),
                                                lambda: torch.empty(tuple(k8.shape[:2]), dtype=torch.int32, device=k8.device)))
            out[id(blk)] = (k8, v8, vsum, kterm)
        self._slot_gen[slot] = self._slot_gen.get(slot, 0) + 1
        copy = self._bufs_for(("pin-copy", slot, tuple(context.shape), context.dtype, context.device),
                              lambda: torch.empty(tuple(context.shape), dtype=context.dtype, device=context.device))
        copy.copy_(context.detach())
        # `stream`: the projections of the chain emitted rows of this type (fp32 / fp16 activation stream) — the codes of another
        # stream type differ at ties, so an entry serves evaluations of its own stream type only
        fresh = dict(slot=slot, gen=self._slot_gen[slot], token=tok, copy=copy, shape=tuple(context.shape), dtype=context.dtype,
                     device=context.device, alias={}, out=out, stream=engine.STREAM_DTYPE)
        if e is None:
            e = dict(fresh, locked=0)
            self._pins.append(e)
        else:
            e.update(fresh)
            self._touch(e)
        self._alias(e, context, engine.tensor_version(context))
        return e

    def unpin(self):
        """Forget every prepared context that no whole-step graph holds (a locked entry stays: its graph reads the buffers)."""
        self._pins = [e for e in self._pins if e["locked"]]
        self._sel = None

    def select(self, entry, context, owner="forward"):
        """The evaluation about to be issued reads `entry`'s operands for `context` (None: the per-evaluation branch)."""
        self._sel = (entry, context, owner) if entry is not None else None

    def pinned(self, context, deep=True):
        """`context` is prepared (by identity or by value) under the model's current state token."""
        return self.match(context) is not None

    def _bufs_for(self, key, make):
        bufs = self.__dict__.setdefault("_bufs", {})
        if key not in bufs:
            bufs[key] = make()
        return bufs[key]

    def register(self, blk):
        self.members.append(blk)
        blk.__dict__["_ctx_group"] = self

    def reset(self):
        """Start of a UNet evaluation: see EmbGroup.reset.  A branch that nobody joined (no cross-attention ran) is joined
        here so that no work is left dangling on the side stream."""
        if not self._joined and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._ctx, self._out, self._joined = None, None, True

    def begin(self, context, fork):
        """Forward pre-hook of the wrapped model.  QuantModel.forward has already selected the prepared entry for this very
        tensor (or none); a DIRECT call of the wrapped model (graph warm-up / capture, tools) selects by identity only — no
        device read-back here: this may run under stream capture, where only a LOCKED entry is taken (an unlocked one could
        be rewritten for another prompt while the captured graph keeps reading its buffers)."""
        sel = self._sel
        if sel is None or sel[1] is not context:
            self._sel = None
            e = self.match(context, by_value=False) if torch.is_tensor(context) else None
            if e is not None and context.is_cuda and torch.cuda.is_current_stream_capturing() and not e["locked"]:
                e = None
            if e is not None:
                self._sel = (e, context, "hook")
        if fork and torch.is_tensor(context):
            self.start(context)

    def start(self, context):
        """Fork the branch for this evaluation's context (no-op for a prepared context, for a context already started, for None,
        and whenever `_prepare` finds a module that is not ready for the integer path: `get` then answers None = in-line path)."""
        if context is not None and self._sel is not None and self._sel[1] is context:
            return
        if context is not None and self._ctx is not context:
            self._ctx, self._out = context, None
            self._prepare(context)

    def get(self, blk, context):
        """(k8, v8, vsum, kterm) of blk.attn2 for this context, or None (in-line path)."""
        sel = self._sel
        if sel is not None and sel[1] is context:
            return sel[0]["out"][id(blk)]
        self.start(context)
        if self._out is None:
            return None
        if not self._joined:
            torch.cuda.current_stream().wait_stream(self._side)
            self._joined = True
        return self._out[id(blk)]

    def _ready(self, context):
        if not torch.is_tensor(context) or context.dim() != 3 or len(self.members) < 2:
            return False
        mods = [m for blk in self.members for m in (blk.attn2.to_k, blk.attn2.to_v)]
        if not _int_mode(*mods) or any(m.split or not m.act_quantizer.inited for m in mods):
            return False
        return all(blk.attn2.use_act_quant and blk._attn_inited(blk.attn2) for blk in self.members)

    def _work(self, context, tag="eval"):
        """The chain itself, on the current stream: {id(blk): (k8, v8^T, vsum, None)} in buffers owned by `tag`."""
        self.chain_runs += 1
        B, S, Cc = context.shape
        dev = context.device
        # the fp32 row view of the context is made HERE, i.e. on the side stream when there is one: a converted /
        # compacted copy (fp16 CLIP output under autocast) then belongs to the side stream's allocator pool and
        # cannot be handed to a main-stream kernel while the branch still reads it
        ctx = context.reshape(B * S, Cc).float()
        if ctx.stride(1) != 1:
            ctx = ctx.contiguous()
        out = {}
        for blk in self.members:
            att = blk.attn2
            h = att.heads
            inner = att.to_k.conv_plan().Cout
            d = inner // h
            ap = blk._attn_plan(att, float(att.scale), 1.0, dev)
            Sp, dp = engine.pad32(S), engine.pad32(d)
            k8, v8, vsum = self._bufs_for((tag, id(blk), B, S, h, d),
                                          lambda: (torch.zeros((B * h, Sp, dp), dtype=torch.int8, device=dev),
                                                   torch.zeros((B * h, dp, Sp), dtype=torch.int8, device=dev),
                                                   torch.zeros((B * h, dp), dtype=torch.int32, device=dev)))
            for mod, which, buf in ((att.to_k, 1, k8), (att.to_v, 2, v8)):
                y = _linear_rows(mod, ctx)
                engine.heads_from_float(ap, which, y, B, S, h, d, (S * inner, inner, d, 1), buf, vsum)
            out[id(blk)] = (k8, v8, vsum, None)
        return out

    def _prepare(self, context):
        if not self._ready(context):
            return
        dev = context.device
        if dev.type != "cuda":
            self._out, self._joined = self._work(context), True
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        self._side.wait_stream(main)                  # after everything queued so far (incl. the previous evaluation's readers)
        context.record_stream(self._side)             # the caller's tensor is read by the branch: not reusable before it ends
        with torch.cuda.stream(self._side):
            self._out = self._work(context)
        self._joined = False

    def finish(self):
        """End of a UNet evaluation (forward hook of the model): a branch no cross-attention joined — every block took
        the in-line path, or the walk stopped early (calibration capture) — is joined NOW, so that no work dangles on the
        side stream past the evaluation, in particular not past the end of a HIP-graph capture."""
        if not self._joined and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
            self._joined = True
        if self._sel is not None and self._sel[2] == "hook":
            self._sel = None


def time_mlp(lin0, lin1, t_emb, act=F.silu):
    """`Linear -> SiLU -> Linear` on the sinusoid table (reference openaimodel.py:758-759 `time_embed`, ddim
    diffusion.py:318-320): two K6 launches on the integer path, the plain composition otherwise."""
    if (_int_mode(lin0, lin1) and lin0.kind == 'linear' and lin1.kind == 'linear' and not (lin0.split or lin1.split)
            and lin0.act_quantizer.inited and lin1.act_quantizer.inited and torch.is_tensor(t_emb) and t_emb.dim() == 2):
        p0, p1 = lin0.conv_plan(), lin1.conv_plan()
        if len(p0.segs) == 1 and len(p1.segs) == 1 and p0.pack.Cin % 16 == 0 and p1.pack.Cin % 16 == 0:      # (qd_temb_mlp streams 16-byte chunks of K)
            h = torch.empty((t_emb.shape[0], p0.Cout), dtype=torch.float32, device=t_emb.device)
            engine.hip.temb_mlp(t_emb.float(), False, [p0], [0], h)
            out = torch.empty((t_emb.shape[0], p1.Cout), dtype=torch.float32, device=t_emb.device)
            engine.hip.temb_mlp(h, True, [p1], [0], out)
            return out
    return lin1(act(lin0(t_emb)))


class _AttnQuant:
    """Mixin: cached AttnPlan for the four attention quantisers of a block."""

    def _attn_plan(self, owner, scale, prescale, device):
        qs = (owner.act_quantizer_q, owner.act_quantizer_k, owner.act_quantizer_v, owner.act_quantizer_w)
        key = tuple(engine.quantizer_key(q) for q in qs) + (scale, prescale)
        cache = owner.__dict__.setdefault("_attn_plan_cache", [None, None])
        if cache[0] != key:
            cache[0], cache[1] = key, engine.build_attn_plan(*qs, scale, prescale, device)
        return cache[1]


_REF_CACHE = {}


def reference_classes():
    """The reference's own UNet classes, when its `ldm` / `ddim` packages are importable (drop-in use inside the
    q-diffusion source tree: `import qdiff` resolves to this package, `ldm` / `ddim` to the reference's): they are
    rewritten exactly like this repo's classes.  Looked up lazily (and cached once found) so that the reference tree may
    be put on sys.path after this module was imported."""
    if _REF_CACHE.get("complete"):
        return _REF_CACHE
    out = {}
    try:
        from ldm.modules.diffusionmodules import openaimodel as ref_oai
        from ldm.modules import attention as ref_att
        out.update(ResBlock=ref_oai.ResBlock, AttentionBlock=ref_oai.AttentionBlock, QKMatMul=ref_oai.QKMatMul,
                   SMVMatMul=ref_oai.SMVMatMul, BasicTransformerBlock=ref_att.BasicTransformerBlock,
                   TimestepBlock=ref_oai.TimestepBlock, SpatialTransformer=ref_att.SpatialTransformer,
                   Upsample=ref_oai.Upsample, Downsample=ref_oai.Downsample, UNetModel=ref_oai.UNetModel,
                   TimestepEmbedSequential=ref_oai.TimestepEmbedSequential)
    except Exception:  # noqa: BLE001 - optional dependency
        pass
    try:
        from ddim.models import diffusion as ref_ddim
        out.update(ResnetBlock=ref_ddim.ResnetBlock, AttnBlock=ref_ddim.AttnBlock, DdimModel=ref_ddim.Model,
                   DdimUpsample=ref_ddim.Upsample, DdimDownsample=ref_ddim.Downsample)
    except Exception:  # noqa: BLE001
        pass
    if out:
        _REF_CACHE.update(out)
        _REF_CACHE["complete"] = "ResBlock" in out and "ResnetBlock" in out
    return out or _REF_CACHE


# ------------------------------------------------------------------------------------------------
# base
# ------------------------------------------------------------------------------------------------
class BaseQuantBlock(nn.Module):
    """reference quant_block.py:20-41"""

    def __init__(self, act_quant_params: dict = {}):
        super().__init__()
        self.use_weight_quant = False
        self.use_act_quant = False
        self.act_quantizer = UniformAffineQuantizer(**act_quant_params)   # never initialised: no state-dict keys
        self.activation_function = StraightThrough()
        self.ignore_reconstruction = False

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        engine.bump_state()
        for m in self.modules():
            if isinstance(m, QuantModule):
                m.set_quant_state(weight_quant, act_quant)


# ------------------------------------------------------------------------------------------------
# LDM / SD residual block  (reference quant_block.py:44-111)
# ------------------------------------------------------------------------------------------------
class QuantResBlock(BaseQuantBlock, ldm_unet.TimestepBlock):
    def __init__(self, res, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        for name in ("channels", "emb_channels", "dropout", "out_channels", "use_conv", "use_checkpoint",
                     "use_scale_shift_norm", "in_layers", "updown", "h_upd", "x_upd", "emb_layers", "out_layers",
                     "skip_connection"):
            setattr(self, name, getattr(res, name))

    qd_takes_out_slot = True

    def forward(self, x, emb=None, split=0, out_slot=None):
        """out_slot (engine-internal, optional): engine.CatSlot side that receives the block's output."""
        # the split argument is only forwarded until the skip connection has recorded it (reference :75-81)
        if split != 0 and self.skip_connection.split == 0:
            return self._forward(x, emb, split, out_slot)
        return self._forward(x, emb, 0, out_slot)

    def _forward(self, x, emb, split=0, out_slot=None):
        if emb is None:
            x, emb = x
        assert x.shape[2] == x.shape[3]
        H_ = x.shape[2]
        conv1, conv2 = self.in_layers[-1], self.out_layers[-1]
        if (_int_mode(conv1, conv2, self.emb_layers[-1]) and conv1.split == 0 and conv2.split == 0 and not _dropout_live(self)
                and (not self.updown or (H_ % 2 == 0 and not getattr(self.h_upd, "use_conv", False)))):
            return self._forward_int(x, emb, split, conv1, conv2, out_slot)
        return self._forward_sim(x, emb, split)

    def _forward_sim(self, x, emb, split=0):
        if self.updown:
            h = self.in_layers[:-1](x)
            h, x = self.h_upd(h), self.x_upd(x)
            h = self.in_layers[-1](h)
        else:
            h = self.in_layers(x)
        e = self.emb_layers(emb).type(h.dtype)
        while e.dim() < h.dim():
            e = e[..., None]
        live = _dropout_live(self)      # the dropout mask follows MEMORY order: NCHW, as the reference's tensors are laid out
        if self.use_scale_shift_norm:
            scale, shift = th.chunk(e, 2, dim=1)
            h = self.out_layers[0](h) * (1 + scale) + shift
            h = self.out_layers[1:](h.contiguous() if live else h)
        else:
            h = self.out_layers((h + e).contiguous() if live else h + e)
        if split != 0:
            return self.skip_connection(x, split=split) + h
        return self.skip_connection(x) + h

    def _skip_plan(self, split, C):
        """ConvPlan of the 1x1 skip connection when its int8 input rows can be produced by the GroupNorm pass that
        already reads x (engine.groupnorm_silu_quant(raw_plan=...)), else None (the module quantises x itself)."""
        sk = self.skip_connection
        if not (_FUSE_SKIP_QUANT and isinstance(sk, QuantModule) and sk.kind == 'conv2d' and _int_mode(sk)):
            return None
        if split != 0 and sk.split == 0:
            return None                                   # the first call records the split: module path
        if not all(q.inited for q in sk._act_quantizers()) or sk._geometry() != (1, 1, 1, 0):
            return None
        plan = sk.conv_plan()
        return plan if engine.raw_quant_segs(plan, C) is not None else None

    def _forward_int(self, x, emb, split, conv1, conv2, out_slot=None):
        """Reference :83-111 on the integer engine.  Plain blocks: GN.SiLU -> int8 (one pass; the skip connection's rows in
        the same pass), conv1 with the embedding as a row bias and the next norm's statistics in its epilogue, GN.SiLU ->
        int8, conv2 with the residual in its epilogue.  `updown` blocks (:84-90) resample between the first norm and the
        first convolution: nearest-2x commutes with the quantiser, so the SMALL map is quantised and its int8 rows are
        replicated; the 2x2 average does not, so the norm writes fp32 rows, torch averages them (F.avg_pool2d, as the
        reference) and the small map is quantised.  `use_scale_shift_norm` blocks (:99-103) hand the two halves of the
        embedding projection to the second norm as a per-sample modulation (qd_groupnorm_mod_silu_quant) instead of
        adding the projection to h."""
        B, C, H, W = x.shape
        S = H * W
        rows = _nhwc_rows(x)
        ident = isinstance(self.skip_connection, nn.Identity)
        up = self.updown and not hasattr(self.h_upd, "op")            # Upsample has .conv / nothing, Downsample has .op
        down = self.updown and not up
        skp = None if (ident or self.updown) else self._skip_plan(split, C)
        if down:
            y = engine.groupnorm_silu_quant(rows, B, S, C, self.in_layers[0], True, want_float=True,
                                            part=getattr(rows, "qd_gn_part", None))[1]
            hp = self.h_upd(_rows_to_nchw(y, B, H, W))
            x = self.x_upd(x)
            H, W = H // 2, W // 2
            S = H * W
            conv1._init_act_quantizers(hp)
            hp = _nhwc_rows(hp)
            xq = engine.quantize_rows(hp, conv1.conv_plan(), 1, C, B * S, (0, 1, hp.stride(0)))
            rows = _nhwc_rows(x)
        elif skp is not None:
            xq, skq = _gn_silu_to(conv1, rows, B, S, C, self.in_layers[0], raw_plan=skp)
        else:
            xq = _gn_silu_to(conv1, rows, B, S, C, self.in_layers[0])
        if up:
            xq = xq.view(B, H, 1, W, 1, -1).expand(B, H, 2, W, 2, xq.shape[1]).reshape(B * 4 * S, xq.shape[1])
            x = self.x_upd(x)
            H, W = 2 * H, 2 * W
            S = H * W
            rows = _nhwc_rows(x)
        grp = self.__dict__.get("_emb_group")
        e = grp.get(self, emb) if grp is not None else None           # all blocks' projections in one launch (K6)
        if e is None:
            e = self.emb_layers(emb).float().contiguous()             # SiLU + integer linear -> [B, Cout] (2 Cout: scale | shift)
        if self.use_scale_shift_norm:
            h = conv1.forward_codes(xq, B, H, W, gn_stats=True)
            if e.stride(1) != 1:
                e = e.contiguous()
            hq = _gn_silu_to(conv2, h, B, S, self.out_channels, self.out_layers[0], mod=e)
        else:
            h = conv1.forward_codes(xq, B, H, W, rowbias=e, gn_stats=True)
            hq = _gn_silu_to(conv2, h, B, S, self.out_channels, self.out_layers[0])
        if ident:
            res = rows
        elif skp is not None:
            res = self.skip_connection.forward_codes(skq, B, H, W)
        else:
            sk = self.skip_connection(x, split=split) if split != 0 else self.skip_connection(x)
            res = _nhwc_rows(sk)
        out = conv2.forward_codes(hq, B, H, W, residual=res, gn_stats=True, slot=out_slot)   # the next block normalises this
        return _rows_to_nchw(out, B, H, W)


# ------------------------------------------------------------------------------------------------
# LDM attention matmuls  (reference quant_block.py:114-160)
# ------------------------------------------------------------------------------------------------
class QuantQKMatMul(BaseQuantBlock):
    def __init__(self, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.scale = None
        self.use_act_quant = False
        self.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)

    def forward(self, q, k):
        if (self.use_act_quant and not torch.is_grad_enabled() and not engine.SIMULATE and q.dim() == 3 and q.shape[1] % 4 == 0
                and _aq_ready(self.act_quantizer_q, self.act_quantizer_k) and not self.act_quantizer_q.running_stat
                and self.act_quantizer_q.n_bits <= 8 and self.act_quantizer_k.n_bits <= 8):
            # standalone use (this module called outside the fused QuantAttentionBlock path): integer engine,
            # exact int32 contraction, the T x S score matrix is materialised as the API requires
            return engine.qk_matmul_int(self.act_quantizer_q, self.act_quantizer_k, q.float(), k.float(), float(self.scale))
        if self.use_act_quant:
            q, k = self.act_quantizer_q(q * self.scale), self.act_quantizer_k(k * self.scale)
        else:
            q, k = q * self.scale, k * self.scale
        return th.einsum("bct,bcs->bts", q, k)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_act_quant = act_quant


class QuantSMVMatMul(BaseQuantBlock):
    def __init__(self, act_quant_params: dict = {}, sm_abit=8):
        super().__init__(act_quant_params)
        self.use_act_quant = False
        self.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
        params_w = act_quant_params.copy()
        params_w['n_bits'] = sm_abit
        params_w['symmetric'] = False
        params_w['always_zero'] = True
        self.act_quantizer_w = UniformAffineQuantizer(**params_w)

    def forward(self, weight, v):
        if (self.use_act_quant and not torch.is_grad_enabled() and not engine.SIMULATE and v.dim() == 3 and v.shape[1] % 4 == 0
                and _aq_ready(self.act_quantizer_v, self.act_quantizer_w) and not self.act_quantizer_v.running_stat
                and not self.act_quantizer_w.running_stat and self.act_quantizer_v.n_bits <= 8):
            return engine.smv_matmul_int(self.act_quantizer_w, self.act_quantizer_v, weight, v.float())
        if self.use_act_quant:
            weight, v = self.act_quantizer_w(weight), self.act_quantizer_v(v)
        return th.einsum("bts,bcs->bct", weight, v)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_act_quant = act_quant


class QuantAttentionBlock(BaseQuantBlock, _AttnQuant):
    """LDM AttentionBlock (reference quant_block.py:163-187).  With quantised activations and this
    repo's own QKVAttentionLegacy (whose two matmuls were swapped for QuantQKMatMul / QuantSMVMatMul),
    the whole qkv -> attention chain runs fused on the integer engine."""

    def __init__(self, attn, act_quant_params: dict = {}, sm_abit: int = 8, quant_matmuls: bool = False):
        super().__init__(act_quant_params)
        self.channels = attn.channels
        self.num_heads = attn.num_heads
        self.use_checkpoint = attn.use_checkpoint
        self.norm = attn.norm
        self.qkv = attn.qkv
        self.attention = attn.attention
        self.proj_out = attn.proj_out
        # reference quant_model.py:45-61 + quant_block.py:389-401: with quantised activations the reference does NOT wrap the
        # AttentionBlock (its qkv / proj_out QuantModules and the two Quant*MatMul blocks are separate units); calibration
        # walks into this wrapper in that mode (qdiff/calibrate.recon_model) so that the units are the reference's
        self.quant_matmuls = bool(quant_matmuls)
        if quant_matmuls:
            # quantised-activation mode: the two matmul modules inside QKVAttentionLegacy become their
            # quantised counterparts (what the reference's recursion does, quant_model.py:45-61)
            # (this repo's QKMatMul / SMVMatMul or the reference's: same names, same role)
            if type(getattr(self.attention, "qkv_matmul", None)).__name__ == "QKMatMul":
                self.attention.qkv_matmul = QuantQKMatMul(act_quant_params)
            if type(getattr(self.attention, "smv_matmul", None)).__name__ == "SMVMatMul":
                self.attention.smv_matmul = QuantSMVMatMul(act_quant_params, sm_abit=sm_abit)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        super().set_quant_state(weight_quant, act_quant)
        for m in (getattr(self.attention, "qkv_matmul", None), getattr(self.attention, "smv_matmul", None)):
            if isinstance(m, (QuantQKMatMul, QuantSMVMatMul)):
                m.set_quant_state(weight_quant, act_quant)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        return self._forward(x, out_slot)

    def _fusable(self):
        att = self.attention
        qk, smv = getattr(att, "qkv_matmul", None), getattr(att, "smv_matmul", None)
        return (isinstance(qk, QuantQKMatMul) and isinstance(smv, QuantSMVMatMul) and qk.use_act_quant
                and smv.use_act_quant and _int_mode(self.qkv, self.proj_out)
                and _aq_ready(qk.act_quantizer_q, qk.act_quantizer_k, smv.act_quantizer_v, smv.act_quantizer_w))

    def _forward(self, x, out_slot=None):
        b, c, *spatial = x.shape
        if x.dim() == 4 and self._fusable():
            T = x.shape[2] * x.shape[3]
            plans = self.qkv.head_plans(self.num_heads) if (QKV_HEADS and T % 128 == 0 and self.qkv.act_quantizer.inited) else None
            if plans is not None and all(engine.heads_fusable(p, T, self.num_heads) for p in plans):
                return self._forward_heads(x, plans, out_slot)
            return self._forward_int(x, out_slot)
        xf = x.reshape(b, c, -1)
        h = self.proj_out(self.attention(self.qkv(self.norm(xf))))
        return (xf + h).reshape(b, c, *spatial)

    def _forward_int(self, x, out_slot=None):
        B, C, H, W = x.shape
        T, nh = H * W, self.num_heads
        d = C // nh
        rows = _nhwc_rows(x)
        xq = _gn_silu_to(self.qkv, rows, B, T, C, self.norm, silu=False)
        qkv = self.qkv.forward_codes(xq, B, 1, T)                     # [B*T, 3C]; channel = head*3d + {q,k,v}*d + i
        qk, smv = self.attention.qkv_matmul, self.attention.smv_matmul
        holder = self.__dict__.setdefault("_aq_view", type("V", (), {})())
        holder.act_quantizer_q, holder.act_quantizer_k = qk.act_quantizer_q, qk.act_quantizer_k
        holder.act_quantizer_v, holder.act_quantizer_w = smv.act_quantizer_v, smv.act_quantizer_w
        scale = float(qk.scale) if qk.scale is not None else d ** -0.25
        ap = self._attn_plan(holder, 1.0, scale, x.device)
        ld = qkv.stride(0)
        strides = (T * ld, ld, 3 * d, 1)
        att = engine.attention(ap, qkv, qkv[:, d:], qkv[:, 2 * d:], B, T, T, nh, d, strides, strides, strides)
        out = _linear_like_conv1d(self.proj_out, att, B, T, residual=rows, gn_stats=True, slot=out_slot)
        return _rows_to_nchw(out, B, H, W)

    def _forward_heads(self, x, plans, out_slot=None):
        """The same block with the qkv projection run as three GEMMs (QuantModule.head_plans) whose epilogues write the
        attention operand bytes: no fp32 [B*T, 3C] round trip, no separate head quantisers; the attention epilogue
        quantises for proj_out where its quantiser is ready."""
        B, C, H, W = x.shape
        T, nh = H * W, self.num_heads
        d = C // nh
        rows = _nhwc_rows(x)
        xq = _gn_silu_to(self.qkv, rows, B, T, C, self.norm, silu=False)
        qk, smv = self.attention.qkv_matmul, self.attention.smv_matmul
        holder = self.__dict__.setdefault("_aq_view", type("V", (), {})())
        holder.act_quantizer_q, holder.act_quantizer_k = qk.act_quantizer_q, qk.act_quantizer_k
        holder.act_quantizer_v, holder.act_quantizer_w = smv.act_quantizer_v, smv.act_quantizer_w
        scale = float(qk.scale) if qk.scale is not None else d ** -0.25
        ap = self._attn_plan(holder, 1.0, scale, x.device)
        q8, k8, v8, vsum = engine.head_buffers(x.device, B * nh, T, T, d)
        vsum = engine.vsum_slice(id(self), x.device, tuple(vsum.shape))
        engine.project_heads_group([(plan, xq, which, buf) for which, (plan, buf) in enumerate(zip(plans, (q8, k8, v8)))], B, T, nh, ap, vsum)
        po = self.proj_out
        if po.act_quantizer.inited and po.split == 0 and po.conv_plan().ldx == C and len(po.conv_plan().segs) == 1:
            o8 = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, T, nh, d, out_plan=po.conv_plan())
            out = po.forward_codes(o8, B, 1, T, residual=rows, gn_stats=True, slot=out_slot)
        else:
            att = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, T, nh, d)
            out = _linear_like_conv1d(po, att, B, T, residual=rows, gn_stats=True, slot=out_slot)
        return _rows_to_nchw(out, B, H, W)


def _linear_like_conv1d(mod, rows, B, T, residual=None, gn_stats=False, slot=None):
    """conv1d(k=1) QuantModule applied to token rows [B*T, C]."""
    mod._init_act_quantizers(rows.view(B, T, -1).permute(0, 2, 1))
    plan = mod.conv_plan()
    M, K = rows.shape
    xq = engine.quantize_rows(rows, plan, 1, K, M, (0, rows.stride(1), rows.stride(0)))
    return engine.conv_forward(plan, xq, B, 1, T, 1, T, residual=residual, gn_stats=gn_stats, slot=slot)


# ------------------------------------------------------------------------------------------------
# SD transformer block  (reference quant_block.py:190-282)
# ------------------------------------------------------------------------------------------------
def cross_attn_forward(self, x, context=None, mask=None):
    """Bound onto attn1/attn2 (as the reference does, quant_block.py:254-255): simulation-tier
    forward with the four activation quantisers; the fused integer path lives in
    QuantBasicTransformerBlock._attn_int."""
    h = self.heads
    context = x if context is None else context
    q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
    q, k, v = (ldm_unet._split_heads(t, h) for t in (q, k, v))
    if self.use_act_quant:
        q, k = self.act_quantizer_q(q), self.act_quantizer_k(k)
    sim = th.einsum('b i d, b j d -> b i j', q, k) * self.scale
    if mask is not None:
        mask = mask.reshape(mask.shape[0], -1)[:, None, :].repeat_interleave(h, dim=0)
        sim.masked_fill_(~mask, -th.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    if self.use_act_quant:
        attn, v = self.act_quantizer_w(attn), self.act_quantizer_v(v)
    out = th.einsum('b i j, b j d -> b i d', attn, v)
    return self.to_out(ldm_unet._merge_heads(out, h))


class QuantBasicTransformerBlock(BaseQuantBlock, _AttnQuant):
    def __init__(self, tran, act_quant_params: dict = {}, sm_abit: int = 8):
        super().__init__(act_quant_params)
        self.attn1, self.ff, self.attn2 = tran.attn1, tran.ff, tran.attn2
        self.norm1, self.norm2, self.norm3 = tran.norm1, tran.norm2, tran.norm3
        self.checkpoint = tran.checkpoint
        params_w = act_quant_params.copy()
        params_w['n_bits'] = sm_abit
        params_w['always_zero'] = True
        for att in (self.attn1, self.attn2):
            att.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
            att.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)
            att.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
            att.act_quantizer_w = UniformAffineQuantizer(**params_w)
            att.forward = MethodType(cross_attn_forward, att)
            att.use_act_quant = False

    def forward(self, x, context=None, out_plan=None):
        """out_plan (engine-internal, optional): ConvPlan of the module that consumes this block's output and nothing
        else (SpatialTransformer.proj_out).  When the block runs on the integer path and the shape allows it, the FF
        output GEMM then returns that consumer's int8 input rows [B*T][ldx] instead of the fp32 tokens."""
        return self._forward(x, context, out_plan)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.attn1.use_act_quant = act_quant
        self.attn2.use_act_quant = act_quant
        super().set_quant_state(weight_quant, act_quant)

    def _attn_inited(self, att):
        return _aq_ready(att.act_quantizer_q, att.act_quantizer_k, att.act_quantizer_v, att.act_quantizer_w)

    def _forward(self, x, context=None, out_plan=None):
        if context is None and isinstance(x, (tuple, list)):
            x, context = x
        a1, a2 = self.attn1, self.attn2
        mods = [a1.to_q, a1.to_k, a1.to_v, a1.to_out[0], a2.to_q, a2.to_k, a2.to_v, a2.to_out[0], self.ff.net[-1]]
        glu = isinstance(self.ff.net[0], ldm_unet.GEGLU) or type(self.ff.net[0]).__name__ == "GEGLU"
        if (glu and a1.use_act_quant and a2.use_act_quant and not _dropout_live(self) and _int_mode(*mods, self.ff.net[0].proj)
                and self._attn_inited(a1) and self._attn_inited(a2)):
            return self._forward_int(x, context, out_plan)
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x

    def _attn_int(self, att, rows, B, T, C, ln, ctx_rows, S, kv=None, pre_attention=None):
        """norm -> q/k/v projections -> fused quantised attention -> to_out (+ residual rows).
        kv: (k8, v8, vsum, kterm) prepared ahead of time for this block's context (ContextKV), else they are computed here."""
        h = att.heads
        ap = self._attn_plan(att, float(att.scale), 1.0, rows.device)
        if ctx_rows is None:
            xq, xk, xv = _ln_to([att.to_q, att.to_k, att.to_v], rows, B * T, C, ln)
            S = T
        else:
            (xq,) = _ln_to([att.to_q], rows, B * T, C, ln)
            xk = xv = None
        inner = att.to_q.conv_plan().Cout
        d = inner // h
        q8, k8, v8, vsum = engine.head_buffers(rows.device, B * h, T, S, d)
        if kv is None:
            vsum = engine.vsum_slice(id(att), rows.device, tuple(vsum.shape))   # this block's own slice of the per-evaluation arena
        kterm = None
        if kv is not None:
            k8, v8, vsum, kterm = kv
            if k8.shape[0] != B * h or v8.shape[0] != B * h:
                # prepared / branch operands of another batch size (a prepared context handed to a latent batch it was not made for)
                raise engine.hip.HipEngineError(f"cross-attention operands were prepared for {k8.shape[0] // h} samples, the latents have {B}")

        def operand(mod, codes, which, n_tok, buf):
            # projection -> attention operand bytes: inside the GEMM epilogue when the shape allows it, else
            # fp32 projection + qd_quantize_heads (ragged token counts; the 77 context tokens of cross-attention)
            if codes is not None and engine.heads_fusable(mod.conv_plan(), n_tok, h):
                engine.project_heads(mod.conv_plan(), codes, B, n_tok, h, ap, which, buf, vsum)
                return
            y = mod.forward_codes(codes, 1, 1, B * n_tok) if codes is not None else _linear_rows(mod, ctx_rows)
            engine.heads_from_float(ap, which, y, B, n_tok, h, d, (n_tok * inner, inner, d, 1), buf, vsum)

        if (kv is None and ctx_rows is None
                and all(engine.heads_fusable(m.conv_plan(), T, h) for m in (att.to_q, att.to_k, att.to_v))):
            # self-attention: the three projections read rows of one LayerNorm and have one shape -> one grouped launch
            engine.project_heads_group([(att.to_q.conv_plan(), xq, 0, q8), (att.to_k.conv_plan(), xk, 1, k8),
                                        (att.to_v.conv_plan(), xv, 2, v8)], B, T, h, ap, vsum)
        else:
            operand(att.to_q, xq, 0, T, q8)
            if kv is None:
                operand(att.to_k, xk, 1, S, k8)
                operand(att.to_v, xv, 2, S, v8)
        if pre_attention is not None:
            pre_attention()                     # the next launch on this stream is the attention kernel
        out_lin = att.to_out[0]
        if out_lin.act_quantizer.inited and out_lin.conv_plan().ldx == inner and len(out_lin.conv_plan().segs) == 1:
            # the attention epilogue quantises its output for to_out[0]: no fp32 round trip
            o8 = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, S, h, d, out_plan=out_lin.conv_plan(), kterm=kterm)
            return out_lin.forward_codes(o8, 1, 1, B * T, residual=rows)
        o = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, S, h, d, kterm=kterm)
        return _linear_rows(out_lin, o, residual=rows)

    def _forward_int(self, x, context, out_plan=None):
        B, T, C = x.shape
        rows = x.reshape(B * T, C)
        if rows.stride(1) != 1 or rows.stride(0) != C:
            rows = rows.contiguous()
        grp = self.__dict__.get("_ctx_group")
        fork = (lambda: grp.start(context)) if (grp is not None and context is not None and _CTX_FORK == "attn") else None
        rows = self._attn_int(self.attn1, rows, B, T, C, self.norm1, None, T, pre_attention=fork)
        if context is None:
            rows = self._attn_int(self.attn2, rows, B, T, C, self.norm2, None, T)
        else:
            S = context.shape[1]
            kv = grp.get(self, context) if grp is not None else None
            ctx = None
            if kv is None:
                ctx = context.reshape(B * S, context.shape[2]).float()
                if ctx.stride(1) != 1:
                    ctx = ctx.contiguous()
            rows = self._attn_int(self.attn2, rows, B, T, C, self.norm2, ctx if kv is None else rows, S, kv=kv)
        return self._ff_int(rows, B, T, C, out_plan)

    def _ff_int(self, rows, B, T, C, out_plan=None):
        """norm3 -> GEGLU projection -> FF output Linear (+ residual rows): the third sub-layer (attention.py:229-231)."""
        proj, ff_out = self.ff.net[0].proj, self.ff.net[-1]
        (h8,) = _ln_to([proj], rows, B * T, C, self.norm3)
        gplan = proj.geglu_plan() if ff_out.act_quantizer.inited else None
        if gplan is not None:
            # fused epilogue: value*gelu(gate) is quantised for ff_out inside the projection kernel
            g8 = engine.conv_forward_geglu(gplan, h8, B * T, ff_out.conv_plan())
        else:
            hcat = proj.forward_codes(h8, 1, 1, B * T)                # [M, 2F]
            Fdim = hcat.shape[1] // 2
            if not ff_out.act_quantizer.inited:
                ff_out._init_act_quantizers(hcat[:, :Fdim] * F.gelu(hcat[:, Fdim:]))
            g8 = engine.geglu_quant(hcat, B * T, Fdim, ff_out.conv_plan())
        if out_plan is not None and engine.rows_i8_fusable(ff_out.conv_plan(), out_plan, T):
            # FF output + residual quantised for the consumer inside the GEMM epilogue: the fp32 tokens are never written
            return engine.linear_to_rows_i8(ff_out.conv_plan(), g8, B, T, out_plan, residual=rows)
        rows = ff_out.forward_codes(g8, 1, 1, B * T, residual=rows)
        return rows.view(B, T, C)


# ------------------------------------------------------------------------------------------------
# DDIM (CIFAR) blocks  (reference quant_block.py:286-386)
# ------------------------------------------------------------------------------------------------
class QuantResnetBlock(BaseQuantBlock):
    def __init__(self, res, act_quant_params: dict = {}):
        super().__init__(act_quant_params)
        self.in_channels, self.out_channels = res.in_channels, res.out_channels
        self.use_conv_shortcut = res.use_conv_shortcut
        self.norm1, self.conv1, self.temb_proj = res.norm1, res.conv1, res.temb_proj
        self.norm2, self.dropout, self.conv2 = res.norm2, res.dropout, res.conv2
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = res.conv_shortcut
            else:
                self.nin_shortcut = res.nin_shortcut

    qd_takes_out_slot = True

    def forward(self, x, temb=None, split=0, out_slot=None):
        if temb is None:
            x, temb = x
        if _int_mode(self.conv1, self.conv2, self.temb_proj) and not _dropout_live(self):
            return self._forward_int(x, temb, split, out_slot)
        h = self.conv1(ddim_unet.nonlinearity(self.norm1(x)))
        h = h + self.temb_proj(ddim_unet.nonlinearity(temb))[:, :, None, None]
        h = ddim_unet.nonlinearity(self.norm2(h))
        if _dropout_live(self):
            h = h.contiguous()          # the dropout mask follows MEMORY order: NCHW, as the reference's tensors are laid out
        h = self.conv2(self.dropout(h))
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x, split=split)
        return x + h

    def _skip_plan(self, split, C):
        """see QuantResBlock._skip_plan: the 1x1 `nin_shortcut` reads the tensor norm1 normalises"""
        if self.in_channels == self.out_channels or self.use_conv_shortcut:
            return None
        sk = self.nin_shortcut
        if not (_FUSE_SKIP_QUANT and isinstance(sk, QuantModule) and sk.kind == 'conv2d' and _int_mode(sk)):
            return None
        if (split != 0 and sk.split != split) or not all(q.inited for q in sk._act_quantizers()) or sk._geometry() != (1, 1, 1, 0):
            return None
        plan = sk.conv_plan()
        return plan if engine.raw_quant_segs(plan, C) is not None else None

    def _forward_int(self, x, temb, split, out_slot=None):
        B, C, H, W = x.shape
        S = H * W
        rows = _nhwc_rows(x)
        skp = self._skip_plan(split, C)
        if skp is not None:
            xq, skq = _gn_silu_to(self.conv1, rows, B, S, C, self.norm1, raw_plan=skp)
        else:
            xq = _gn_silu_to(self.conv1, rows, B, S, C, self.norm1)
        grp = self.__dict__.get("_emb_group")
        e = grp.get(self, temb) if grp is not None else None
        if e is None:
            e = self.temb_proj(ddim_unet.nonlinearity(temb)).float().contiguous()
        h = self.conv1.forward_codes(xq, B, H, W, rowbias=e, gn_stats=True)
        hq = _gn_silu_to(self.conv2, h, B, S, self.out_channels, self.norm2)
        if skp is not None:
            res = self.nin_shortcut.forward_codes(skq, B, H, W)
        elif self.in_channels != self.out_channels:
            sk = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x, split=split)
            res = _nhwc_rows(sk)
        else:
            res = rows
        out = self.conv2.forward_codes(hq, B, H, W, residual=res, gn_stats=True, slot=out_slot)
        return _rows_to_nchw(out, B, H, W)


class QuantAttnBlock(BaseQuantBlock, _AttnQuant):
    def __init__(self, attn, act_quant_params: dict = {}, sm_abit=8):
        super().__init__(act_quant_params)
        self.in_channels = attn.in_channels
        self.norm, self.q, self.k, self.v, self.proj_out = attn.norm, attn.q, attn.k, attn.v, attn.proj_out
        self.act_quantizer_q = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_k = UniformAffineQuantizer(**act_quant_params)
        self.act_quantizer_v = UniformAffineQuantizer(**act_quant_params)
        params_w = act_quant_params.copy()
        params_w['n_bits'] = sm_abit
        self.act_quantizer_w = UniformAffineQuantizer(**params_w)

    qd_takes_out_slot = True

    def forward(self, x, out_slot=None):
        if (self.use_act_quant and _int_mode(self.q, self.k, self.v, self.proj_out)
                and _aq_ready(self.act_quantizer_q, self.act_quantizer_k, self.act_quantizer_v, self.act_quantizer_w)):
            return self._forward_int(x, out_slot)
        hn = self.norm(x)
        q, k, v = self.q(hn), self.k(hn), self.v(hn)
        b, c, h, w = q.shape
        q = q.reshape(b, c, h * w).permute(0, 2, 1)
        k = k.reshape(b, c, h * w)
        if self.use_act_quant:
            q, k = self.act_quantizer_q(q), self.act_quantizer_k(k)
        w_ = F.softmax(th.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
        v = v.reshape(b, c, h * w)
        w_ = w_.permute(0, 2, 1)
        if self.use_act_quant:
            v, w_ = self.act_quantizer_v(v), self.act_quantizer_w(w_)
        out = th.bmm(v, w_).reshape(b, c, h, w)
        return x + self.proj_out(out)

    def _forward_int(self, x, out_slot=None):
        B, C, H, W = x.shape
        T = H * W
        rows = _nhwc_rows(x)
        # one GroupNorm, three consumers with their own act quantisers
        ws_plan = None
        _, y = engine.groupnorm_silu_quant(rows, B, T, C, self.norm, False, plan=ws_plan, want_float=True,
                                           part=getattr(rows, "qd_gn_part", None))
        ap = self._attn_plan(self, int(C) ** (-0.5), 1.0, x.device)
        po = self.proj_out
        if QKV_HEADS and T % 128 == 0:
            # q / k / v write the attention operand bytes from their epilogues (one head as wide as the layer), the attention
            # epilogue quantises for proj_out: no fp32 projections, no qd_quantize_heads, no zero fills
            for m in (self.q, self.k, self.v):
                m._init_act_quantizers(y)
            plans = [m.conv_plan() for m in (self.q, self.k, self.v)]
            if all(engine.heads_fusable(p, T, 1) and p.Cout == C for p in plans):
                q8, k8, v8, vsum = engine.head_buffers(x.device, B, T, T, C)
                vsum = engine.vsum_slice(id(self), x.device, tuple(vsum.shape))
                members = [(plan, engine.quantize_rows(y, plan, 1, C, B * T, (0, y.stride(1), y.stride(0))), which, buf)
                           for which, (plan, buf) in enumerate(zip(plans, (q8, k8, v8)))]
                engine.project_heads_group(members, B, T, 1, ap, vsum)
                if po.act_quantizer.inited and po.split == 0 and po.conv_plan().ldx == C and len(po.conv_plan().segs) == 1:
                    o8 = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, T, 1, C, out_plan=po.conv_plan())
                    out = po.forward_codes(o8, B, H, W, residual=rows, gn_stats=True, slot=out_slot)
                else:
                    o = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, T, 1, C)
                    out = _linear_like_conv2d(po, o, B, H, W, residual=rows, gn_stats=True, slot=out_slot)
                return _rows_to_nchw(out, B, H, W)
        q = _linear_like_conv2d(self.q, y, B, H, W)
        k = _linear_like_conv2d(self.k, y, B, H, W)
        v = _linear_like_conv2d(self.v, y, B, H, W)
        st = (T * C, C, C, 1)
        o = engine.attention(ap, q, k, v, B, T, T, 1, C, st, st, st)
        out = _linear_like_conv2d(self.proj_out, o, B, H, W, residual=rows, gn_stats=True, slot=out_slot)
        return _rows_to_nchw(out, B, H, W)


def _linear_like_conv2d(mod, rows, B, H, W, residual=None, gn_stats=False, slot=None):
    """1x1 Conv2d QuantModule applied to channels-last rows."""
    mod._init_act_quantizers(rows)
    plan = mod.conv_plan()
    M, K = rows.shape
    xq = engine.quantize_rows(rows, plan, 1, K, M, (0, rows.stride(1), rows.stride(0)))
    return engine.conv_forward(plan, xq, B, H, W, H, W, residual=residual, gn_stats=gn_stats, slot=slot)


# ------------------------------------------------------------------------------------------------
# dispatch table  (reference quant_block.py:389-401)
# ------------------------------------------------------------------------------------------------
_REF_RESBLOCK = {}


def _quant_resblock_for_reference(ref):
    """The reference's TimestepEmbedSequential hands `emb` only to instances of ITS TimestepBlock (openaimodel.py:80-88):
    the block that replaces a reference ResBlock must inherit from it (QuantResBlock itself inherits this repo's)."""
    base = ref["TimestepBlock"]
    cls = _REF_RESBLOCK.get(base)
    if cls is None:
        cls = _REF_RESBLOCK[base] = type("QuantResBlock", (QuantResBlock, base), {"__module__": QuantResBlock.__module__})
    return cls


def get_specials(quant_act=False):
    specials = {
        ldm_unet.ResBlock: QuantResBlock,
        ldm_unet.BasicTransformerBlock: QuantBasicTransformerBlock,
        ddim_unet.ResnetBlock: QuantResnetBlock,
        ddim_unet.AttnBlock: QuantAttnBlock,
    }
    ref = reference_classes()
    if "ResBlock" in ref:
        specials[ref["ResBlock"]] = _quant_resblock_for_reference(ref)
    for name, target in (("BasicTransformerBlock", QuantBasicTransformerBlock), ("ResnetBlock", QuantResnetBlock),
                         ("AttnBlock", QuantAttnBlock)):
        if name in ref:
            specials[ref[name]] = target
    if quant_act:
        specials[ldm_unet.QKMatMul] = QuantQKMatMul
        specials[ldm_unet.SMVMatMul] = QuantSMVMatMul
        # the AttentionBlock is wrapped as well (this repo's and the reference's) so that qkv -> attention -> proj runs
        # fused; the module tree and the state-dict keys stay those of the reference's recursion (quant_model.py:45-61)
        specials[ldm_unet.AttentionBlock] = QuantAttentionBlock
        if "QKMatMul" in ref:
            specials[ref["QKMatMul"]] = QuantQKMatMul
            specials[ref["SMVMatMul"]] = QuantSMVMatMul
            specials[ref["AttentionBlock"]] = QuantAttentionBlock
    else:
        specials[ldm_unet.AttentionBlock] = QuantAttentionBlock
        if "AttentionBlock" in ref:
            specials[ref["AttentionBlock"]] = QuantAttentionBlock
    return specials
