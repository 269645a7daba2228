"""QuantModel — the drop-in boundary (reference qdiff/quant_model.py:12-96).

`QuantModel(model, weight_quant_params, act_quant_params, **kwargs)` rewrites `model` in place:
every Conv2d / Conv1d / Linear becomes a QuantModule and every known UNet block its quantised
counterpart; the wrapped model stays reachable as `.model` and the state-dict schema is the
reference's.  On top of the reference surface it offers
`capture_graph()/forward` replay of a whole UNet evaluation as one HIP graph (qdiff/graph.py).
"""
import logging
import os

import torch

from . import engine
import torch.nn as nn

from types import MethodType

from .quant_block import (BaseQuantBlock, ContextKV, EmbGroup, QuantAttentionBlock, QuantAttnBlock, QuantBasicTransformerBlock,
                          QuantQKMatMul, QuantResBlock, QuantResnetBlock, QuantSMVMatMul, get_specials, reference_classes,
                          time_mlp)
from .quant_layer import QuantModule, StraightThrough, UniformAffineQuantizer
from .arch import ddim_unet, ldm_unet

logger = logging.getLogger(__name__)

_MAX_GRAPHS = max(1, int(os.environ.get("QDIFF_HIP_GRAPH_MAX", "4")))       # captured evaluations kept per model (oldest dropped)


class QuantModel(nn.Module):

    def __init__(self, model: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {}, **kwargs):
        super().__init__()
        self.model = model
        self.sm_abit = kwargs.get('sm_abit', 8)          # `act_quant_mode` is accepted and ignored, as in the reference
        self.in_channels = model.in_channels
        if hasattr(model, 'image_size'):
            self.image_size = model.image_size
        self.quant_act = bool(act_quant_params['leaf_param'])
        self.specials = get_specials(self.quant_act)
        self.quant_module_refactor(self.model, weight_quant_params, act_quant_params)
        self.quant_block_refactor(self.model, weight_quant_params, act_quant_params)
        self._adopt_reference_modules()
        self._fuse_time_embedding()
        # HIP-graph replay of whole evaluations (qdiff/graph.py): ON by default for the integer state — the unmodified reference
        # scripts never ask for it (scripts/txt2img.py:381-390 wraps the UNet and samples) — with a graph captured the SECOND time
        # a call signature is seen, so that one-off shapes (calibration batches) are never captured; QDIFF_HIP_GRAPH=0 switches
        # the default off, enable_hip_graphs() overrides it either way (and captures on first sight).
        self._graphs = {} if os.environ.get("QDIFF_HIP_GRAPH", "1") != "0" else None
        self._graph_after, self._graph_seen, self._graph_tok = 1, {}, None
        self._own_hooks = self._hook_census()           # a replayed graph runs no Python: foreign forward hooks keep the model eager
        self._quant_state = (False, False)
        import weakref
        weakref.finalize(self, engine.release_model, id(self))      # the per-model V-sum arena dies with the model

    def quant_module_refactor(self, module: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        """Conv2d / Conv1d / Linear -> QuantModule, recursively (reference :25-43)."""
        for name, child in module.named_children():
            if isinstance(child, (nn.Conv2d, nn.Conv1d, nn.Linear)):
                setattr(module, name, QuantModule(child, weight_quant_params, act_quant_params))
            elif isinstance(child, StraightThrough):
                continue
            else:
                self.quant_module_refactor(child, weight_quant_params, act_quant_params)

    def quant_block_refactor(self, module: nn.Module, weight_quant_params: dict = {}, act_quant_params: dict = {}):
        """known blocks -> Quant*Block (reference :45-61)."""
        for name, child in module.named_children():
            target = self.specials.get(type(child))
            if target is None:
                self.quant_block_refactor(child, weight_quant_params, act_quant_params)
            elif target in (QuantBasicTransformerBlock, QuantAttnBlock):
                setattr(module, name, target(child, act_quant_params, sm_abit=self.sm_abit))
            elif target is QuantSMVMatMul:
                setattr(module, name, target(act_quant_params, sm_abit=self.sm_abit))
            elif target is QuantQKMatMul:
                setattr(module, name, target(act_quant_params))
            elif target is QuantAttentionBlock:
                setattr(module, name, target(child, act_quant_params, sm_abit=self.sm_abit, quant_matmuls=self.quant_act))
            else:
                setattr(module, name, target(child, act_quant_params))

    def _fuse_time_embedding(self):
        """K6: one launch for the `SiLU -> Linear` embedding projections of all residual blocks (they all receive the same
        embedding tensor), two for the `time_embed` MLP of the LDM / SD UNet (this repo's class or the reference's)."""
        group = EmbGroup()
        for m in self.model.modules():
            if isinstance(m, QuantResBlock) and not m.use_scale_shift_norm and isinstance(m.emb_layers[-1], QuantModule):
                group.register(m, m.emb_layers[-1])
            elif isinstance(m, QuantResnetBlock) and isinstance(m.temb_proj, QuantModule):
                group.register(m, m.temb_proj)
        ctx = ContextKV()                # cross-attention keys / values of every transformer block: one side-stream branch
        ctx.token = lambda: self.__dict__.get("_tok", -1)      # the token forward() / prepare_context() derived last (cheap: no re-derivation per block)
        self.__dict__["_ctx_kv"] = ctx
        for m in self.model.modules():
            if isinstance(m, QuantBasicTransformerBlock):
                ctx.register(m)
        # QDIFF_CTX_FORK=start: the context branch forks HERE, before the stem (`context` is the third positional argument of
        # every UNet this package wraps; QuantModel.forward passes it positionally); see quant_block.ContextKV
        from . import quant_block as _qb
        self.model.register_forward_pre_hook(lambda _m, _a: (group.reset(), ctx.reset(), engine.begin_evaluation(id(self)),
                                                             self._select_stream(),
                                                             ctx.begin(_a[2] if len(_a) > 2 else None, _qb._CTX_FORK == "start")) and None)
        # end of the evaluation: the context branch is joined, and the stream type in force goes back to "unset" (code that
        # drives kernels directly afterwards gets engine.STREAM_DTYPE, not this model's verdict)
        self.model.register_forward_hook(lambda _m, _a, _o: (ctx.finish(), engine._EFFECTIVE.__setitem__(0, None)) and None, always_call=True)
        te = getattr(self.model, "time_embed", None)
        if (isinstance(te, nn.Sequential) and len(te) == 3 and isinstance(te[0], QuantModule) and isinstance(te[2], QuantModule)
                and isinstance(te[1], nn.SiLU)):
            te.forward = MethodType(lambda seq, t_emb: time_mlp(seq[0], seq[2], t_emb), te)

    def _adopt_reference_modules(self):
        """Drop-in use on the REFERENCE's own UNet classes (scripts/txt2img.py:381-383 builds its LatentDiffusion UNet, then
        wraps it): the glue modules this engine fuses around the quantised blocks keep the reference's class, attributes
        and state-dict keys, and get this repo's forward bound onto them —
          * SpatialTransformer: GroupNorm -> proj_in int8 rows, FF-out -> proj_out int8 rows, `+ x` in the epilogue;
          * Upsample: quantise the small map, replicate int8 rows;
          * UNetModel / TimestepEmbedSequential / Downsample, in the INTEGER state only: this repo's walk
            (arch/ldm_unet.py UNetModel.forward: planned skip-concatenation buffers, engine.CatSlot; the output head on the
            integer path) — every other state runs the reference's own forward, bit for bit;
          * should the reference's forward run on the integer path after all (`QDIFF_REF_WALK=1`): the skip concatenation
            `th.cat([h, hs.pop()], dim=1)` (openaimodel.py:776) drops the GroupNorm statistics that travel with the two
            producers' outputs: hooks on the input / middle / output blocks keep a shadow stack of them and re-attach the
            concatenated statistics to the block input."""
        ref = reference_classes()
        from .first_stage_hip import adopt_reference_decoder
        adopt_reference_decoder()            # the reference's first-stage Decoder class (outside this UNet): GPU decode on the MFMA kernels
        st, up = ref.get("SpatialTransformer"), ref.get("Upsample")
        for m in self.model.modules():
            if st is not None and type(m) is st:
                m.forward = MethodType(ldm_unet.SpatialTransformer.forward, m)
            elif up is not None and type(m) is up and getattr(m, "dims", 2) == 2:
                m.forward = MethodType(ldm_unet.Upsample.forward, m)
        ddim = ref.get("DdimModel")
        if ddim is not None and type(self.model) is ddim and os.environ.get("QDIFF_REF_WALK", "0") != "1":
            # the reference's DDIM `Model` (ddim/models/diffusion.py:199-348): same treatment, integer state only
            qm, model, ref_forward = self, self.model, self.model.forward

            def ddim_forward(m, x, t=None, context=None):
                if qm._quant_state == (True, True) and not torch.is_grad_enabled() and not engine.SIMULATE:
                    return ddim_unet.Model.forward(m, x, t, context)
                return ref_forward(x, t, context)
            model.forward = MethodType(ddim_forward, model)
            for m in model.modules():
                if type(m) is ref.get("DdimUpsample"):
                    m.forward = MethodType(ddim_unet.Upsample.forward, m)
                    m.qd_takes_out_slot = True
                elif type(m) is ref.get("DdimDownsample"):
                    m.forward = MethodType(ddim_unet.Downsample.forward, m)
                    m.qd_takes_out_slot = True
            return
        unet = ref.get("UNetModel")
        if unet is None or type(self.model) is not unet:
            return
        if os.environ.get("QDIFF_REF_WALK", "0") != "1":
            self._adopt_reference_walk(ref)
        state = {"stack": [], "last": None}

        def part_of(t):
            return getattr(t, "qd_gn_part", None) if torch.is_tensor(t) else None

        def reset(_m, _args):
            state["stack"], state["last"] = [], None

        def push(_m, _args, out):
            state["stack"].append(part_of(out))

        def keep_last(_m, _args, out):
            state["last"] = part_of(out)

        def reattach(_m, args):
            pb = state["stack"].pop() if state["stack"] else None
            pa, h = state["last"], args[0]
            if torch.is_tensor(h) and getattr(h, "qd_gn_part", None) is not None:
                return                                     # this repo's walk already concatenated the statistics (as a view)
            if pa is not None and pb is not None and torch.is_tensor(h) and pa.shape[0] * pa.shape[1] == pb.shape[0] * pb.shape[1] \
                    and pa.shape[2] + pb.shape[2] == h.shape[1]:
                n = pa.shape[0] * pa.shape[1]
                h.qd_gn_part = torch.cat([pa.reshape(1, n, -1, 2), pb.reshape(1, n, -1, 2)], dim=2)

        self.model.input_blocks[0].register_forward_pre_hook(reset)
        for blk in self.model.input_blocks:
            blk.register_forward_hook(push)
        self.model.middle_block.register_forward_hook(keep_last)
        for blk in self.model.output_blocks:
            blk.register_forward_pre_hook(reattach)
            blk.register_forward_hook(keep_last)

    def _adopt_reference_walk(self, ref):
        """Bind this repo's UNet walk onto the reference's UNetModel for the integer state (see _adopt_reference_modules)."""
        qm, model = self, self.model
        ref_forward = model.forward                        # bound method of the reference class
        ref_st, ref_seq, ref_down = ref.get("SpatialTransformer"), ref.get("TimestepEmbedSequential"), ref.get("Downsample")

        def unet_forward(m, x, timesteps=None, context=None, y=None, **kw):
            if (qm._quant_state == (True, True) and not torch.is_grad_enabled() and not engine.SIMULATE
                    and not getattr(m, "predict_codebook_ids", False)):
                return ldm_unet.UNetModel.forward(m, x, timesteps, context, y, **kw)
            return ref_forward(x, timesteps, context, y, **kw)

        def seq_forward(seq, x, emb, context=None, split=0, out_slot=None):
            last = len(seq) - 1
            for i, layer in enumerate(seq):
                kw = {"out_slot": out_slot} if (i == last and out_slot is not None and getattr(layer, "qd_takes_out_slot", False)) else {}
                if isinstance(layer, ref["TimestepBlock"]):
                    x = layer(x, emb, split=split, **kw)
                elif ref_st is not None and isinstance(layer, ref_st):
                    x = layer(x, context, **kw)
                else:
                    x = layer(x, **kw)
            return x

        model.forward = MethodType(unet_forward, model)
        model._out = MethodType(ldm_unet.UNetModel._out, model)
        for m in model.modules():
            if ref_seq is not None and type(m) is ref_seq:
                m.forward = MethodType(seq_forward, m)
            elif ref_st is not None and type(m) is ref_st:
                m.qd_takes_out_slot = True                 # its forward is this repo's SpatialTransformer.forward
            elif ref_down is not None and type(m) is ref_down and getattr(m, "dims", 2) == 2:
                m.forward = MethodType(ldm_unet.Downsample.forward, m)
                m.qd_takes_out_slot = True
            elif ref.get("Upsample") is not None and type(m) is ref["Upsample"] and getattr(m, "dims", 2) == 2:
                m.qd_takes_out_slot = True

    def _hook_census(self, refresh=True):
        """Forward (pre-)hooks registered below the wrapped model.  Calibration and the tests' recorders hook sub-modules and
        expect every evaluation to run them; a HIP-graph replay runs no Python at all, so an evaluation is captured — AND
        replayed — only while the census is the one this wrapper left behind (its own hooks).  refresh=False counts over the
        hook tables collected last time (the per-replay check: ~3k `len` calls, no module walk; a module swapped in since
        moves the state token, which drops the graphs and refreshes the tables)."""
        d = self.__dict__
        if refresh or d.get("_hook_tables") is None:
            d["_hook_tables"] = [t for m in self.model.modules() for t in (m._forward_hooks, m._forward_pre_hooks)]
        return sum(map(len, d["_hook_tables"]))

    # ---- state token: "the whole model is on the integer path, and these are the tensors its plans were made from" ------------
    # HIP graphs and prepared contexts bake device pointers and quantiser values in; they are valid for as long as the token
    # does not move.  Per evaluation the check costs one integer compare (engine.STATE_GENERATION: bumped by every (re)assignment
    # of a quantiser attribute, weight, switch — the __setattr__ hooks of quant_layer / adaptive_rounding — and by
    # invalidate()) plus the sum of ~1.5k tensor version counters (in-place edits: optimizer steps, load_state_dict); only when
    # one of the two has moved are the per-layer plan keys re-derived and compared.  Writes through `.data` bypass both, as they
    # bypass the plan caches: invalidate_plans() after such an edit.
    def _fingerprint(self):
        if self._quant_state != (True, True) or engine.SIMULATE:
            return None, None
        keys, tens = [], []
        names = ("act_quantizer_q", "act_quantizer_k", "act_quantizer_v", "act_quantizer_w")
        for m in self.model.modules():
            if isinstance(m, QuantModule):
                if not m.int_ready() or any((not q.inited) or q.running_stat for q in m._act_quantizers()):
                    return None, None
                keys.append(m.plan_keys()[1])
                tens += m.state_tensors()
                continue
            qs = [q for q in (m.__dict__.get("_modules", {}).get(n) for n in names) if isinstance(q, UniformAffineQuantizer)]
            if qs:
                if any((not q.inited) or q.running_stat for q in qs):
                    return None, None
                keys.append(tuple(engine.quantizer_key(q) for q in qs))
                tens += [t for q in qs for t in (q.delta, q.zero_point) if torch.is_tensor(t)]
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                keys.append(tuple((id(t), t.data_ptr()) for t in (m.weight, m.bias) if torch.is_tensor(t)))
        return tuple(keys), tens

    @staticmethod
    def _vsum(tens):
        try:
            return sum(t._version for t in tens)
        except RuntimeError:                               # inference tensors carry no counter
            return sum(engine.tensor_version(t) or 0 for t in tens)

    def _state_token(self):
        """>= 0: every layer takes the integer path and nothing its plans were made from has changed since this number was
        handed out; -1: not (yet) the case.  Never re-derives under stream capture (the warm-up evaluations did)."""
        d = self.__dict__
        gen = engine.STATE_GENERATION[0]
        tens = d.get("_tok_tensors")
        if d.get("_tok_gen") == gen and (tens is None or self._vsum(tens) == d["_tok_vsum"]):
            return d["_tok"]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return d.get("_tok", -1)
        keys, tens = self._fingerprint()
        if keys is None:
            d["_tok"], d["_tok_keys"] = -1, None
        elif d.get("_tok_keys") != keys:
            d["_tok_serial"] = d.get("_tok_serial", -1) + 1
            d["_tok"], d["_tok_keys"] = d["_tok_serial"], keys
        else:
            d["_tok"] = d["_tok_serial"]
        d["_tok_tensors"], d["_tok_vsum"] = tens, (self._vsum(tens) if tens is not None else 0)
        d["_tok_gen"] = engine.STATE_GENERATION[0]         # plan_keys() may have initialised weight quantisers (bumps)
        return d["_tok"]

    def state_token(self):
        return self._state_token()

    def _select_stream(self):
        """fp16 activation stream (engine.STREAM_DTYPE) only for evaluations that run entirely on the integer path."""
        if engine.STREAM_DTYPE == torch.float32:
            engine._EFFECTIVE[0] = None
            return
        ok = (not torch.is_grad_enabled()) and not self.model.training and self._state_token() >= 0
        engine._EFFECTIVE[0] = engine.STREAM_DTYPE if ok else torch.float32

    def _prepare(self, context):
        """ContextKV.pin under this model's stream verdict (the chain's projections must emit the rows an evaluation would:
        engine._EFFECTIVE is unset outside evaluations).  Returns the entry or None."""
        ctx = self.__dict__.get("_ctx_kv")
        if (ctx is None or not torch.is_tensor(context) or torch.is_grad_enabled() or self.model.training
                or (context.is_cuda and torch.cuda.is_current_stream_capturing())      # captured launches would not have run yet
                or self._state_token() < 0):
            return None
        self._select_stream()
        try:
            return ctx.pin(context)
        finally:
            engine._EFFECTIVE[0] = None

    def prepare_context(self, context):
        """Compute the cross-attention K / V^T operands of every transformer block for `context` ONCE; evaluations that are
        handed this tensor — or ANY tensor with the same bytes: the reference's samplers rebuild `torch.cat([uncond, c])` at
        every step (plms.py:184-187), see ContextKV.match — skip the ~150-launch chain of to_k / to_v / head-layout quantisers
        that the reference repeats in each of the 51 evaluations of a sampling run (quant_block.py:193-195).  Static input,
        static weights, static quantisers: the bytes are those the per-evaluation branch produces, bit for bit
        (tests/test_engine_models.py::test_prepared_context_changes_nothing).  Only in the integer state with every
        quantiser initialised and no running statistics; returns False (and changes nothing) otherwise or when QDIFF_CTX_PIN=0.
        Calling it is OPTIONAL since round 5: forward() prepares a context it has not seen by itself (QDIFF_CTX_AUTO=0 to
        switch that off) and keeps QDIFF_CTX_PINS (2) of them; a changed quant state or re-packed weights drop them."""
        return self._prepare(context) is not None

    def release_context(self):
        ctx = self.__dict__.get("_ctx_kv")
        if ctx is not None:
            ctx.unpin()

    def lock_context(self, context):
        """For callers that capture evaluations of this model into graphs of their OWN (sampling.DevicePLMS): the prepared
        entry of `context` (prepared now if need be) is pinned down — not evicted, not rewritten with other bytes — until
        unlock_context(handle).  Under stream capture only a locked entry is used.  Returns the handle or None."""
        ctx = self.__dict__.get("_ctx_kv")
        if ctx is None or not torch.is_tensor(context):
            return None
        e = ctx.match(context) if self._state_token() >= 0 else None
        if e is None:
            e = self._prepare(context)
        if e is not None:
            e["locked"] += 1
        return e

    def unlock_context(self, handle):
        if handle is not None and handle["locked"] > 0:
            handle["locked"] -= 1

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        engine.bump_state()
        self.release_context()
        self._quant_state = (bool(weight_quant), bool(act_quant))
        for m in self.model.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.set_quant_state(weight_quant, act_quant)

    def forward(self, x, timesteps=None, context=None):
        from . import quant_block as qb
        ckv = self.__dict__.get("_ctx_kv")
        cuda = torch.is_tensor(x) and x.is_cuda
        capturing = cuda and torch.cuda.is_current_stream_capturing()
        tok = -1
        if not torch.is_grad_enabled() and not self.model.training and not capturing:
            tok = self._state_token()
        entry = None
        graphs = self._graphs is not None and tok >= 0 and cuda and torch.is_tensor(timesteps)
        if graphs:
            from .graph import GraphedUNet, signature
            if self._graph_tok != tok:                     # plans were rebuilt: the captured pointers are stale
                self._graphs.clear()
                self._graph_seen.clear()
                self._graph_tok = tok
                self.__dict__["_hook_tables"] = None
            # a replay runs no Python: with a foreign hook below the model (calibration capture, recorders — registered at any
            # time, also AFTER a graph of this signature was captured) the evaluation stays eager
            graphs = self._hook_census(refresh=False) == self._own_hooks
        if tok >= 0 and ckv is not None and torch.is_tensor(context) and qb._CTX_PIN:
            # the run's conditioning: prepared before (same tensor, or the same BYTES in a fresh tensor), or prepared now
            entry = ckv.match(context, by_value=False)
            if entry is None:
                entry = ckv.match(context)
            if entry is None and qb._CTX_AUTO:
                entry = self._prepare(context)
        if ckv is not None:
            ckv.select(entry, context)
        try:
            if graphs:
                key = signature(x, timesteps, context) + (engine.STREAM_DTYPE, None if entry is None else entry["slot"])
                g = self._graphs.get(key)
                if g is None:
                    if len(self._graph_seen) > 256:
                        self._graph_seen.clear()
                    seen = self._graph_seen.get(key, 0)
                    self._graph_seen[key] = seen + 1
                    if seen >= self._graph_after and self._hook_census() == self._own_hooks:
                        while len(self._graphs) >= _MAX_GRAPHS:
                            old = next(iter(self._graphs))
                            logger.info("HIP graph of signature %s dropped (QDIFF_HIP_GRAPH_MAX=%d)", old[:5], _MAX_GRAPHS)
                            self._graphs.pop(old)
                        # ONE private memory pool for all captures of this model: replays are serialised on the launch stream,
                        # so the activations of different signatures may share the same bytes
                        pool = self.__dict__.get("_graph_pool")
                        if pool is None:
                            pool = self.__dict__["_graph_pool"] = torch.cuda.graph_pool_handle()
                        g = self._graphs[key] = GraphedUNet(self, x, timesteps, context, pinned=entry is not None, pool=pool)
                        logger.info("HIP graph captured for signature %s (%d kept)", key[:5], len(self._graphs))
                if g is not None:
                    return g(x, timesteps, context).to(x.dtype, copy=True)
            y = self.model(x, timesteps, context)
        finally:
            if ckv is not None:
                ckv.select(None, None)
        # an fp16 activation stream ends here: the samplers' update arithmetic runs in the latent's own type
        return y.to(x.dtype) if torch.is_tensor(y) and y.dtype == torch.float16 and x.dtype != torch.float16 else y

    def invalidate_plans(self):
        """Forget every packed weight / epilogue constant / captured graph (see QuantModule.invalidate)."""
        engine.bump_state()
        self.__dict__["_tok_keys"] = None                  # the next token is a new one even if every key looks the same
        self.release_context()
        for m in self.model.modules():
            if isinstance(m, QuantModule):
                m.invalidate()
            m.__dict__.pop("_attn_plan_cache", None)
        if self._graphs is not None:
            self._graphs = {}
            self._graph_seen = {}

    def enable_hip_graphs(self, on: bool = True):
        """Replay each (shape, quant-state) UNet evaluation as one HIP graph (qdiff/graph.py), captured the FIRST time a call
        signature is seen (the default — replay on, capture on second sight — needs no call).  Quantiser parameters are baked
        into device tensors at capture; the state token (see _state_token) drops the graphs when they change."""
        self._graphs = {} if on else None
        self._graph_seen = {}
        self._graph_after = 0

    def set_running_stat(self, running_stat: bool, sm_only=False):
        """reference :71-87"""
        engine.bump_state()
        self.release_context()
        for m in self.model.modules():
            if isinstance(m, QuantBasicTransformerBlock):
                names = ("act_quantizer_w",) if sm_only else ("act_quantizer_q", "act_quantizer_k", "act_quantizer_v",
                                                                 "act_quantizer_w")
                for att in (m.attn1, m.attn2):
                    for n in names:
                        getattr(att, n).running_stat = running_stat
            if isinstance(m, QuantModule) and not sm_only:
                m.set_running_stat(running_stat)

    def set_grad_ckpt(self, grad_ckpt: bool):
        """reference :89-96 (transformer blocks only)."""
        btb = tuple(t for t, q in self.specials.items() if q is QuantBasicTransformerBlock)  # unwrapped blocks, if any
        for _, m in self.model.named_modules():
            if isinstance(m, (QuantBasicTransformerBlock,) + btb):
                m.checkpoint = grad_ckpt
