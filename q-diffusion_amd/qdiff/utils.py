"""Checkpoint resume path of the scripts (reference qdiff/utils.py:325-457): `get_train_samples`,
`convert_adaround`, `resume_cali_model`, plus `export_cali_state_dict` — the save sequence the
reference scripts spell out inline (sample_diffusion_ddim.py:223-234, txt2img.py:477-488).

Calibration-time data capture (`save_inp_oup_data`, `save_grad_data` over the `tap_unit` context manager: reference
:18-322) feeds qdiff/recon.py.
"""
import contextlib
import logging
import types
from typing import Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .adaptive_rounding import AdaRoundQuantizer
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule, UniformAffineQuantizer

logger = logging.getLogger(__name__)


def get_train_samples(args, sample_data, custom_steps=None):
    """Pick calibration samples spread over the sampler's timesteps (reference utils.py:325-348)."""
    num_samples, num_st = args.cali_n, args.cali_st
    custom_steps = args.custom_steps if custom_steps is None else custom_steps
    if num_st == 1:
        xs = sample_data[:num_samples]
        ts = torch.ones(num_samples) * 800
        return xs, ts
    nsteps = len(sample_data["ts"])
    assert nsteps >= custom_steps
    picks = list(range(0, nsteps, nsteps // num_st))
    logger.info(f'Selected {len(picks)} steps from {nsteps} sampling steps')
    xs = [sample_data["xs"][i][:num_samples] for i in picks]
    ts = [sample_data["ts"][i][:num_samples] for i in picks]
    if getattr(args, "cond", False):
        xs, ts = xs + xs, ts + ts
        conds = [sample_data["cs"][i][:num_samples] for i in picks] + [sample_data["ucs"][i][:num_samples] for i in picks]
        return torch.cat(xs, dim=0), torch.cat(ts, dim=0), torch.cat(conds, dim=0)
    return torch.cat(xs, dim=0), torch.cat(ts, dim=0)


def _to_adaround(mod: QuantModule):
    if mod.split != 0:
        s = mod.split
        mod.weight_quantizer = AdaRoundQuantizer(uaq=mod.weight_quantizer, round_mode='learned_hard_sigmoid',
                                                 weight_tensor=mod.org_weight.data[:, :s, ...])
        mod.weight_quantizer_0 = AdaRoundQuantizer(uaq=mod.weight_quantizer_0, round_mode='learned_hard_sigmoid',
                                                   weight_tensor=mod.org_weight.data[:, s:, ...])
    else:
        mod.weight_quantizer = AdaRoundQuantizer(uaq=mod.weight_quantizer, round_mode='learned_hard_sigmoid',
                                                 weight_tensor=mod.org_weight.data)


def convert_adaround(model):
    """Swap every (initialised) uniform weight quantiser for an AdaRound one (reference :351-379).
    Note the reference only honours `split` for modules inside blocks; a top-level QuantModule is
    converted whole — kept as is."""
    for _, module in model.named_children():
        if isinstance(module, QuantModule):
            if not module.ignore_reconstruction:
                module.weight_quantizer = AdaRoundQuantizer(uaq=module.weight_quantizer, round_mode='learned_hard_sigmoid',
                                                            weight_tensor=module.org_weight.data)
        elif isinstance(module, BaseQuantBlock):
            if not module.ignore_reconstruction:
                for _, sub in module.named_modules():
                    if isinstance(sub, QuantModule):
                        _to_adaround(sub)
        else:
            convert_adaround(module)


def _wrap_adaround_params(qnn):
    for m in qnn.model.modules():
        if isinstance(m, AdaRoundQuantizer):
            m.zero_point = nn.Parameter(m.zero_point)
            m.delta = nn.Parameter(m.delta)


def _unwrap_adaround_params(qnn):
    for m in qnn.model.modules():
        if isinstance(m, AdaRoundQuantizer):
            for name in ("zero_point", "delta"):
                data = getattr(m, name).data
                delattr(m, name)
                setattr(m, name, data)


def _model_device(qnn):
    return next(qnn.parameters()).device


def _cali_forward(qnn, cali_data, cond):
    dev = _model_device(qnn)
    args = [t[:1].to(dev) for t in cali_data[:3 if cond else 2]]
    with torch.no_grad():
        qnn(*args)


def resume_cali_model(qnn, ckpt_path, cali_data, quant_act=False, act_quant_mode='qdiff', cond=False):
    """Rebuild the quantiser objects a calibrated checkpoint expects, then load it
    (reference utils.py:382-457; same two-stage sequence, same resulting attribute types:
    AdaRound delta/zero_point plain tensors, act delta nn.Parameter, act zero_point Python int)."""
    print("Loading quantized model checkpoint")
    ckpt = torch.load(ckpt_path, map_location='cpu')

    print("Initializing weight quantization parameters")
    qnn.set_quant_state(True, False)
    _cali_forward(qnn, cali_data, cond)           # initialises weight quantisers, triggers set_split
    convert_adaround(qnn)
    _wrap_adaround_params(qnn)
    weights_only = {k: v for k, v in ckpt.items() if "act" not in k}   # act-quantiser state comes later
    qnn.load_state_dict(weights_only, strict=(act_quant_mode == 'qdiff'))
    qnn.set_quant_state(weight_quant=True, act_quant=False)
    _unwrap_adaround_params(qnn)

    if quant_act:
        print("Initializing act quantization parameters")
        qnn.set_quant_state(True, True)
        _cali_forward(qnn, cali_data, cond)       # initialises every activation quantiser from data
        print("Loading quantized model checkpoint again")
        _wrap_adaround_params(qnn)
        for m in qnn.model.modules():
            if isinstance(m, UniformAffineQuantizer) and m.zero_point is not None:
                zp = m.zero_point if torch.is_tensor(m.zero_point) else torch.tensor(float(m.zero_point))
                m.zero_point = nn.Parameter(zp)
        qnn.load_state_dict(torch.load(ckpt_path, map_location='cpu'))
        qnn.set_quant_state(weight_quant=True, act_quant=True)
        _unwrap_adaround_params(qnn)
        for m in qnn.model.modules():
            if isinstance(m, UniformAffineQuantizer) and m.zero_point is not None:
                zp = m.zero_point.item()
                delattr(m, "zero_point")
                assert int(zp) == zp
                m.zero_point = int(zp)


def export_cali_state_dict(qnn):
    """state_dict() in the reference checkpoint format: AdaRound delta/zero_point and activation
    zero_points are temporarily wrapped as Parameters so that they land in the dict
    (sample_diffusion_ddim.py:223-234), then restored."""
    wrapped = []
    for m in qnn.model.modules():
        if isinstance(m, AdaRoundQuantizer):
            for name in ("zero_point", "delta"):
                v = getattr(m, name)
                if not isinstance(v, nn.Parameter):
                    wrapped.append((m, name, v))
                    setattr(m, name, nn.Parameter(v))     # Module.__setattr__ drops the plain attribute
        elif isinstance(m, UniformAffineQuantizer) and m.zero_point is not None:
            v = m.zero_point
            if not isinstance(v, nn.Parameter):
                wrapped.append((m, "zero_point", v))
                t = v if torch.is_tensor(v) else torch.tensor(float(v))
                m.zero_point = nn.Parameter(t.float())
    sd = {k: v.detach().clone() for k, v in qnn.state_dict().items()}
    for m, name, v in wrapped:
        delattr(m, name)
        setattr(m, name, v)
    return sd


# ------------------------------------------------------------------------------------------------
# packed checkpoint: what the kernels read, not what calibration wrote
# ------------------------------------------------------------------------------------------------
PACKED_FORMAT = "qdiff-packed-v1"


def export_packed_ckpt(qnn, to_cpu=True):
    """Serializable inference state of a calibrated QuantModel in (True, True) state.  The reference's checkpoint keeps,
    per quantised layer, the fp32 weight, the fp32 AdaRound alpha of the same shape, delta and zero_point (SD-v1.4: ~7 GB);
    this one keeps the packed int4/int8 codes the kernels contract (tile-ordered), the per-channel constants and the
    activation quantisers (SD-v1.4 W4: ~0.45 GB).  Float parameters that are not quantised weights (norms, biases) are
    stored as they are.  See load_packed_ckpt.  to_cpu=False keeps every tensor where it lives (no copies): the form
    sampling.broadcast_packed_model ships over RCCL."""
    from . import engine
    keep = (lambda t: t.detach().cpu().clone()) if to_cpu else (lambda t: t.detach())
    mods, quantizers, skip = {}, {}, set()
    # the GEGLU projections (`ff.net[0].proj` of every transformer block, reference attention.py:37-44): they run through a
    # SECOND, value/gate-interleaved pack (QuantModule.geglu_plan) which must travel whether or not an integer forward has
    # built it on the exporting side yet — a receiver without it silently takes the unfused projection + qd_geglu_quant path
    geglu_ids = set()
    for blk in qnn.model.modules():
        proj = getattr(getattr(getattr(getattr(blk, "ff", None), "net", [None])[0], "proj", None), "geglu_plan", None) \
            if hasattr(blk, "attn1") and hasattr(blk, "norm3") else None
        if proj is not None:
            geglu_ids.add(id(blk.ff.net[0].proj))
    for name, m in qnn.model.named_modules():
        if isinstance(m, QuantModule):
            if not m.int_ready():
                raise ValueError(f"{name}: export_packed_ckpt needs set_quant_state(True, True) and initialised quantisers")
            entry = dict(pack=engine.pack_to_dict(m.conv_plan().pack, to_cpu), split=int(m.split))
            if id(m) in geglu_ids or "_geglu_cache" in m.__dict__ or m.__dict__.get("_frozen_geglu_pack") is not None:
                gp = m.geglu_plan()              # None when the layer does not qualify; always for the CURRENT quantisers
                if gp is not None:
                    entry["geglu_pack"] = engine.pack_to_dict(gp.pack, to_cpu)
            mods[name] = entry
            skip.add(f"{name}.weight")
        if isinstance(m, UniformAffineQuantizer) and not isinstance(m, AdaRoundQuantizer) and "weight_quantizer" not in name \
                and m.inited and m.delta is not None:
            z = m.zero_point
            quantizers[name] = dict(delta=keep(m.delta) if torch.is_tensor(m.delta) else float(m.delta),
                                    zero_point=keep(z) if torch.is_tensor(z) else float(z))
    tensors = {}
    for k, v in qnn.model.state_dict().items():
        if k in skip or "weight_quantizer" in k or "act_quantizer" in k:
            continue
        tensors[k] = keep(v)
    return dict(format=PACKED_FORMAT, modules=mods, quantizers=quantizers, tensors=tensors)


def save_packed_ckpt(qnn, path):
    torch.save(export_packed_ckpt(qnn), path)


def load_packed_ckpt(qnn, ckpt, free_weights=True):
    """Load a packed checkpoint (dict or path) into a QuantModel built on the same architecture (its weights are
    irrelevant).  Leaves the model in (True, True) state; with free_weights the fp32 weights of the quantised layers are
    released (the integer path does not read them; the floating-point states are then unavailable)."""
    from . import engine
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    if ckpt.get("format") != PACKED_FORMAT:
        raise ValueError("not a qdiff packed checkpoint")
    dev = _model_device(qnn)
    named = dict(qnn.model.named_modules())
    missing = [n for n in ckpt["modules"] if not isinstance(named.get(n), QuantModule)]
    if missing:
        raise KeyError(f"packed checkpoint has layers this model lacks: {missing[:4]}...")
    for name, entry in ckpt["modules"].items():
        m = named[name]
        if entry["split"]:
            m._note_split(entry["split"])
    named = dict(qnn.model.named_modules())                      # split created act_quantizer_0 / weight_quantizer_0
    for name, st in ckpt["quantizers"].items():
        qz = named[name]
        d = st["delta"]
        d = d.to(dev) if torch.is_tensor(d) else torch.tensor(float(d), device=dev)
        if isinstance(getattr(qz, "delta", None), nn.Parameter) or "delta" in qz._parameters:
            del qz.delta
        qz.delta = nn.Parameter(d) if qz.leaf_param else d
        z = st["zero_point"]
        if "zero_point" in qz._parameters:
            del qz.zero_point
        qz.zero_point = z.to(dev) if torch.is_tensor(z) else z
        qz.inited = True
    own = qnn.model.state_dict()
    qnn.model.load_state_dict({k: v for k, v in ckpt["tensors"].items() if k in own}, strict=False)
    for name, entry in ckpt["modules"].items():
        m = named[name]
        gp = engine.pack_from_dict(entry["geglu_pack"], dev) if "geglu_pack" in entry else None
        m.load_packed(engine.pack_from_dict(entry["pack"], dev), gp)
        if free_weights:
            m.weight.data = torch.empty(0, device=dev)
            m.org_weight = torch.empty(0, device=dev)
            for wq in m._weight_quantizers():
                if isinstance(wq, AdaRoundQuantizer):
                    wq.alpha.data = torch.empty(0, device=dev)
    qnn.set_quant_state(True, True)
    return qnn


# ------------------------------------------------------------------------------------------------
# calibration-time capture of what flows through one reconstruction unit (reference utils.py:18-322: `save_inp_oup_data`
# and `save_grad_data` are the names block_recon / layer_recon call; the taps below are this package's own)
# ------------------------------------------------------------------------------------------------
class _UnitReached(Exception):
    """Unwinds a UNet evaluation once the tapped unit has run: nothing downstream of it is needed."""


@contextlib.contextmanager
def tap_unit(unit, *, backward=False):
    """Record what passes through `unit` while the body runs.  Forward tap: `rec.args` / `rec.out` are taken on every call
    of the unit (`rec.keep_out = False` freezes the output for a later pass) and the evaluation is abandoned right after;
    backward tap: `rec.grad` is the gradient w.r.t. the unit's (first) output, the backward pass runs to its end."""
    rec = types.SimpleNamespace(args=None, out=None, grad=None, keep_out=True)
    if backward:
        handle = unit.register_full_backward_hook(lambda _m, _gin, gout: setattr(rec, "grad", gout[0]))
    else:
        def on_forward(_m, args, out):
            rec.args = args
            if rec.keep_out:
                rec.out = out
            raise _UnitReached
        handle = unit.register_forward_hook(on_forward)
    try:
        yield rec
    finally:
        handle.remove()


def _evaluate_up_to_tap(model, args):
    try:
        model(*args)
    except _UnitReached:
        pass


def _hand_back(model, unit, act_quant):
    """State every capture leaves behind (reference :248-250, :306-308): only `unit` quantised, model in train mode."""
    model.set_quant_state(False, False)
    unit.set_quant_state(True, act_quant)
    model.train()


def unit_inputs_and_target(model, unit, args, asym=False, act_quant=False):
    """One calibration batch through the network up to `unit` (reference GetLayerInpOut, :213-255).  The target is the
    unit's output in the FULL-PRECISION network; the inputs come from the same pass, or — `asym`, BRECQ's asymmetric
    reconstruction — from a second pass with everything quantised (weights, activations too when `act_quant`).
    Returns (x | (x, second), target), detached; `second` is the embedding / context of two-tensor units."""
    model.eval()
    with torch.no_grad(), tap_unit(unit) as rec:
        model.set_quant_state(False, False)
        _evaluate_up_to_tap(model, args)
        if asym:
            rec.keep_out = False
            model.set_quant_state(weight_quant=True, act_quant=act_quant)
            _evaluate_up_to_tap(model, args)
    _hand_back(model, unit, act_quant)
    ins = [a.detach() for a in rec.args[:2] if torch.is_tensor(a)]
    return (tuple(ins) if len(ins) == 2 else ins[0]), rec.out.detach()


def _is_attention_map(t):
    return torch.is_tensor(t) and t.dim() > 2 and t.shape[1] == t.shape[2] == 4096


def save_inp_oup_data(model, layer: Union[QuantModule, BaseQuantBlock], cali_data, asym: bool = False, act_quant: bool = False,
                      batch_size: int = 32, keep_gpu: bool = True, cond: bool = False, is_sm: bool = False):
    """Inputs and full-precision outputs of `layer` over the calibration set (reference :18-149).  Units that take two
    tensors (x, emb) / (x, context) return `[xs, seconds]`.  `is_sm`: when the unit's input or output is a 4096 x 4096
    attention map only a random half of the calibration samples is kept (the reference's memory guard, :38-70 — kept
    because it decides WHICH samples the unit is calibrated on)."""
    device = next(model.parameters()).device
    columns = tuple(cali_data[:3]) if cond else tuple(cali_data[:2])     # xs, ts[, conds]
    n = columns[0].size(0)

    def batch(sel):
        return unit_inputs_and_target(model, layer, [c[sel].to(device) for c in columns], asym, act_quant)

    order = None
    if is_sm:
        probe_in, probe_out = batch(slice(0, 1))
        if _is_attention_map(probe_in[0] if isinstance(probe_in, tuple) else None) or _is_attention_map(probe_out):
            logger.info("attention-map unit: calibrating on a random half of the samples")
            order = torch.as_tensor(np.random.choice(n, n // 2, replace=False))
    nbatch = int(n / batch_size) // (2 if order is not None else 1)
    keep = (lambda t: t) if keep_gpu else (lambda t: t.cpu())
    firsts, seconds, targets = [], [], []
    for b in range(nbatch):
        sel = slice(b * batch_size, (b + 1) * batch_size)
        got, target = batch(order[sel] if order is not None else sel)
        if isinstance(got, tuple):
            seconds.append(keep(got[1]))
            got = got[0]
        firsts.append(keep(got))
        targets.append(keep(target))
    if device.type == 'cuda':
        torch.cuda.empty_cache()
    xs = torch.cat(firsts)
    return ([xs, torch.cat(seconds)] if seconds else xs), torch.cat(targets)


def quantize_up_to(model, last, act_quant=False):
    """Only the units that precede `last` in module order — and `last` itself — are quantised (reference :313-322)."""
    model.set_quant_state(False, False)
    for m in model.modules():
        if isinstance(m, (QuantModule, BaseQuantBlock)):
            m.set_quant_state(True, act_quant)
        if m is last:
            return


def unit_output_gradient(model, unit, args, act_quant=False):
    """d KL(softmax(full precision) || softmax(quantised up to `unit`)) / d(output of `unit`): the Fisher weights of
    opt_mode 'fisher_diag' / 'fisher_full' (reference GetLayerGrad, :271-310)."""
    model.eval()
    with torch.enable_grad(), tap_unit(unit, backward=True) as rec:
        model.zero_grad()
        model.set_quant_state(False, False)
        target = F.softmax(model(*args), dim=1)
        quantize_up_to(model, unit, act_quant)
        F.kl_div(F.log_softmax(model(*args), dim=1), target, reduction='batchmean').backward()
    _hand_back(model, unit, act_quant)
    return rec.grad.detach()


def save_grad_data(model, layer, cali_data, damping: float = 1., act_quant: bool = False, batch_size: int = 32, keep_gpu: bool = True):
    """|dL/d(output)| + 1 of `layer` over the calibration set (reference :152-180).  `cali_data` may be the (xs, ts[, conds])
    tuple of the diffusion scripts (the reference indexes a single tensor here, a leftover of its classification origin)."""
    device = next(model.parameters()).device
    columns = tuple(cali_data) if isinstance(cali_data, (tuple, list)) else (cali_data,)
    grads = []
    for b in range(int(columns[0].size(0) / batch_size)):
        sel = slice(b * batch_size, (b + 1) * batch_size)
        g = unit_output_gradient(model, layer, [c[sel].to(device) for c in columns], act_quant)
        grads.append(g if keep_gpu else g.cpu())
    if device.type == 'cuda':
        torch.cuda.empty_cache()
    return torch.cat(grads).abs() + 1.0
