"""Checkpoint resume path of the scripts (reference qdiff/utils.py:325-457): `get_train_samples`,
`convert_adaround`, `resume_cali_model`, plus `export_cali_state_dict` — the save sequence the
reference scripts spell out inline (sample_diffusion_ddim.py:223-234, txt2img.py:477-488).

Calibration-time data capture (`save_inp_oup_data`, `GetLayerInpOut`, ... reference :18-322) belongs
to the offline reconstruction pipeline, which is out of scope for this engine.
"""
import logging

import torch
import torch.nn as nn

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
