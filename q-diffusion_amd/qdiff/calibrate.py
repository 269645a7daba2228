"""Calibration driver — the flow the reference's scripts spell out inline (scripts/sample_diffusion_ddim.py:150-234,
scripts/sample_diffusion_ldm.py, scripts/txt2img.py:393-488), as two library functions:

    recon_model(qnn, **kwargs)      walk the wrapped UNet: QuantModule children -> layer_reconstruction, BaseQuantBlock
                                    children -> block_reconstruction, anything else -> recurse (scripts' `recon_model`)
    calibrate_model(qnn, cali_data, ...)   initialise weight quantisers -> weight (AdaRound) phase -> initialise activation
                                    quantisers (+ optional EMA range tracking) -> activation step-size phase -> the
                                    reference-format state dict (utils.export_cali_state_dict)

On MI355X the whole calibration set, the unit caches and the model live in HBM (288 GB); nothing is paged from the
host between iterations (qdiff/recon.py).  SURVEY.md §8(f) N2.
"""
import logging

import numpy as np
import torch

from .block_recon import block_reconstruction
from .layer_recon import layer_reconstruction
from .quant_block import BaseQuantBlock
from .quant_layer import QuantModule

logger = logging.getLogger(__name__)


def recon_model(qnn, module=None, on_unit=None, **kwargs):
    """Block reconstruction over `module` (default: the whole QuantModel); the first and the last convolution, which are
    bare QuantModules, get layer reconstruction.  A QuantAttentionBlock built for quantised activations is walked INTO:
    the reference leaves the LDM AttentionBlock unwrapped in that mode (quant_block.py:389-401), so its qkv / proj_out
    layers and its QuantQKMatMul / QuantSMVMatMul blocks are reconstruction units of their own.  `on_unit(name, unit)` is called after each unit (checkpointing hook: the
    reference saves a temporary checkpoint before the output blocks, txt2img.py:422-428)."""
    module = qnn if module is None else module
    for name, child in module.named_children():
        if isinstance(child, QuantModule):
            if child.ignore_reconstruction is True:
                logger.info('Ignore reconstruction of layer {}'.format(name))
                continue
            logger.info('Reconstruction for layer {}'.format(name))
            layer_reconstruction(qnn, child, **kwargs)
        elif isinstance(child, BaseQuantBlock) and not getattr(child, "quant_matmuls", False):
            if child.ignore_reconstruction is True:
                logger.info('Ignore reconstruction of block {}'.format(name))
                continue
            logger.info('Reconstruction for block {}'.format(name))
            block_reconstruction(qnn, child, **kwargs)
        else:
            recon_model(qnn, child, on_unit=on_unit, **kwargs)
            continue
        if on_unit is not None:
            on_unit(name, child)


def sync_quantisers(qnn, src=0):
    """Data-parallel calibration: every rank initialised its quantisers from ITS shard; adopt rank `src`'s ranges (delta,
    zero_point, the EMA range trackers) everywhere so that the averaged-gradient trajectories start from one point."""
    import torch.distributed as dist
    from .adaptive_rounding import AdaRoundQuantizer
    from .quant_layer import UniformAffineQuantizer
    quants = [m for m in qnn.modules() if isinstance(m, (UniformAffineQuantizer, AdaRoundQuantizer))]
    names = ("delta", "zero_point", "x_min", "x_max", "inited")
    payload = [None]
    if dist.get_rank() == src:
        cpu = lambda v: v.detach().cpu() if torch.is_tensor(v) else v
        payload = [[{n: cpu(getattr(q, n, None)) for n in names} for q in quants]]
    dist.broadcast_object_list(payload, src=src)
    if dist.get_rank() != src:
        dev = next(qnn.parameters()).device
        for q, st in zip(quants, payload[0]):
            for n in names:
                if not hasattr(q, n) and st[n] is None:
                    continue
                v = st[n].to(dev) if torch.is_tensor(st[n]) else st[n]
                if n == "delta" and torch.is_tensor(v) and getattr(q, "leaf_param", False):
                    v = torch.nn.Parameter(v)                  # activation step sizes are trainable leaves
                q._parameters.pop(n, None)
                q.__dict__.pop(n, None)
                setattr(q, n, v)
    for m in qnn.modules():
        if isinstance(m, QuantModule):
            m.invalidate()


def _to_dev(qnn, *ts):
    dev = next(qnn.parameters()).device
    return tuple(t.to(dev) for t in ts)


def calibrate_model(qnn, cali_data, cond=False, quant_act=True, cali_batch_size=32, cali_iters=20000, cali_iters_a=5000,
                    cali_lr=4e-4, cali_p=2.4, running_stat=False, rs_sm_only=False, init_batch=8, act_init_batch=16,
                    resume_w=False, is_sm=False, on_unit=None, multi_gpu=False):
    """The scripts' calibration sequence; returns the reference-format state dict (what `torch.save(qnn.state_dict())`
    writes there after the Parameter wrapping of delta / zero_point).  cali_data = (xs, ts[, conds]).
    multi_gpu: data-parallel calibration, one process per GPU (torch.distributed initialised by the caller, backend "nccl" =
    RCCL): every rank passes ITS shard of the calibration samples and the same seed; gradients are averaged by an
    all-reduce every iteration, so all ranks end with identical parameters (tests/test_calibration.py, 2 ranks over gloo)."""
    from . import engine
    from .utils import export_cali_state_dict
    with engine.simulation():           # quantiser initialisation and range tracking see the reference's fp32 arithmetic
        _calibrate(qnn, cali_data, cond, quant_act, cali_batch_size, cali_iters, cali_iters_a, cali_lr, cali_p, running_stat,
                   rs_sm_only, init_batch, act_init_batch, resume_w, is_sm, on_unit, multi_gpu)
    return export_cali_state_dict(qnn)


def _calibrate(qnn, cali_data, cond, quant_act, cali_batch_size, cali_iters, cali_iters_a, cali_lr, cali_p, running_stat,
               rs_sm_only, init_batch, act_init_batch, resume_w, is_sm, on_unit, multi_gpu=False):
    xs, ts = cali_data[0], cali_data[1]
    cs = cali_data[2] if cond else None

    def run(sel):
        args = _to_dev(qnn, xs[sel], ts[sel]) + (_to_dev(qnn, cs[sel]) if cond else ())
        return qnn(*args)

    if not resume_w:
        logger.info("Initializing weight quantization parameters")
        qnn.set_quant_state(True, False)
        with torch.no_grad():
            run(slice(0, init_batch))
        if multi_gpu:
            sync_quantisers(qnn)
        kwargs = dict(cali_data=cali_data, batch_size=cali_batch_size, iters=cali_iters, weight=0.01, asym=True,
                      b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode='mse', cond=cond, is_sm=is_sm, multi_gpu=multi_gpu)
        logger.info("Doing weight calibration")
        recon_model(qnn, on_unit=on_unit, **kwargs)
        qnn.set_quant_state(weight_quant=True, act_quant=False)
    if quant_act:
        logger.info("Doing activation calibration")
        qnn.set_quant_state(True, True)
        with torch.no_grad():
            inds = np.random.choice(xs.shape[0], min(act_init_batch, xs.shape[0]), replace=False)
            run(torch.as_tensor(inds))
            if running_stat:
                logger.info('Running stat for activation quantization')
                order = np.arange(xs.shape[0])
                np.random.shuffle(order)
                qnn.set_running_stat(True, rs_sm_only)
                for i in range(int(xs.size(0) / act_init_batch)):
                    run(torch.as_tensor(order[i * act_init_batch:(i + 1) * act_init_batch]))
                qnn.set_running_stat(False, rs_sm_only)
        if multi_gpu:
            sync_quantisers(qnn)
        kwargs = dict(cali_data=cali_data, batch_size=cali_batch_size, iters=cali_iters_a, act_quant=True, opt_mode='mse',
                      lr=cali_lr, p=cali_p, cond=cond, is_sm=is_sm, multi_gpu=multi_gpu)
        recon_model(qnn, on_unit=on_unit, **kwargs)
        qnn.set_quant_state(weight_quant=True, act_quant=True)
