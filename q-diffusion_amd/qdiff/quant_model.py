"""QuantModel — the drop-in boundary (reference qdiff/quant_model.py:12-96).

`QuantModel(model, weight_quant_params, act_quant_params, **kwargs)` rewrites `model` in place:
every Conv2d / Conv1d / Linear becomes a QuantModule and every known UNet block its quantised
counterpart; the wrapped model stays reachable as `.model` and the state-dict schema is the
reference's.  On top of the reference surface it offers
`capture_graph()/forward` replay of a whole UNet evaluation as one HIP graph (qdiff/graph.py).
"""
import logging

import torch
import torch.nn as nn

from .quant_block import (BaseQuantBlock, QuantAttentionBlock, QuantAttnBlock, QuantBasicTransformerBlock,
                          QuantQKMatMul, QuantResBlock, QuantSMVMatMul, get_specials)
from .quant_layer import QuantModule, StraightThrough
from .arch import ldm_unet

logger = logging.getLogger(__name__)


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
        self._graphs = None
        self._quant_state = (False, False)

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
            elif target is QuantAttentionBlock and isinstance(child, ldm_unet.AttentionBlock):
                setattr(module, name, target(child, act_quant_params, sm_abit=self.sm_abit, quant_matmuls=self.quant_act))
            else:
                setattr(module, name, target(child, act_quant_params))

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self._quant_state = (bool(weight_quant), bool(act_quant))
        for m in self.model.modules():
            if isinstance(m, (QuantModule, BaseQuantBlock)):
                m.set_quant_state(weight_quant, act_quant)

    def forward(self, x, timesteps=None, context=None):
        if self._graphs is not None and not torch.is_grad_enabled() and torch.is_tensor(timesteps) and x.is_cuda:
            from .graph import GraphedUNet, signature
            key = signature(x, timesteps, context) + (self._quant_state,)
            g = self._graphs.get(key)
            if g is None:
                g = self._graphs[key] = GraphedUNet(self, x, timesteps, context)
            return g(x, timesteps, context).clone()
        return self.model(x, timesteps, context)

    def invalidate_plans(self):
        """Forget every packed weight / epilogue constant / captured graph (see QuantModule.invalidate)."""
        for m in self.model.modules():
            if isinstance(m, QuantModule):
                m.invalidate()
            m.__dict__.pop("_attn_plan_cache", None)
        if self._graphs is not None:
            self._graphs = {}

    def enable_hip_graphs(self, on: bool = True):
        """Replay each (shape, quant-state) UNet evaluation as one HIP graph (qdiff/graph.py).
        Quantiser parameters are baked into device tensors at capture; call again (or toggle the
        quant state) after changing them."""
        self._graphs = {} if on else None

    def set_running_stat(self, running_stat: bool, sm_only=False):
        """reference :71-87"""
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
        btb = tuple(t for t, q in self.specials.items() if q is QuantBasicTransformerBlock)
        for _, m in self.model.named_modules():
            if isinstance(m, (QuantBasicTransformerBlock,) + btb):
                m.checkpoint = grad_ckpt
