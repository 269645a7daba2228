"""AdaRound weight quantiser (https://arxiv.org/abs/2004.10568) — counterpart of the reference's
qdiff/adaptive_rounding.py with the same constructor, attributes (`alpha`, `delta`, `zero_point`,
`soft_targets`, `n_levels`, `round_mode`, ...) and state-dict keys.

At inference (round_mode='learned_hard_sigmoid', soft_targets=False) the codes are
    W = clamp(floor(w/delta) + (alpha >= 0) + zero_point, 0, n_levels-1)
(reference adaptive_rounding.py:49-59).  The integer engine evaluates that formula once, inside
qd_pack_weights (csrc/quantize.hip), instead of on every forward; `forward` below is the
differentiable fp32 simulation that calibration (block/layer reconstruction) optimises through.
"""
import logging

import torch
from torch import nn

from .quant_layer import UniformAffineQuantizer, round_ste

logger = logging.getLogger(__name__)


class AdaRoundQuantizer(nn.Module):
    # see UniformAffineQuantizer._STATE_ATTRS (resume_cali_model re-attaches delta / zero_point as plain tensors, reference
    # utils.py:409-417: QuantModel's packed weights, HIP graphs and prepared contexts must notice)
    _STATE_ATTRS = frozenset(("alpha", "delta", "zero_point", "soft_targets", "round_mode", "n_bits", "n_levels", "sym"))

    def __setattr__(self, name, value):
        if name in self._STATE_ATTRS:
            from . import engine
            engine.bump_state()
        super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in self._STATE_ATTRS:
            from . import engine
            engine.bump_state()
        super().__delattr__(name)

    def __init__(self, uaq: UniformAffineQuantizer, weight_tensor: torch.Tensor, round_mode='learned_round_sigmoid'):
        super().__init__()
        # inherit the uniform quantiser's grid
        self.n_bits = uaq.n_bits
        self.sym = uaq.sym
        self.delta = uaq.delta
        self.zero_point = uaq.zero_point
        self.n_levels = uaq.n_levels
        self.round_mode = round_mode
        self.alpha = None
        self.soft_targets = False
        # rectified-sigmoid stretch parameters
        self.gamma, self.zeta = -0.1, 1.1
        self.beta = 2 / 3
        self.init_alpha(x=weight_tensor.clone())

    def rounding(self, x):
        """Integer (pre-zero-point) codes for the active rounding mode."""
        scaled = x / self.delta
        if self.round_mode == 'nearest':
            return torch.round(scaled)
        if self.round_mode == 'nearest_ste':
            return round_ste(scaled)
        if self.round_mode == 'stochastic':
            base = torch.floor(scaled)
            logger.info('Draw stochastic sample')
            return base + torch.bernoulli(scaled - base)
        if self.round_mode == 'learned_hard_sigmoid':
            base = torch.floor(scaled)
            up = self.get_soft_targets() if self.soft_targets else (self.alpha >= 0).float()
            return base + up
        raise ValueError('Wrong rounding mode')

    def forward(self, x):
        codes = torch.clamp(self.rounding(x) + self.zero_point, 0, self.n_levels - 1)
        return (codes - self.zero_point) * self.delta

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def init_alpha(self, x: torch.Tensor):
        """alpha such that the rectified sigmoid equals the fractional part of w/delta
        (reference adaptive_rounding.py:66-74)."""
        if self.round_mode != 'learned_hard_sigmoid':
            raise NotImplementedError
        scaled = x / self.delta
        rest = scaled - torch.floor(scaled)
        self.alpha = nn.Parameter(-torch.log((self.zeta - self.gamma) / (rest - self.gamma) - 1))

    def extra_repr(self):
        return f'bit={self.n_bits}, symmetric={self.sym}, round_mode={self.round_mode}'
