"""layer_reconstruction — counterpart of the reference's qdiff/layer_recon.py:13-119 (the first / last convolutions and any
QuantModule outside a block); the loop lives in qdiff/recon.py."""
from .quant_layer import QuantModule
from .recon import reconstruct


def layer_reconstruction(model, layer: QuantModule, cali_data, batch_size: int = 32, iters: int = 20000,
                         weight: float = 0.001, opt_mode: str = 'mse', asym: bool = False, include_act_func: bool = True,
                         b_range: tuple = (20, 2), warmup: float = 0.0, act_quant: bool = False, lr: float = 4e-5,
                         p: float = 2.0, multi_gpu: bool = False, cond: bool = False, is_sm: bool = False):
    """Optimise the output of a single quantised layer (AdaRound); parameters as block_reconstruction."""
    reconstruct(model, layer, cali_data, batch_size=batch_size, iters=iters, weight=weight, opt_mode=opt_mode, asym=asym,
                include_act_func=include_act_func, b_range=b_range, warmup=warmup, act_quant=act_quant, lr=lr, p=p,
                multi_gpu=multi_gpu, cond=cond, is_sm=is_sm)
