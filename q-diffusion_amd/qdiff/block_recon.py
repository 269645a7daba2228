"""block_reconstruction — counterpart of the reference's qdiff/block_recon.py:13-166 (same signature, same arithmetic per
iteration); the loop itself lives in qdiff/recon.py, shared with layer_reconstruction."""
from .quant_block import BaseQuantBlock
from .recon import reconstruct


def block_reconstruction(model, block: BaseQuantBlock, cali_data, batch_size: int = 32, iters: int = 20000,
                         weight: float = 0.01, opt_mode: str = 'mse', asym: bool = False, include_act_func: bool = True,
                         b_range: tuple = (20, 2), warmup: float = 0.0, act_quant: bool = False, lr: float = 4e-5,
                         p: float = 2.0, multi_gpu: bool = False, cond: bool = False, is_sm: bool = False):
    """Optimise the output of one quantised block (BRECQ).  Parameters as in the reference: `cali_data` = (xs, ts[, conds]),
    `weight` of the rounding regulariser, `b_range` / `warmup` of its temperature, `asym` = quantised inputs against
    full-precision outputs, `act_quant` selects the activation-step-size phase (lr, p), `is_sm` see utils.save_inp_oup_data."""
    reconstruct(model, block, cali_data, batch_size=batch_size, iters=iters, weight=weight, opt_mode=opt_mode, asym=asym,
                include_act_func=include_act_func, b_range=b_range, warmup=warmup, act_quant=act_quant, lr=lr, p=p,
                multi_gpu=multi_gpu, cond=cond, is_sm=is_sm)
