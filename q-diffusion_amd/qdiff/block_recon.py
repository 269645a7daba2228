"""block_reconstruction — importable for script compatibility (reference qdiff/block_recon.py:13-17).

BRECQ/AdaRound block reconstruction is the *offline producer* of the calibrated checkpoint (20k Adam
iterations through the fp32 fake-quant graph); it is outside the hot path this engine accelerates
(SURVEY.md §2 row 9, §8(f) N2).  The quantiser classes keep their differentiable simulation
(`UniformAffineQuantizer.forward`, `AdaRoundQuantizer.forward` with soft targets), so the reference's
own qdiff/block_recon.py can be run against this package; this stub only reserves the name.
"""


def block_reconstruction(model, block, cali_data, batch_size=32, iters=20000, weight=0.01, opt_mode='mse',
                         asym=False, include_act_func=True, b_range=(20, 2), warmup=0.0, act_quant=False,
                         lr=4e-5, p=2.0, multi_gpu=False, cond=False, is_sm=False):
    raise NotImplementedError(
        "calibration (block reconstruction) is an offline step outside this engine's scope; calibrate with the "
        "reference implementation and load the checkpoint with qdiff.utils.resume_cali_model")
