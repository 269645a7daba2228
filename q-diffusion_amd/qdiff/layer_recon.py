"""layer_reconstruction — importable for script compatibility (reference qdiff/layer_recon.py:13-17).
See block_recon.py: calibration is the offline producer of the checkpoint, out of scope here."""


def layer_reconstruction(model, layer, cali_data, batch_size=32, iters=20000, weight=0.001, opt_mode='mse',
                         asym=False, include_act_func=True, b_range=(20, 2), warmup=0.0, act_quant=False,
                         lr=4e-5, p=2.0, multi_gpu=False, cond=False, is_sm=False):
    raise NotImplementedError(
        "calibration (layer reconstruction) is an offline step outside this engine's scope; calibrate with the "
        "reference implementation and load the checkpoint with qdiff.utils.resume_cali_model")
