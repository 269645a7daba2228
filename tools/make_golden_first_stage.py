#!/usr/bin/env python3
"""tests/golden/first_stage.pt: the REAL reference Decoder (ldm/modules/diffusionmodules/model.py:465-572) with key-derived
synthetic weights on seeded latents, CPU.  Build container only:  python tools/make_golden_first_stage.py
(AutoencoderKL / VQModelInterface themselves need pytorch_lightning and taming-transformers, which are not installed;
their decode() is `decoder(post_quant_conv(z))`, autoencoder.py:274-283,330-333 — post_quant_conv is applied here with
torch.nn.Conv2d exactly as they do.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402

from ldm.modules.diffusionmodules.model import Decoder  # noqa: E402  (the reference's)

CASES = {
    # a small KL-f8-shaped decoder with attention in the mid block AND in one up stage
    "kl_tiny": dict(embed_dim=4, z=(2, 4, 8, 8),
                    dd=dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2],
                            num_res_blocks=1, attn_resolutions=[8], dropout=0.0)),
    # VQ-f4 shaped (no attention in the up path), tanh head off
    "vq_tiny": dict(embed_dim=3, z=(1, 3, 8, 8),
                    dd=dict(double_z=False, z_channels=3, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                            num_res_blocks=2, attn_resolutions=[], dropout=0.0)),
}


def main():
    fx = {}
    for name, c in CASES.items():
        dec = Decoder(**c["dd"]).eval()
        pq = torch.nn.Conv2d(c["embed_dim"], c["dd"]["z_channels"], 1).eval()
        dec.load_state_dict({k: MG.synthetic.tensor_for("decoder." + k, v.shape, seed=0) for k, v in dec.state_dict().items()})
        pq.load_state_dict({k: MG.synthetic.tensor_for("post_quant_conv." + k, v.shape, seed=0) for k, v in pq.state_dict().items()})
        g = torch.Generator().manual_seed(77)
        z = torch.randn(c["z"], generator=g)
        with torch.no_grad():
            out = dec(pq(z))
        fx[name] = dict(dd=c["dd"], embed_dim=c["embed_dim"], z=z, out=out.clone())
        print(f"[golden] first_stage {name}: out {tuple(out.shape)} |max| {out.abs().max():.4f}")
    fx["torch_version"] = torch.__version__
    torch.save(fx, os.path.join(MG.OUT, "first_stage.pt"))


if __name__ == "__main__":
    main()
