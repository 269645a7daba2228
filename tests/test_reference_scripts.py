"""The reference's own, UNMODIFIED sampling script driven by this repository's `qdiff` ("so
sample_diffusion_ddim.py ... run unmodified"; BASELINE.json configs[0]: CIFAR-10 DDIM UNet, W8 weights-only, on CPU).

`scripts/sample_diffusion_ddim.py` is executed twice through tests/run_reference_script.py — its `__main__` block, with its
own argument parser, `configs/cifar10.yml`, `Diffusion.sample()`, `ddim/functions/denoising.generalized_steps` — once with
`qdiff` resolving to the reference's package and once with it resolving to q-diffusion_amd/qdiff.  Same seed, same
synthetic fp32 checkpoint, same reference-format calibrated checkpoint (`--resume --cali_ckpt`):

  * weights-only W8 (`--ptq --weight_bit 8 --split`): the images the script writes are BIT-IDENTICAL;
  * W8A8 (`--quant_act --act_bit 8 --a_sym`): this package runs the integer engine (on the C-ABI emulator: no GPU here),
    the reference its fp32 fake-quant simulation; the script's whole loop completes and the two pictures agree as
    pictures (see the comment at the assertion for why nothing sharper can be asked of clamped images of a random-weight
    network).

Build container only: skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest
import torch

from golden_util import build_ckpt, load_fixture

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isfile(os.path.join(REF, "scripts", "sample_diffusion_ddim.py")):
    pytest.skip("reference tree not present (GPU box)", allow_module_level=True)


def _run(qdiff_root, fp_ckpt, out, logdir, extra, emulator=False):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "run_reference_script.py"), qdiff_root, fp_ckpt, out]
    cmd += ["--emulator"] if emulator else []
    cmd += ["--", "--config", os.path.join(os.path.dirname(fp_ckpt), "cifar10_batch2.yml"), "--timesteps", "4", "--eta", "0", "--skip_type", "quad", "--max_images", "2",
            "--ptq", "--quant_mode", "qdiff", "--split", "--resume", "-l", logdir, "--seed", "1234"] + extra
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="8")
    env.pop("PYTHONPATH", None)
    r = subprocess.run(cmd, cwd=REF, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out, weights_only=False)


@pytest.fixture(scope="module")
def ckpts(tmp_path_factory):
    """(fp32 'pretrained' state dict, reference-format calibrated checkpoint) of the CIFAR-10 UNet with key-derived weights."""
    d = tmp_path_factory.mktemp("ref_script")
    fx = load_fixture("model_cifar_full.pt")
    cali = build_ckpt(fx)
    fp = {k[len("model."):]: v for k, v in cali.items()
          if k.startswith("model.") and k.rsplit(".", 1)[-1] not in ("alpha", "delta", "zero_point")}
    # QuantModule keeps the fp32 tensors under the wrapped layer's own key (…conv1.weight): exactly the fp model's state dict
    fp_path, cali_path = str(d / "ema_cifar10.pth"), str(d / "cali.pth")
    torch.save(fp, fp_path)
    torch.save(cali, cali_path)
    # the script's own config (configs/cifar10.yml) with 2 images per sampling round instead of 64: a config file is an
    # input of the script, and 64-image rounds of the 32 x 32 UNet are too slow for a unit test on the host
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "cifar10.yml")))
    cfg["sampling"]["batch_size"] = 2
    yaml.safe_dump(cfg, open(d / "cifar10_batch2.yml", "w"))
    return d, fp_path, cali_path


def test_ddim_script_weights_only_is_bit_identical(ckpts):
    d, fp_path, cali_path = ckpts
    args = ["--weight_bit", "8", "--cali_ckpt", cali_path]
    ref = _run(REF, fp_path, str(d / "ref_w.pt"), str(d / "log_ref_w"), args)
    ours = _run(os.path.join(ROOT, "q-diffusion_amd"), fp_path, str(d / "our_w.pt"), str(d / "log_our_w"), args)
    assert "/root/reference/qdiff" in ref["qdiff"] and "q-diffusion_amd/qdiff" in ours["qdiff"]
    assert ref["names"] == ours["names"] == ["0.png", "1.png"]
    assert torch.isfinite(ref["images"]).all() and ref["images"].std() > 0
    assert torch.equal(ref["images"], ours["images"])


def test_ddim_script_w8a8_runs_on_the_integer_engine(ckpts):
    d, fp_path, cali_path = ckpts
    args = ["--weight_bit", "8", "--quant_act", "--act_bit", "8", "--a_sym", "--cali_ckpt", cali_path]
    ref = _run(REF, fp_path, str(d / "ref_wa.pt"), str(d / "log_ref_wa"), args)
    ours = _run(os.path.join(ROOT, "q-diffusion_amd"), fp_path, str(d / "our_wa.pt"), str(d / "log_our_wa"), args, emulator=True)
    a, b = ref["images"].double(), ours["images"].double()
    assert a.shape == b.shape and torch.isfinite(b).all()
    cos = torch.nn.functional.cosine_similarity((a - a.mean()).flatten(), (b - b.mean()).flatten(), dim=0).item()
    far = ((a - b).abs() > 0.05).double().mean().item()
    # What this can and cannot show: with key-derived random weights the 4-step sampler divides by sqrt(alpha_t) of t = 800
    # (x 58) and the script clamps the result to [0, 1] — 97 % of the pixels saturate, and a pixel near the clamp boundary
    # moves by O(1) for an O(1e-2) change of eps.  The assertion is therefore about the PATTERN (the integer engine drove the
    # script's whole loop and produced the same picture: measured 2.7 % of the pixels further than 0.05 apart, cosine
    # 0.995); value-level parity of the W8A8 state is the business of the model- and block-level tests.
    assert far <= 0.08 and cos >= 0.99, (far, cos)
