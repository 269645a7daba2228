#!/usr/bin/env python3
"""Run one of the reference's OWN, unmodified sampling scripts (its `__main__` block, through runpy) on the host, with the
`qdiff` package taken from a chosen root — the reference's tree or this repository — and store the images it wrote.
TEST INFRASTRUCTURE (tests/test_reference_scripts.py starts it in subprocesses: both packages are called `qdiff`, so they
cannot share an interpreter).

    python tests/run_reference_script.py <qdiff_root> ddim <fp_ckpt> <out.pt> [--emulator] -- <script arguments...>
    python tests/run_reference_script.py <qdiff_root> ldm  -         <out.pt> [--emulator] -- <script arguments...>
    python tests/run_reference_script.py <qdiff_root> txt2img -      <out.pt> [--emulator] -- <script arguments...>

Build-container only (needs /root/reference).  What is substituted, all on the test side, identically for both roots:
  * third-party packages that are not installed here and that the scripts touch only at their edges:
    `pytorch_lightning` (seed_everything seeds python / numpy / torch as the original; LightningModule = nn.Module with a
    `.device`), `omegaconf` (load / merge / from_dotlist of plain YAML into attribute dictionaries), `torchvision`
    (save_image keeps the tensor; the dataset zoo that ddim/datasets imports at module level is never used),
    `taming`'s VectorQuantizer2 (this repository's restatement, qdiff/arch/first_stage.py, loaded by file path), `lmdb`;
  * ddim: `ddim.functions.ckpt_util.get_ckpt_path` (a download in the original) returns <fp_ckpt>;
  * txt2img: what surrounds the denoising loop and downloads weights in the original — the safety checker (`diffusers`,
    `transformers.AutoFeatureExtractor`; its call is commented out in the script anyway), the invisible watermark
    (`imwatermark`, `cv2`: identity here) and the CLIP text encoder, which the test's config replaces by
    `qd_script_stubs.TextEncoder` (a deterministic [77, 768] embedding per prompt string);
  * the literal 'cuda' devices of `ddim/functions/denoising.py:24,30`, `qdiff/utils.py:390-393` and the scripts'
    `.cuda()` calls map to "stay where you are" on this CPU-only container;
  * --emulator: this repository's C-ABI emulator (tests/abi_emulator.py) stands in for libqdiff_hip.so, so that the
    quantised-activation state can execute without a GPU.
"""
import glob
import importlib.util
import os
import random
import runpy
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True          # the reference tree is read-only for this project: no __pycache__ next to its sources


def stub(name, **attrs):
    """A stand-in for a module a script imports but never uses on this path: any attribute is an empty class."""
    m = types.ModuleType(name)
    m.__path__ = []

    def missing(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return m.__dict__.setdefault(attr, type(attr, (), {}))
    m.__getattr__ = missing
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_omegaconf():
    import yaml

    class DictConfig(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    class ListConfig(list):
        pass

    def wrap(v):
        if isinstance(v, dict):
            return DictConfig({k: wrap(x) for k, x in v.items()})
        if isinstance(v, (list, tuple)):
            return ListConfig(wrap(x) for x in v)
        return v

    def _merge(*cfgs):
        out = DictConfig()
        for c in cfgs:
            for k, v in c.items():
                out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else wrap(v)
        return out

    def _from_dotlist(items):
        out = {}
        for it in items:
            key, val = it.split("=", 1)
            node = out
            parts = key.lstrip("-").split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = yaml.safe_load(val)
        return wrap(out)

    class OmegaConf:
        load = staticmethod(lambda path: wrap(yaml.safe_load(open(path))))
        merge = staticmethod(_merge)
        from_dotlist = staticmethod(_from_dotlist)
        create = staticmethod(wrap)
        to_container = staticmethod(lambda c, **k: c)

    lc = stub("omegaconf.listconfig", ListConfig=ListConfig)
    stub("omegaconf", OmegaConf=OmegaConf, ListConfig=ListConfig, DictConfig=DictConfig, listconfig=lc)


def install_lightning(torch, np):
    def seed_everything(seed):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    dist = stub("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    util = stub("pytorch_lightning.utilities", distributed=dist)
    stub("pytorch_lightning", seed_everything=seed_everything, LightningModule=LightningModule, utilities=util, __version__="1.4.2")


def install_taming():
    spec = importlib.util.spec_from_file_location("amd_first_stage", os.path.join(ROOT, "q-diffusion_amd", "qdiff", "arch", "first_stage.py"))
    fs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fs)

    class VectorQuantizer2(fs.VectorQuantizer):
        """taming.modules.vqvae.quantize.VectorQuantizer2's constructor and (z_q, loss, info) return value."""

        def __init__(self, n_e, e_dim, beta=0.25, remap=None, unknown_index="random", sane_index_shape=False, legacy=True):
            super().__init__(n_e, e_dim)

        def forward(self, z, *a, **k):
            idx = self.indices(z)
            b, c, h, w = z.shape
            return self.get_codebook_entry(idx, (b, h, w, c)), None, (None, None, idx)

    q = stub("taming.modules.vqvae.quantize", VectorQuantizer2=VectorQuantizer2)
    v = stub("taming.modules.vqvae", quantize=q)
    mo = stub("taming.modules", vqvae=v)
    stub("taming", modules=mo)


def install_txt2img_edges(torch):
    import zlib

    class Pretrained:
        """from_pretrained() without a download; called like the safety checker / feature extractor."""

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

        def __call__(self, *a, images=None, **k):
            if images is not None:
                return images, [False] * len(images)
            return types.SimpleNamespace(pixel_values=None)

    class WatermarkEncoder:
        def set_watermark(self, *a, **k):
            pass

        def encode(self, img, method):
            return img

    class TextEncoder(torch.nn.Module):
        """Stands where FrozenCLIPEmbedder stands (ldm/modules/encoders/modules.py:137-170): prompts -> [B, 77, 768]."""

        def encode(self, prompts):
            out = []
            for p in prompts:
                g = torch.Generator().manual_seed(zlib.crc32(p.encode()))
                out.append(torch.randn(77, 768, generator=g))
            return torch.stack(out)

    stub("cv2", cvtColor=lambda a, code: a[:, :, ::-1], COLOR_RGB2BGR=4)
    stub("imwatermark", WatermarkEncoder=WatermarkEncoder)
    sc = stub("diffusers.pipelines.stable_diffusion.safety_checker", StableDiffusionSafetyChecker=Pretrained)
    sdm = stub("diffusers.pipelines.stable_diffusion", safety_checker=sc)
    pm = stub("diffusers.pipelines", stable_diffusion=sdm)
    stub("diffusers", pipelines=pm)
    stub("transformers", AutoFeatureExtractor=Pretrained)
    stub("qd_script_stubs", TextEncoder=TextEncoder)


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    head, script_args = argv[:cut], argv[cut + 1:]
    qdiff_root, which, aux, out_path = head[:4]
    emulator = "--emulator" in head[4:]
    sys.path[:0] = [qdiff_root, REF]

    import numpy as np
    import torch

    install_lightning(torch, np)
    install_omegaconf()
    install_taming()
    images = []
    # ddim/datasets/__init__.py imports the dataset zoo (torchvision datasets / transforms, lmdb) at module level; the
    # sampling scripts only take `inverse_data_transform` / `make_grid` from those corners
    tv = stub("torchvision")
    for sub in ("transforms", "datasets"):
        setattr(tv, sub, stub("torchvision." + sub))
    stub("torchvision.transforms.functional")
    stub("torchvision.datasets.utils")
    tv.utils = stub("torchvision.utils", save_image=lambda t, path, **k: images.append((os.path.basename(path), t.detach().clone())))
    try:
        __import__("lmdb")
    except ImportError:
        stub("lmdb")

    to = torch.Tensor.to

    def is_cuda(d):
        return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")

    def to_host(self, *a, **k):
        if a and is_cuda(a[0]):
            a = a[1:]
        if is_cuda(k.get("device")):
            k = {kk: v for kk, v in k.items() if kk != "device"}
        if not a and not k:
            return self
        return to(self, *a, **k)
    torch.Tensor.to = to_host
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if which == "ddim":
        import ddim.functions.ckpt_util as cu
        cu.get_ckpt_path = lambda *a, **k: aux
    if which == "txt2img":
        install_txt2img_edges(torch)

    import qdiff
    assert os.path.abspath(os.path.dirname(os.path.dirname(qdiff.__file__))) == os.path.abspath(qdiff_root), qdiff.__file__
    if emulator:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import abi_emulator
        from _pytest.monkeypatch import MonkeyPatch
        abi_emulator.install(MonkeyPatch())

    script = {"ddim": "sample_diffusion_ddim.py", "ldm": "sample_diffusion_ldm.py", "txt2img": "txt2img.py"}[which]
    sys.argv = [os.path.join(REF, "scripts", script)] + script_args
    runpy.run_path(sys.argv[0], run_name="__main__")
    if which == "ddim":
        images.sort(key=lambda kv: int(os.path.splitext(kv[0])[0]))
        res = {"names": [n for n, _ in images], "images": torch.stack([t for _, t in images])}
    elif which == "txt2img":
        # the PNGs the script wrote: <outdir>/<time>/samples/00000.png ...
        from PIL import Image
        outdir = script_args[script_args.index("--outdir") + 1]
        pngs = sorted(glob.glob(os.path.join(outdir, "*", "samples", "*.png")))
        res = {"names": [os.path.basename(p) for p in pngs],
               "images": torch.stack([torch.from_numpy(np.array(Image.open(p).convert("RGB"))) for p in pngs])}
    else:
        # the script's own outputs: <logdir>/<run>/samples/<time>/numpy/<shape>-samples.npz (uint8 NHWC) and the PNGs
        logdir = script_args[script_args.index("-l") + 1]
        npz = sorted(glob.glob(os.path.join(logdir, "**", "numpy", "*-samples.npz"), recursive=True))
        pngs = sorted(glob.glob(os.path.join(logdir, "**", "img", "*.png"), recursive=True))
        assert len(npz) == 1, npz
        res = {"names": [os.path.basename(p) for p in pngs], "images": torch.from_numpy(np.load(npz[0])["arr_0"])}
    res["qdiff"] = qdiff.__file__
    torch.save(res, out_path)


if __name__ == "__main__":
    main()
