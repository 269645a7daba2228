#!/usr/bin/env python3
"""Run the reference's OWN, unmodified `scripts/sample_diffusion_ddim.py` (its `__main__` block, through runpy) on the
host, with the `qdiff` package taken from a chosen root — the reference's tree or this repository — and store the images
it would have written.  TEST INFRASTRUCTURE (tests/test_reference_scripts.py starts it twice in subprocesses; both
packages are called `qdiff`, so they cannot share an interpreter).

    python tests/run_reference_script.py <qdiff_root> <fp_ckpt> <out.pt> [--emulator] -- <script arguments...>

Build-container only (needs /root/reference).  What is substituted, all on the test side:
  * third-party modules that are not installed here and that the script only touches at its edges:
    `pytorch_lightning.seed_everything` (seeds python / numpy / torch, as the original), `torchvision.utils.save_image`
    (keeps the tensor instead of encoding a PNG);
  * `ddim.functions.ckpt_util.get_ckpt_path` (a download in the original) returns <fp_ckpt>, a state dict of key-derived
    synthetic weights written by the test;
  * the literal 'cuda' devices of `ddim/functions/denoising.py:24,30` and `qdiff/utils.py:390-393` map to "stay where
    you are" on this CPU-only container;
  * --emulator: this repository's C-ABI emulator (tests/abi_emulator.py) stands in for libqdiff_hip.so, so that the
    quantised-activation state can execute without a GPU.
"""
import os
import random
import runpy
import sys
import types

REF = "/root/reference"
sys.dont_write_bytecode = True          # the reference tree is read-only for this project: no __pycache__ next to its sources


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    head, script_args = argv[:cut], argv[cut + 1:]
    qdiff_root, fp_ckpt, out_path = head[:3]
    emulator = "--emulator" in head[3:]
    sys.path[:0] = [p for p in (qdiff_root, REF) if p not in sys.path[:2]]

    import numpy as np
    import torch

    pl = types.ModuleType("pytorch_lightning")

    def seed_everything(seed):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed
    pl.seed_everything = seed_everything
    sys.modules["pytorch_lightning"] = pl

    def stub(name, **attrs):
        """A stand-in for a module the script imports but never uses on this path: any attribute is an empty class."""
        m = types.ModuleType(name)
        m.__path__ = []
        m.__getattr__ = lambda attr: m.__dict__.setdefault(attr, type(attr, (), {})) if not attr.startswith("__") else (_ for _ in ()).throw(AttributeError(attr))
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    images = []
    # ddim/datasets/__init__.py imports the dataset zoo (torchvision datasets / transforms, lmdb, PIL) at module level; the
    # sampling script only takes `inverse_data_transform` from it
    tv = stub("torchvision")
    for sub in ("transforms", "transforms.functional", "datasets", "datasets.utils"):
        setattr(tv, sub.split(".")[0], sys.modules.get("torchvision." + sub.split(".")[0]) or stub("torchvision." + sub.split(".")[0]))
        stub("torchvision." + sub) if "torchvision." + sub not in sys.modules else None
    tv.utils = stub("torchvision.utils", save_image=lambda t, path, **k: images.append((os.path.basename(path), t.detach().clone())))
    for name in ("lmdb", "PIL", "PIL.Image"):
        try:
            __import__(name)
        except ImportError:
            stub(name)

    to = torch.Tensor.to

    def to_host(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            a = a[1:]
            if not a and not k:
                return self
        return to(self, *a, **k)
    torch.Tensor.to = to_host
    torch.Tensor.cuda = lambda self, *a, **k: self

    import ddim.functions.ckpt_util as cu
    cu.get_ckpt_path = lambda *a, **k: fp_ckpt

    import qdiff
    assert os.path.abspath(os.path.dirname(os.path.dirname(qdiff.__file__))) == os.path.abspath(qdiff_root), qdiff.__file__
    if emulator:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import abi_emulator
        from _pytest.monkeypatch import MonkeyPatch
        abi_emulator.install(MonkeyPatch())

    sys.argv = [os.path.join(REF, "scripts", "sample_diffusion_ddim.py")] + script_args
    runpy.run_path(sys.argv[0], run_name="__main__")
    images.sort(key=lambda kv: int(os.path.splitext(kv[0])[0]))
    torch.save({"names": [n for n, _ in images], "images": torch.stack([t for _, t in images]), "qdiff": qdiff.__file__}, out_path)


if __name__ == "__main__":
    main()
