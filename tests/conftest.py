import os
import sys

import pytest

# Some tests import the reference's `ldm` / `ddim` packages from /root/reference, which is read-only for this project:
# no byte-code caches next to its sources — in this process and in every process it spawns (gloo workers, script runs).
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "q-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from qdiff import hip
    hip.load()  # fail loudly if the extension is missing on a GPU box
    assert hip.available(), "libqdiff_hip.so loaded but no gfx950 device is usable"
    return torch.device("cuda:0")
