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
    config.addinivalue_line("markers", "late_runner: the test that runs the `late` GPU tests in a child process (collected last of all)")
    config.addinivalue_line("markers", "late: collected last (end-to-end jobs whose failure must not hide the kernel parity "
                                       "tests behind `-x`)")


def pytest_collection_modifyitems(config, items):
    late = [it for it in items if it.get_closest_marker("late")]
    if late:
        # among themselves: single kernels first, then the (independent) calibration job, then the whole models that need
        # those kernels — `-x` stops at the first failure, and a failing kernel test says more than the model tests it
        # would take down with it
        rank = {"test_hip_kernels.py": 0, "test_calibration.py": 1, "test_engine_models.py": 2, "test_block_parity.py": 3}
        late.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), 4))
        items[:] = [it for it in items if not it.get_closest_marker("late")] + late
        if os.environ.get("QDIFF_RUN_LATE") != "1":
            # GPU tests of code that no GPU has executed yet run in a CHILD interpreter (tests/test_zz_late_gpu.py): a device
            # fault there ends the child, not the run that holds the verified tests' results
            skip = pytest.mark.skip(reason="runs inside test_late_gpu_tests_in_a_child_process (QDIFF_RUN_LATE=1 runs it here)")
            for it in late:
                if it.get_closest_marker("gpu"):
                    it.add_marker(skip)
    runner = [it for it in items if it.get_closest_marker("late_runner")]
    if runner:
        items[:] = [it for it in items if not it.get_closest_marker("late_runner")] + runner


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from qdiff import hip
    hip.load()  # fail loudly if the extension is missing on a GPU box
    assert hip.available(), "libqdiff_hip.so loaded but no gfx950 device is usable"
    return torch.device("cuda:0")
