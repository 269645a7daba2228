"""GPU tests marked `late` cover code that was written after the round's last GPU minute: no GPU has executed it yet.  They
are skipped in the main run and executed here in a CHILD interpreter — a wrong result fails this test with the child's
report, and a device fault (which aborts the interpreter it happens in) cannot take the verified tests' results with it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.late_runner
def test_late_gpu_tests_in_a_child_process(cuda):
    if os.environ.get("QDIFF_RUN_LATE") == "1":
        pytest.skip("the late tests run in this interpreter")
    env = dict(os.environ, QDIFF_RUN_LATE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu and late", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    tail = (r.stdout[-6000:] + "\n" + r.stderr[-2000:])
    print(tail)
    assert r.returncode == 0, f"late GPU tests: exit code {r.returncode}\n{tail}"


# Experimental kernels behind environment knobs (default off; csrc/norm_quant.hip gn_apply_rows_kernel, csrc/igemm_dma.hip
# splitk_finalize4_kernel): plain streaming kernels without barriers or inter-block hand-offs, so running their parity here
# cannot hang the device; the knobs are read once per process, hence one child per setting.  The 3x3 halo-slab kernel
# (QDIFF_HALO=1) is NOT run here: its LDS-DMA pipeline with counted waits is exactly the kind of code whose first run
# belongs in a call of its own (tools/r03_first_call.sh).
@pytest.mark.gpu
@pytest.mark.late_runner
@pytest.mark.parametrize("knob,value,select", [("QD_GN_ROWS", "2", "concatenation or quantised_unet_matches_reference and cifar_tiny"),
                                               ("QD_GN_ROWS", "4", "concatenation or quantised_unet_matches_reference and sd_tiny"),
                                               ("QD_FIN_VEC", "1", "splitk")])
def test_experimental_streaming_kernels_in_a_child_process(cuda, knob, value, select):
    env = dict(os.environ)
    env[knob] = value
    env.pop("QDIFF_RUN_LATE", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_kernels.py"),
                        os.path.join(ROOT, "tests", "test_engine_models.py"), "-m", "gpu", "-q", "-k", select, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout[-4000:] + "\n" + r.stderr[-1500:])
    print(tail)
    assert r.returncode == 0, f"{knob}={value}: exit code {r.returncode}\n{tail}"
