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
