#!/usr/bin/env python3
"""Envelope of the opt-in fp16 activation stream on the six whole-UNet fixtures (what tests/test_engine_models.py::
test_fp16_activation_stream_envelope asserts), as a JSON summary for profiles/ — bench.py attaches the newest
profiles/*_fp16_envelope.json to `other_configs.sd_fp16_stream`.

    python tools/fp16_envelope.py gpurun_out/r05_xx/fp16_envelope.json        (on an MI355X)

ratio = max|engine(fp16 stream) - fp64 evaluation| / max|reference fp32 - fp64 evaluation|   (stated bound: 2.0; the fp32 stream's is 1.25)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "q-diffusion_amd"), ROOT):
    sys.path.insert(0, p)

import torch  # noqa: E402
from golden_util import load_fixture  # noqa: E402
import test_engine_models as T  # noqa: E402


def main():
    from qdiff import engine
    dev = torch.device("cuda", 0)
    out = {"bound": 2.0, "fixtures": {}}
    for name in T.TINY + T.FULL:
        fx = load_fixture(f"model_{name}.pt")
        qnn = T._resume(fx, dev)
        y32 = T._run(qnn, fx, dev)
        engine.set_stream_dtype(torch.float16)
        try:
            y16 = T._run(qnn, fx, dev)
        finally:
            engine.set_stream_dtype(torch.float32)
        y64 = T._oracle64(fx)
        d16, _, mx = T._metrics(y16.double(), y64)
        d32, _, _ = T._metrics(y32.double(), y64)
        dself, _, _ = T._metrics(fx["out_wa"].double(), y64)
        out["fixtures"][name] = {"fp16_stream_over_reference": round(d16 / dself, 3), "fp32_stream_over_reference": round(d32 / dself, 3),
                                 "reference_fp32_vs_fp64_of_range": float(f"{dself / mx:.3e}")}
        print(name, out["fixtures"][name], flush=True)
        del qnn
        torch.cuda.empty_cache()
    try:
        out["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
    except OSError:
        out["commit"] = None
    out["sd_full_ratio"] = out["fixtures"]["sd_full"]["fp16_stream_over_reference"]
    path = sys.argv[1] if len(sys.argv) > 1 else "fp16_envelope.json"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
