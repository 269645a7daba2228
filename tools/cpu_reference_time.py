#!/usr/bin/env python3
"""CPU baseline through the UNMODIFIED reference (`PYTHONPATH=/root/reference`, its own QuantModel in the
(True, True) fake-quant state).  Build container only — the GPU box has no /root/reference, so bench.py's `cpu_baseline`
leg there times the oracle port instead; this script gives the number that leg cannot: warm-up + k timed evaluations of
ONE sample of the full-size model on this container's cores.

    python tools/cpu_reference_time.py [sd_full|ldm_full|cifar_full] [k=2]   -> one JSON line
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G   # noqa: E402  (puts /root/reference first on sys.path, stubs omegaconf, imports the reference's qdiff)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sd_full"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    spec = G.MODELS[name]
    wq, aq = G.quant_params(spec)
    qnn = G.QuantModel(G.build_fp(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    qnn.set_quant_state(True, True)
    x, t, c = G.inputs(spec, 1, seed=100)
    t0 = time.time()
    G.call(qnn, x, t, c)                    # data-dependent initialisation of every quantiser (not timed as an evaluation)
    init_s = time.time() - t0
    G.call(qnn, x, t, c)                    # warm-up
    ts = []
    for _ in range(k):
        t0 = time.time()
        G.call(qnn, x, t, c)
        ts.append(time.time() - t0)
    print(json.dumps({"model": name, "kind": "reference", "where": "build container (no GPU)", "cores": torch.get_num_threads(),
                      "torch": torch.__version__, "init_s": round(init_s, 1), "eval_s": [round(v, 2) for v in ts],
                      "eval_s_mean": round(sum(ts) / len(ts), 2),
                      "sample": f"one sample, one UNet evaluation, reference QuantModel (True, True), warm, k={k}"}))


if __name__ == "__main__":
    main()
