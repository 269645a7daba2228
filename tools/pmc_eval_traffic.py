#!/usr/bin/env python3
"""HBM traffic of ONE steady-state UNet evaluation from rocprofv3 PMC passes of `tools/eval_breakdown.py run` (eager
launches; the measured evaluations sit between two spin-kernel markers, so initialisation / warm-up evaluations at other
batch sizes are excluded).

    python tools/pmc_eval_traffic.py <dir with the FETCH_SIZE and WRITE_SIZE pass outputs> <evals> <out.json> [kind]

Per kernel class: FETCH_SIZE / WRITE_SIZE sums (KiB as rocprofv3 reports them), raw bytes, and the gfx950-corrected bytes
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of 16-B/lane streaming reads on gfx950 ->
bytes = (2*FETCH + WRITE) * 1024; an upper bound where reads are narrower)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from eval_breakdown import klass  # noqa: E402


def load(path, counter):
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def between_markers(rows):
    spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[1]]
    if len(spins) < 2:
        raise SystemExit("markers not found")
    return rows[spins[-2] + 1:spins[-1]]


def main(path, evals, out_path, kind="sd"):
    res = {}
    calls = 0
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        seg = between_markers(load(path, c))
        per = collections.Counter()
        n = collections.Counter()
        for _, nm, v in seg:
            per[klass(nm)] += v
            n[klass(nm)] += 1
        res[c] = {k: per[k] / evals for k in per}
        res[c + "_dispatches_per_eval"] = {k: n[k] / evals for k in n}
        # contraction launches: igemm_kernel / igemm_k2_kernel / igemm_heads_group_kernel (a grouped q / k / v launch counts once,
        # as bench.py's per-launch events count it); the split-K finalise passes belong to their partial launches
        calls = sum(1 for _, nm, _ in seg if "igemm_" in nm and "_kernel" in nm) / evals
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- "
                      f"python tools/eval_breakdown.py run {kind} 8 {evals}   (steady-state evaluations between spin markers only)",
           "evaluations": evals, "classes": {}}
    for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
        f, w = res["FETCH_SIZE"].get(k, 0.0), res["WRITE_SIZE"].get(k, 0.0)
        out["classes"][k] = {"FETCH_SIZE_KiB_per_eval": round(f, 1), "WRITE_SIZE_KiB_per_eval": round(w, 1),
                             "bytes_per_eval_raw": round((f + w) * 1024), "bytes_per_eval_corrected": round((2 * f + w) * 1024),
                             "dispatches_per_eval": res["FETCH_SIZE_dispatches_per_eval"].get(k)}
    ig = out["classes"].get("igemm")
    if ig and calls:
        out["qd_conv2d_i8_calls_per_eval"] = calls
        out["hbm_bytes_per_call_raw"] = round(ig["bytes_per_eval_raw"] / calls)
        out["hbm_bytes_per_call_corrected"] = round(ig["bytes_per_eval_corrected"] / calls)
    out["correction"] = ("MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane streaming read -> "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; an upper bound for kernels with narrower reads")
    out["commit"] = os.environ.get("QD_COMMIT")          # the GPU box has no .git: the caller passes `git rev-parse --short HEAD`
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "sd")
