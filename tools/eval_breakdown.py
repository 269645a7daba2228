#!/usr/bin/env python3
"""Steady-state per-kernel breakdown of ONE UNet evaluation (eager launches, GPU kept busy so kernels run back to back).

  step 1 (under rocprofv3 --kernel-trace):  python tools/eval_breakdown.py run [sd|ldm|cifar] [n images] [evals] [graph] [pin]
  step 2:                                   python tools/eval_breakdown.py join results.db [evals]

The measured evaluations are bracketed by two spin kernels (torch.cuda._sleep), which `join` looks for in the trace;
everything between them is attributed to `evals` evaluations.  Kernel classes: igemm (+ split-K finalise), attention,
producers (GroupNorm / LayerNorm / row quantisers), torch glue (everything else)."""
import collections
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(kind, n, evals, graph=False, pin=False):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
    import bench
    from qdiff import synthetic
    dev = torch.device("cuda:0")
    qnn, _ = bench.build_quantised_unet(kind, dev)
    x, t, c = synthetic.synthetic_inputs(kind, 2 * n if kind == "sd" else n)
    args = [a.to(dev) for a in (x, t, c) if a is not None]
    if pin and len(args) > 2:
        with torch.no_grad():
            qnn(*args)                       # plans, packs
            assert qnn.prepare_context(args[2])      # what bench.py times since round 4: the run's context prepared once
    if graph:
        qnn.enable_hip_graphs(True)          # what bench.py times: the captured evaluation, replayed
    fwd = (lambda: qnn(*args)) if graph else (lambda: qnn.model(*args))
    with torch.no_grad():
        for _ in range(5):
            fwd()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(6e8))          # marker + lets the host run ahead of the GPU
        for _ in range(evals):
            fwd()
        torch.cuda._sleep(int(1e7))          # closing marker
        torch.cuda.synchronize()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:120]


def klass(name):
    if "igemm" in name or "splitk_finalize" in name:
        return "igemm"
    if "attn_" in name or "bmm_" in name:
        return "attention"
    if any(s in name for s in ("gn_", "ln_quant", "quant_rows", "quant_strided", "quant_heads", "geglu_quant", "temb_", "qparams")):
        return "producers"
    return "torch glue"


def join(db_path, evals):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
        "on d.kernel_id = s.id order by d.start").fetchall()
    spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
    assert len(spins) >= 2, "markers not found"
    a, b = spins[-2], spins[-1]
    seg = rows[a + 1:b]
    wall = (rows[b][1] - rows[a][2]) / 1e6 / evals
    per = collections.OrderedDict()
    cls = collections.Counter()
    for nm, s, e in seg:
        k = short(nm)
        v = per.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += (e - s) / 1e6
        cls[klass(nm)] += (e - s) / 1e6
    tot = sum(v[1] for v in per.values())
    print(f"{len(seg) / evals:.0f} dispatches per evaluation, kernel time {tot / evals:.3f} ms per evaluation, "
          f"GPU wall between markers {wall:.3f} ms per evaluation")
    for k, v in cls.most_common():
        print(f"  class {k:12s} {v / evals:8.3f} ms  ({100 * v / tot:4.1f} %)")
    print("| ms/eval | calls/eval | avg us | kernel |\n|---|---|---|---|")
    for k, (n, ms) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"| {ms / evals:7.3f} | {n / evals:6.1f} | {1000 * ms / n:8.2f} | {k} |")


def timeline(db_path, evals, out_path):
    """Per-dispatch timeline of the LAST evaluation between the markers: start, duration, gap to the previous kernel's
    end, grid, kernel.  (The table's grid columns are discovered: rocpd schemas differ between ROCm releases.)"""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    gcols = [c for c in cols if "grid" in c.lower()] + [c for c in cols if "workgroup" in c.lower()]
    sel = ", ".join("d." + c for c in gcols)
    rows = db.execute(
        f"select s.kernel_name, d.start, d.end{', ' + sel if sel else ''} from rocpd_kernel_dispatch d join "
        "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    spins = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
    a, b = spins[-2], spins[-1]
    seg = rows[a + 1:b]
    per = len(seg) // evals
    last = seg[len(seg) - per:]
    t0 = last[0][1]
    prev_end = None
    with open(out_path, "w") as f:
        f.write("# idx\tstart_us\tdur_us\tgap_us\t" + "/".join(gcols) + "\tkernel\n")
        for i, r in enumerate(last):
            gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
            prev_end = max(prev_end or 0, r[2])
            f.write(f"{i}\t{(r[1] - t0) / 1e3:.2f}\t{(r[2] - r[1]) / 1e3:.2f}\t{gap:.2f}\t{'/'.join(str(x) for x in r[3:])}\t{short(r[0])[:100]}\n")
    print(f"{per} dispatches -> {out_path}")


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "sd", int(sys.argv[3]) if len(sys.argv) > 3 else 8,
            int(sys.argv[4]) if len(sys.argv) > 4 else 3, graph="graph" in sys.argv, pin="pin" in sys.argv)
    else:
        join(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
