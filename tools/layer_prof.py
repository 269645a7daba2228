#!/usr/bin/env python3
"""Per-layer GPU time of qd_conv2d_i8 inside one SD UNet evaluation, from a rocprofv3 kernel trace
(no host-launch overhead in the numbers, unlike event timing around Python calls).

  step 1 (under rocprofv3 --kernel-trace):  python tools/layer_prof.py run  out.json [n]
  step 2:                                   python tools/layer_prof.py join out.json results.db
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(out_json, n):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
    import bench
    from qdiff import hip, synthetic
    dev = torch.device("cuda:0")
    qnn, _ = bench.build_quantised_unet("sd", dev)
    x, t, c = synthetic.synthetic_inputs("sd", 2 * n)
    args = [a.to(dev) for a in (x, t, c)]
    recs = []
    orig = hip.conv2d_i8

    def spy(call, acc_out=None):
        K = call.kh * call.kw * sum(s["clen"] for s in call.segs)
        split = bool(call.w_tiled and call.splitk is not False and acc_out is None and hip.splitk_ws_bytes(call) > 0)
        recs.append(dict(M=call.B * call.Ho * call.Wo, N=call.Cout, K=K, k=call.kh, nseg=len(call.segs), splitk=split,
                         geglu=call.epilogue == hip.EPI_GEGLU_I8, res=call.residual is not None))
        orig(call, acc_out)
    with torch.no_grad():
        for _ in range(5):               # warm: plans, allocator, clocks
            qnn.model(*args)
        torch.cuda.synchronize()
        hip.conv2d_i8 = spy
        torch.cuda._sleep(int(6e8))      # the host enqueues the whole evaluation behind a spin kernel: kernels then run
        qnn.model(*args)                 # back to back (the LAST evaluation in the trace is the one that is joined)
        torch.cuda.synchronize()
    hip.conv2d_i8 = orig
    json.dump(recs, open(out_json, "w"))


def join(in_json, db_path):
    recs = json.load(open(in_json))
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
        "on d.kernel_id = s.id order by d.start").fetchall()
    ig = [(nm, e - s) for nm, s, e in rows if "igemm" in nm or "splitk_finalize" in nm]
    need = sum(2 if r["splitk"] else 1 for r in recs)
    ig = ig[-need:]
    agg = collections.OrderedDict()
    i = 0
    for r in recs:
        dur = ig[i][1]
        assert "igemm" in ig[i][0], ig[i][0]
        i += 1
        if r["splitk"]:
            assert "splitk_finalize" in ig[i][0], (r, ig[i][0])
            dur += ig[i][1]
            i += 1
        key = (r["M"], r["N"], r["K"], r["k"], r["nseg"], r["splitk"], r["geglu"], r["res"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += dur / 1e3
    tot = sum(a[1] for a in agg.values())
    ops_tot = sum(2.0 * k[0] * k[1] * k[2] * a[0] for k, a in agg.items())
    print(f"{len(recs)} conv launches, {tot / 1e3:.2f} ms GPU time, {ops_tot / tot / 1e6:.1f} TOP/s aggregate")
    for (M, N, K, k, nseg, sk, gg, res), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        ops = 2.0 * M * N * K
        outb = (1 if gg else 4) * M * (N // 2 if gg else N) + (4 * M * N if res else 0)
        byts = M * K / (k * k) + N * K / 2 + outb
        tag = ("S" if sk else "-") + ("G" if gg else "-") + ("R" if res else "-") + str(nseg)
        print(f"M={M:6d} N={N:5d} K={K:5d} k{k} {tag} x{cnt:3d} {us / cnt:8.1f} us  {us / 1e3:6.2f} ms ({100 * us / tot:4.1f}%) "
              f"{ops / (us / cnt) / 1e6:7.1f} TOP/s {byts / (us / cnt) / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    else:
        join(sys.argv[2], sys.argv[3])
