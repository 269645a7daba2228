#!/usr/bin/env python3
"""Headline benchmark: denoising images/sec, SD-v1.4 UNet W4A8 (sm_abit=16, split shortcut),
512x512 (latent 4x64x64), PLMS 50 steps = 51 UNet evaluations per image at CFG batch 2n.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one sampler step of the hot path on one batch of synthetic input: a UNet evaluation on
the CFG-doubled batch (2n latents, context 77x768) through qdiff.QuantModel (HIP graph replay) +
guidance combine + PLMS multistep update.  Weights are random-init (key-derived, qdiff/synthetic.py),
quantisers are initialised from data exactly as the reference does on a first forward and converted to
AdaRound with alpha ~ U(-1,1); no network, no checkpoints.  All inputs are resident in HBM before the
timed region.  images/sec = N * n / (51 * seconds_per_step)   (51 evals per 50-step PLMS image batch).

Extra objects on the JSON line:
  roofline     — the dominant kernel class (igemm int8 MFMA contraction): algorithmic int ops of every
                 launch of one UNet evaluation / their HIP-event durations on the launch stream; fractions against
                 the nominal 5.0 POP/s and the 4.404 POP/s micro-benchmark ceiling; whole_step_* = all integer ops of
                 the evaluation / wall time of the sampler step.  by_class = EVERY library launch of the evaluation by class
                 (igemm / attention / producers / other, ms per evaluation, attention with its own fraction of the MFMA peak);
                 frac_of_box_ubench = the igemm class against what THIS box sustains (box.mfma_ubench_tops).
  box          — ~50 ms of issue loops in this run (qd_box_probe): dense int8 MFMA TOP/s, v_exp_f32 G wave-instructions/s and
                 the shader clock under each load: tells box speed from code speed when two lines are compared.
  gpu_denominators — the same UNet at the same batch on the same GPU with quantisation off (fp32 PyTorch-ROCm) and as
                 the reference's fp32 fake-quant simulation (rank 0, N=1 only).
  cpu_baseline — the oracle (CPU port of the reference fake-quant forward, oracle/unet_ref.py) timed on
                 this box's host cores: one warm-up + two timed UNet evaluations of one sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd"))
sys.path.insert(0, ROOT)

I8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA = 2x the 2.5 PF bf16 dense peak (MI355X_MICROARCH.md)
I8_MFMA_UBENCH_TOPS = 4404.0   # measured ceiling of v_mfma_i32_32x32x32_i8 (cdna_hip_programming.md §3; SURVEY.md §8d formula)
EVALS_PER_IMAGE_BATCH = 51     # PLMS S=50: 50 steps + 1 extra evaluation on the first step (plms.py:222-227)


def build_quantised_unet(kind, device, seed=0):
    import qdiff
    from qdiff import synthetic
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.arch import ddim_unet, ldm_unet
    from qdiff.utils import convert_adaround
    if kind == "sd":
        model = ldm_unet.UNetModel(**ldm_unet.sd_v1_config())
        model.split = True
        wq = dict(n_bits=4, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)
        sm_abit = 16
    elif kind == "ldm":
        model = ldm_unet.UNetModel(**ldm_unet.lsun_beds_config())
        model.split = True
        wq = dict(n_bits=4, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True, symmetric=True)
        sm_abit = 8
    elif kind == "churches":
        # LSUN-Churches LDM-8 (README.md:53-55): resampling ResBlocks with scale-shift norms; no split shortcut (the
        # reference cannot combine it with resblock_updown, tools/make_golden.py)
        model = ldm_unet.UNetModel(**ldm_unet.lsun_churches_config())
        model.split = False
        wq = dict(n_bits=4, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)     # README.md:55: no --a_sym
        sm_abit = 8
    else:
        model = ddim_unet.Model(ddim_unet.cifar10_config(split_shortcut=True))
        wq = dict(n_bits=8, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True, symmetric=True)
        sm_abit = 8
    synthetic.load_synthetic_weights(model, seed=seed)
    model = model.to(device).eval()
    qnn = qdiff.QuantModel(model, wq, aq, sm_abit=sm_abit).to(device).eval()
    qnn.set_quant_state(True, True)
    x, t, c = synthetic.synthetic_inputs(kind, 2, seed=seed)
    args = [a.to(device) for a in (x, t, c) if a is not None]
    with torch.no_grad():
        qnn(*args)                                   # data-dependent init of every quantiser (+ split)
    convert_adaround(qnn)
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    for m in qnn.modules():
        if isinstance(m, AdaRoundQuantizer):
            m.alpha.data.copy_(torch.rand(m.alpha.shape, generator=g, device=device) * 2 - 1)
    return qnn, dict(w_bits=wq["n_bits"], a_bits=8, a_sym=bool(aq.get("symmetric", False)), sm_abit=sm_abit)


def build_skeleton(kind, device, seed=777):
    """A QuantModel on the same architecture with UNRELATED weights and no calibration: what every rank but 0 holds
    before sampling.broadcast_packed_model hands it the packed state (its fp32 weights are never read)."""
    import qdiff
    from qdiff import synthetic
    from qdiff.arch import ddim_unet, ldm_unet
    if kind == "sd":
        model, sm_abit = ldm_unet.UNetModel(**ldm_unet.sd_v1_config()), 16
        wq, aq = dict(n_bits=4, channel_wise=True, scale_method="max"), dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)
    elif kind in ("ldm", "churches"):
        cfg = ldm_unet.lsun_beds_config() if kind == "ldm" else ldm_unet.lsun_churches_config()
        model, sm_abit = ldm_unet.UNetModel(**cfg), 8
        wq = dict(n_bits=4, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)
        if kind == "ldm":
            aq["symmetric"] = True
    else:
        model, sm_abit = ddim_unet.Model(ddim_unet.cifar10_config(split_shortcut=True)), 8
        wq = dict(n_bits=8, channel_wise=True, scale_method="max")
        aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True, symmetric=True)
    if kind in ("sd", "ldm"):
        model.split = True
    synthetic.load_synthetic_weights(model, seed=seed)
    qnn = qdiff.QuantModel(model.to(device).eval(), wq, aq, sm_abit=sm_abit).to(device).eval()
    return qnn, dict(w_bits=wq["n_bits"], a_bits=8, a_sym=bool(aq.get("symmetric", False)), sm_abit=sm_abit)


PRODUCER_ENTRIES = ("groupnorm_silu_quant", "layernorm_quant", "geglu_quant", "quantize_act", "quantize_heads", "temb_mlp")


def measure_classes(qnn, args):
    """HIP events around EVERY library launch of one eager UNet evaluation, on the launch stream, by class:
      igemm      qd_conv2d_i8 (the integer contractions; sub-classes by epilogue / K as before),
      attention  qd_attn_i8 (+ the key-term table it builds for a self-attention),
      producers  the row producers between them (GroupNorm / LayerNorm / GEGLU / activation / head-layout quantisers, the
                 timestep-embedding MLP group),
      other      whatever else the evaluation enqueues (torch glue: the sinusoid table, views that copy) = the evaluation's own
                 first-to-last interval minus the three classes.
    The stream is first parked behind a ~100 ms spin kernel so that the host finishes enqueueing the whole evaluation before
    the GPU starts it: the event intervals are then back-to-back GPU time of the kernels (what rocprofv3 --kernel-trace
    reports), not host launch latency of the eager Python path."""
    from qdiff import hip
    records = []                                       # (e0, e1, class, sub-class, integer ops)
    saved = {}

    def bracket(fn, classify):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            e0.record(st)
            r = fn(*a, **k)
            e1.record(st)
            records.append((e0, e1) + classify(*a, **k))
            return r
        return timed

    def conv_class(call, acc_out=None):
        k = call.kh * call.kw * sum(s["clen"] for s in call.segs)
        m = call.B * call.Ho * call.Wo
        epi = getattr(call, "epilogue", 0) or 0
        if epi == hip.EPI_GEGLU_I8:
            cls = "geglu_i8_out"
        elif epi in (hip.EPI_HEADS_I8, hip.EPI_HEADS_T_I8):
            cls = "heads_i8_out"
        elif len(call.segs) == 2:
            cls = "split_shortcut"
        elif call.splitk is not False and hip.splitk_ws_bytes(call) > 0:
            cls = "split_k(+finalise)"
        elif k >= 2880:
            cls = "long_k_f32_out"
        else:
            cls = "short_k_f32_out"
        return ("igemm", cls, 2.0 * m * k * call.Cout)

    def group_class(calls):
        ops, cls = 0.0, "heads_i8_out"
        for c in calls:
            _, cls, o = conv_class(c)
            ops += o
        return ("igemm", cls, ops)

    def attn_class(q, k, vt, vsum, BH, H, T, S, d, *a, **kw):
        return ("attention", f"T{T}_S{S}_d{d}", 4.0 * T * S * d * BH)

    # per-launch kernel time is measured with the launches one after another: the cross-attention K / V branch, which the
    # timed steps run CONCURRENTLY with the stem (quant_block.ContextKV, fork point "start"), is serialised into the main
    # stream here ("late") — overlapping event intervals of two streams would count the same wall time twice
    from qdiff import quant_block as qb
    fork0 = qb._CTX_FORK
    wrap = {"conv2d_i8": conv_class, "attn_i8": attn_class}
    if hasattr(hip, "conv2d_i8_group"):
        wrap["conv2d_i8_group"] = group_class
    for name in PRODUCER_ENTRIES:
        wrap[name] = (lambda nm: (lambda *a, **k: ("producers", nm, 0.0)))(name)
    for name, classify in wrap.items():
        saved[name] = getattr(hip, name)
        setattr(hip, name, bracket(saved[name], classify))
    qb._CTX_FORK = "late"
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        with torch.no_grad():
            torch.cuda._sleep(int(2.5e8))
            ev0.record(torch.cuda.current_stream())
            qnn.model(*args)
            ev1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(hip, name, fn)
        qb._CTX_FORK = fork0
    # an (e0, e1) pair with nothing in between still measures the event-record packets themselves:
    # calibrate that on the same parked stream and take it off every interval
    torch.cuda._sleep(int(2.5e7))
    st = torch.cuda.current_stream()
    empty = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        e1.record(st)
        empty.append((e0, e1))
    torch.cuda.synchronize()
    overhead = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    dur = lambda a, b: max(a.elapsed_time(b) - overhead, 0.0)
    by_class, sub = {}, {}
    for a, b, cls, sc, ops in records:
        for table, key in ((by_class, cls), (sub, (cls, sc))):
            c = table.setdefault(key, dict(launches=0, ms=0.0, GOP=0.0))
            c["launches"] += 1
            c["ms"] += dur(a, b)
            c["GOP"] += ops / 1e9
    # the evaluation's own interval carries one event pair per bracketed launch inside it: take those packets off as well
    eval_ms = max(ev0.elapsed_time(ev1) - overhead * (len(records) + 1), 0.0)
    for table in (by_class, sub):
        for c in table.values():
            if c["GOP"] > 0:
                c["TOPs"] = round(c["GOP"] / max(c["ms"], 1e-9), 1)               # GOP / ms == TOP/s
                c["frac"] = round(c["TOPs"] / I8_MFMA_PEAK_TOPS, 4)
            else:
                c.pop("GOP")
            c["ms"] = round(c["ms"], 3)
            if "GOP" in c:
                c["GOP"] = round(c["GOP"], 1)
    ig = by_class.get("igemm", dict(launches=0, ms=0.0, GOP=0.0))
    known = sum(c["ms"] for c in by_class.values())
    by_class["other"] = dict(ms=round(max(eval_ms - known, 0.0), 3),
                             what="the eager evaluation's first-to-last interval minus the classes: torch glue kernels (~0.07 ms per SD evaluation in "
                                  "the kernel traces) + the inter-launch gaps of EAGER launches with an event pair around each; a graph replay "
                                  "keeps only the ~1.5 us dependent-launch boundary per dispatch (see graph_replay_eval_ms)")
    return dict(launches=ig["launches"], total_ms=ig["ms"], ops=ig.get("GOP", 0.0) * 1e9, event_overhead_us=1000.0 * overhead,
                classes={k[1]: v for k, v in sub.items() if k[0] == "igemm"}, by_class=by_class, eval_ms=round(eval_ms, 3),
                attention_calls={k[1]: v for k, v in sub.items() if k[0] == "attention"},
                producer_entries={k[1]: v for k, v in sub.items() if k[0] == "producers"}, library_launches=len(records))


def measure_igemm(qnn, args):
    return measure_classes(qnn, args)


def box_calibration(dev):
    """~50 ms of issue loops (qd_box_probe): what THIS box sustains on the two instruction classes the evaluation is bound by —
    dense v_mfma_i32_32x32x32_i8 (two waves per SIMD) and v_exp_f32 (four) — and the shader clock under each load."""
    from qdiff import hip
    out = {}
    try:
        ms, ticks = hip.box_probe(dev, 0, 512, 60000)
        out["mfma_ubench_tops"] = round(512 * 4 * 60000 * 8 * 65536.0 / (ms * 1e-3) / 1e12, 1)
        # one v_mfma_i32_32x32x32_i8 occupies a SIMD's matrix pipe for 32 cycles (MI355X_MICROARCH.md): MFMAs per SIMD per second x 32
        out["sclk_mhz_observed"] = round(2 * 60000 * 8 * 32 / (ms * 1e3), 1)
        out["memtime_ticks_per_us_mfma"] = round(ticks / (ms * 1e3), 1)
        out["mfma_probe_ms"] = round(ms, 3)
        ms, ticks = hip.box_probe(dev, 1, 1024, 40000)
        out["exp_ginst_s"] = round(1024 * 4 * 40000 * 32 / (ms * 1e-3) / 1e9, 2)              # wave-instructions per second, chip-wide
        out["memtime_ticks_per_us_exp"] = round(ticks / (ms * 1e3), 1)
        out["exp_probe_ms"] = round(ms, 3)
        out["what"] = ("qd_box_probe in this run: dense v_mfma_i32_32x32x32_i8 at two waves per SIMD (TOP/s; sclk_mhz_observed = its "
                       "per-SIMD rate x 32 cycles), v_exp_f32 at four waves per SIMD (G wave-instructions/s); memtime_ticks = s_memtime "
                       "ticks of a wave / event time, reported as read")
    except Exception as exc:  # noqa: BLE001 - never lose the line over the calibration
        out["error"] = repr(exc)[:200]
    return out


def gpu_denominators(qnn, margs, k=2):
    """The two GPU-side denominators of SURVEY.md §8(d) / BASELINE.json north_star, same UNet, same batch, same
    GPU, same run: (a) the fp32 PyTorch-ROCm UNet — quantisation off, `org_weight` path of
    reference qdiff/quant_layer.py:273-276, library (MIOpen / rocBLAS) fp32 kernels; (b) the reference's fake-quant
    SIMULATION in fp32 torch ops (quant_layer.py:256-272, quant_block.py:190-221) on the GPU.  (b) runs this
    repo's restatement of that arithmetic, which caches the fake-quantised weights per quantiser state (the
    reference re-quantises them on every forward), so it flatters the reference."""
    from qdiff import engine

    def timed():
        with torch.no_grad():
            for _ in range(2):                         # warm: MIOpen find + kernel compilation, allocator
                qnn.model(*margs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                qnn.model(*margs)
            torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t0) / k
    graphs, qnn._graphs = qnn._graphs, None
    res = {"evals_timed": k}
    try:
        qnn.set_quant_state(False, False)
        res["fp32_unet_ms"] = round(timed(), 3)
        qnn.set_quant_state(True, True)
        engine.SIMULATE = True
        res["fake_quant_sim_ms"] = round(timed(), 3)
    finally:
        engine.SIMULATE = False
        qnn.set_quant_state(True, True)
        qnn._graphs = graphs
    return res


def cpu_baseline(qnn, qspec, kind, cfg, k=2):
    """The reference's CPU fake-quant path, timed on this box's host cores: oracle/unet_ref.py — the restatement of the
    reference forward, same ATen calls in the same order, pinned bit for bit to the real reference's outputs
    (tests/test_oracle_golden.py) — because /root/reference does not exist on the GPU box.  One warm-up evaluation,
    then k timed evaluations of ONE sample (batch 1; a CFG image needs two samples per evaluation)."""
    from oracle import unet_ref as U
    from qdiff import synthetic
    from qdiff.utils import export_cali_state_dict
    sd = {kk: v.cpu() for kk, v in export_cali_state_dict(qnn).items()}
    Q = U.QuantCkpt(sd, qspec["w_bits"], qspec["a_bits"], qspec["a_sym"], qspec["sm_abit"])
    x, t, c = synthetic.synthetic_inputs(kind, 1, seed=7)

    def one():
        with torch.no_grad():
            if kind == "cifar":
                U.cifar_forward(Q, cfg, x, t, split_shortcut=True)
            else:
                U.ldm_forward(Q, cfg, x, t, c, split=kind != "churches")
    # best-effort CPU number: batch 1 on every hardware thread of a 128-thread host is oversubscribed (round 2: 22.4 s on 128
    # threads against 13.3 s for the unmodified reference on 8 cores), so sweep the intra-op thread count on one evaluation
    # each, then time k evaluations at the best setting
    n0 = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64, n0) if t <= n0})
    torch.set_num_threads(cands[len(cands) // 2])
    one()                                              # warm-up (allocator, oneDNN primitives)
    sweep = {}
    for nthr in cands:
        torch.set_num_threads(nthr)
        t0 = time.time()
        one()
        sweep[nthr] = time.time() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.time()
    for _ in range(k):
        one()
    dt = (time.time() - t0) / k
    torch.set_num_threads(n0)
    return dt, best, {str(n): round(v, 2) for n, v in sweep.items()}


# engines of the first-stage decode: library convolutions in fp32 / under fp16 autocast (the reference scripts' mode,
# txt2img.py:231-236) / under bf16 autocast; this package's MFMA kernels with fp16 operands ("hip", the default) or bf16 operands
DECODE_LEGS = {"fp32": (None, None), "fp16_autocast": (torch.float16, None), "bf16_autocast": (torch.bfloat16, None),
               "hip": (None, "hip"), "hip_bf16": (None, "hip_bf16")}


def decode_leg(kind, n, leg, dev, k=3):
    """ONE engine of the first-stage decode (child process of first_stage_decode): latents -> uint8 images for the n images
    of a sampler batch (qdiff/arch/first_stage.py; reference ldm/models/autoencoder.py:330-333 + the scripts' clamp / scale).
    The distance from the fp32 library decode is taken on the first latent only (one image: no chunking involved)."""
    from qdiff import synthetic
    from qdiff.arch import first_stage as fs
    m, scale = {"sd": fs.sd_v1_first_stage, "ldm": fs.lsun_beds_first_stage, "churches": fs.lsun_churches_first_stage}[kind]()
    synthetic.load_synthetic_weights(m, seed=0)
    m = m.to(dev).eval()
    z = torch.randn({"sd": (n, 4, 64, 64), "ldm": (n, 3, 64, 64), "churches": (n, 4, 32, 32)}[kind], device=dev)
    dt, eng = DECODE_LEGS[leg]
    res = {"leg": leg, "images": n}
    if leg != "fp32":
        ref = fs.decode_first_stage(m, z[:1], scale).float()
        out = fs.decode_first_stage(m, z[:1], scale, autocast_dtype=dt, engine=eng).float()
        res["max_err_of_range_vs_fp32_library"] = float(f"{((out - ref).abs().max() / ref.abs().max()).item():.3e}")
        del ref, out
    img = fs.decode_first_stage(m, z, scale, autocast_dtype=dt, to_uint8=True, engine=eng)       # warm-up (packing, MIOpen find)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        img = fs.decode_first_stage(m, z, scale, autocast_dtype=dt, to_uint8=True, engine=eng)
    torch.cuda.synchronize()
    res["ms_per_image"] = round((time.perf_counter() - t0) * 1000.0 / k / n, 3)
    res["output"] = list(img.shape)
    return res


def _child(argv, cap_s, tag):
    """Run `bench.py <argv>` as a child process with a time cap; return its last JSON line or a record of how it ended.
    The name of what is about to run goes to stderr BEFORE it starts, so that a GPU fault names its leg."""
    import subprocess
    print(f"[bench] child: {tag} ...", file=sys.stderr, flush=True)
    t0 = time.time()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=cap_s, env=env)
    except subprocess.TimeoutExpired:
        print(f"[bench] child: {tag} hit its {cap_s} s cap", file=sys.stderr, flush=True)
        return {"error": f"time cap {cap_s} s"}
    dt = round(time.time() - t0, 1)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        print(f"[bench] child: {tag} FAILED rc={r.returncode}\n{r.stderr[-2000:]}", file=sys.stderr, flush=True)
        return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-300:], "wall_s": dt}
    print(f"[bench] child: {tag} ok in {dt} s", file=sys.stderr, flush=True)
    d = json.loads(lines[-1])
    d["wall_s"] = dt
    return d


def first_stage_decode(kind, n, legs=("fp32", "fp16_autocast", "hip", "hip_bf16"), cap_s=300):
    """First-stage decode of the n images of one sampler batch, every engine in its OWN child process (a fault in one — round
    3's library decode at exactly 2^31 bytes per activation — names itself and loses nothing else).  Not part of the denoising
    metric: reported so that the end-to-end cost of an image is visible next to the 51 / 200 UNet evaluations it follows."""
    return {leg: _child(["--decode-leg", leg, "--model", kind, "--images-per-gpu", str(n)], cap_s, f"first-stage decode, {leg}, {n} latents")
            for leg in legs}


def as_script_line(a, dev):
    """`other_configs.sd_as_script` (child mode): the UNet driven EXACTLY as the reference's unmodified sampler drives it
    (ldm/models/diffusion/plms.py:176-190 p_sample_plms / apply_model, scripts/txt2img.py:381-390,490): a fresh
    `torch.cat([x] * 2)`, `torch.cat([t] * 2)` and `torch.cat([uncond, c])` at EVERY step, no enable_hip_graphs(), no
    prepare_context() — what the model does about graphs and the constant conditioning it does by itself (default-on replay,
    captured on second sight; context recognised by value, prepared on first sight).  Timed: one whole 50-step PLMS run = 51
    evaluations of a NEW prompt (its first-sight preparation and the per-step value comparison are inside the timed region),
    after two short runs with other prompts (quantiser state; both context slots own a captured graph)."""
    from qdiff import sampling
    qnn, qspec = build_quantised_unet("sd", dev)
    n = a.images_per_gpu
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 50, eta=0.0)
    short = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 5, eta=0.0)
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *sh: torch.randn(sh, device=dev, generator=g)
    ckv = qnn.__dict__["_ctx_kv"]

    def p_sample_loop(tb, x, c, uc, scale=7.5):
        import numpy as np
        order = np.flip(tb.timesteps)
        old = []
        evals = 0
        for i, step in enumerate(order):
            index = len(order) - i - 1
            ts = torch.full((x.shape[0],), int(step), device=dev, dtype=torch.long)

            def eps(xx, tt):
                x_in, t_in, c_in = torch.cat([xx] * 2), torch.cat([tt] * 2), torch.cat([uc, c])       # plms.py:184-187
                e_u, e_c = qnn(x_in, t_in, c_in).chunk(2)
                return e_u + scale * (e_c - e_u)
            e = eps(x, ts)
            evals += 1
            if len(old) == 0:
                x_e, _ = tb.update(x, e, index)
                tn = torch.full((x.shape[0],), int(order[min(i + 1, len(order) - 1)]), device=dev, dtype=torch.long)
                e_prime = (e + eps(x_e, tn)) / 2
                evals += 1
            elif len(old) == 1:
                e_prime = (3 * e - old[-1]) / 2
            elif len(old) == 2:
                e_prime = (23 * e - 16 * old[-1] + 5 * old[-2]) / 12
            else:
                e_prime = (55 * e - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
            x, _ = tb.update(x, e_prime, index)
            old.append(e)
            if len(old) >= 4:
                old.pop(0)
        return x, evals

    with torch.no_grad():
        # two earlier prompts: quantiser state, and BOTH context slots get their captured graph (a graph reads the operand
        # buffers of the slot it was captured with) — the state of a process that has served two prompts before this one
        for _ in range(2):
            p_sample_loop(short, rnd(n, 4, 64, 64), rnd(n, 77, 768), rnd(n, 77, 768))
        torch.cuda.synchronize()
        # host cost of enqueueing one replay of the captured evaluation (no synchronisation in between)
        g0 = next(iter(qnn._graphs.values()))
        t0 = time.perf_counter()
        for _ in range(5):
            g0.graph.replay()
        replay_enqueue_ms = 1000.0 * (time.perf_counter() - t0) / 5
        torch.cuda.synchronize()
        runs0, vm0 = ckv.chain_runs, ckv.value_matches
        x, c, uc = rnd(n, 4, 64, 64), rnd(n, 77, 768), rnd(n, 77, 768)                          # prompt B
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, evals = p_sample_loop(table, x, c, uc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    ms = 1000.0 * dt / evals
    return {"metric": "denoising images/sec, SD-v1.4 W4A8, the unmodified sampler's call pattern", "value": round(n / dt, 4), "unit": "images/s",
            "ms_per_step": round(ms, 4), "evals_timed": evals, "images": n,
            "context_chain_runs_in_run": ckv.chain_runs - runs0, "contexts_recognised_by_value": ckv.value_matches - vm0,
            "graphs_captured": len(qnn._graphs or {}),
            "graph_replay_enqueue_ms": round(replay_enqueue_ms, 3),
            "explicit_calls": "none (no enable_hip_graphs, no prepare_context)",
            "config": {"workload": f"sd UNet eval batch {2 * n}, fresh torch.cat of x / t / context per step (plms.py:184-187), 50 PLMS steps = {evals} evaluations of a new prompt"}}


def extra_lines(a):
    """BASELINE.json configs 2 / 3 (CIFAR-10 W8A8, LDM-4 W4A8) and the first-stage decode on this package's own kernels, as
    extra keys of the headline line: short runs in child processes with a time cap, started AFTER the headline numbers are
    computed, so that nothing here can lose them."""
    out = {}
    common = ["--no-cpu-baseline", "--no-denominators", "--no-extras", "--steps", "10", "--warmup", "2"]
    for kind, n in (("cifar", 64), ("ldm", 64)):
        # BASELINE configs[2] (LDM-4) is quoted at `-n 10 -e 1.0` (README.md:47-49): batch 10 is that line's headline, batch 64 the
        # throughput point; both from one child (same model, same process)
        d = _child(["--model", kind, "--images-per-gpu", str(n)] + (["--extra-batch", "10"] if kind == "ldm" else []) + common, 240, f"{kind} line")
        if "error" in d:
            out[kind] = d
            continue
        line = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "wall_s") if k in d} | \
            {"workload": d["config"]["workload"], "eta": d["config"].get("eta"), "igemm_frac": d.get("roofline", {}).get("frac"),
             "launches_per_eval_igemm": d.get("roofline", {}).get("launches_per_eval"),
             "library_launches_per_eval": d.get("roofline", {}).get("library_launches_per_eval"),
             "by_class_ms": {k: v.get("ms") for k, v in (d.get("roofline", {}).get("by_class") or {}).items()}} | \
            {k: d["config"][k] for k in ("whole_step_graph_ms", "whole_step_graph_images_per_s") if k in d["config"]}
        xb = d["config"].get("extra_batch")
        if kind == "ldm" and isinstance(xb, dict) and "ms_per_step" in xb:
            out[kind] = {"metric": line["metric"], "value": xb["images_per_s"], "unit": "images/s", "ms_per_step": xb["ms_per_step"],
                         "whole_step_frac": xb.get("whole_step_frac"), "dtype": line.get("dtype"), "eta": line.get("eta"),
                         "workload": f"ldm UNet eval batch 10 (README.md:47-49 `-n 10 -e 1.0 -c 200`), per-step noise drawn full-batch-then-sliced inside the step",
                         "batch_64": line}
        else:
            out[kind] = line
    # the SD workload driven exactly like the reference's unmodified sampler drives it (no graph / context calls by the caller)
    out["sd_as_script"] = _child(["--as-script", "--images-per-gpu", str(a.images_per_gpu)], 300, "sd line, the unmodified sampler's call pattern")
    out["first_stage_decode_sd"] = first_stage_decode("sd", a.images_per_gpu, legs=("hip",), cap_s=240)["hip"]
    return out


def _committed(suffix):
    """Newest committed summary profiles/*<suffix> (measurements bench.py cannot take itself: counters, oracle-side envelopes)."""
    pd = os.path.join(ROOT, "profiles")
    for cand in sorted((f for f in os.listdir(pd) if f.endswith(suffix)), reverse=True):
        try:
            return dict(json.load(open(os.path.join(pd, cand))), source="profiles/" + cand)
        except (OSError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE when launched by torch.distributed.run, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images-per-gpu", type=int, default=8, help="n images per GPU (UNet batch 2n with CFG)")
    ap.add_argument("--model", default="sd", choices=["sd", "ldm", "cifar", "churches"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-denominators", action="store_true", help="skip the fp32 / fake-quant GPU denominators")
    ap.add_argument("--decode", action="store_true", help="also time the first-stage decode of the image batch, all three engines, "
                    "each in its own child process (sd / ldm / churches; extra field)")
    ap.add_argument("--decode-leg", default=None, choices=sorted(DECODE_LEGS), help="(child mode) time ONE decode engine and print its JSON line")
    ap.add_argument("--no-extras", action="store_true", help="skip the cifar / ldm / first-stage-decode child runs of the default SD line")
    ap.add_argument("--stream", default=None, choices=["fp32", "fp16"],
                    help="storage type of the inter-kernel activations (default: QDIFF_STREAM or fp32); fp16 = the precision the "
                         "reference scripts run at (--precision autocast); compute stays int8 MFMA / fp32 epilogues")
    ap.add_argument("--extra-batch", type=int, default=0, help="also time the same loop at this (smaller) number of images per GPU; extra key of config")
    ap.add_argument("--as-script", action="store_true",
                    help="(child mode) SD driven exactly as the reference's unmodified PLMS sampler drives it: fresh torch.cat of x / t / context "
                         "per step, no enable_hip_graphs / prepare_context calls; prints its own JSON line")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU): prove that `python bench.py --gpus N` becomes N ranks; used by tests")
    a = ap.parse_args()
    if a.gpus is None:                                  # `torchrun --nproc-per-node N bench.py` without --gpus: N ranks
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if a.decode_leg:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X")
        print(f"[bench] decode leg {a.decode_leg}: {a.images_per_gpu} {a.model} latents", file=sys.stderr, flush=True)
        with torch.no_grad():
            print(json.dumps(decode_leg(a.model, a.images_per_gpu, a.decode_leg, torch.device("cuda", 0))))
        return

    if a.as_script:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X")
        from qdiff import hip
        hip.load()
        print(json.dumps(as_script_line(a, torch.device("cuda", 0))))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks of one node, one process per GPU (the driver's own command line,
        # SURVEY.md §8e); the children see WORLD_SIZE and fall through to the code below
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {a.gpus}")
    if a.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(t.item())}))
        if world > 1:
            dist.destroy_process_group()
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the integer engine has no host path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    from qdiff import engine, hip, sampling
    hip.load()
    if a.stream:
        engine.set_stream_dtype(torch.float16 if a.stream == "fp16" else torch.float32)
    stream_name = "fp16" if engine.STREAM_DTYPE == torch.float16 else "fp32"

    kind, n = a.model, a.images_per_gpu
    if kind == "sd" and not a.no_extras and a.extra_batch == 0 and n > 5:
        a.extra_batch = 5                              # BASELINE configs[3]: `--n_samples 5` (README.md:59-61) = evaluation batch 10, next to the headline
    # rank 0 owns the calibrated model (synthetic here: data-dependent init + AdaRound conversion); every other rank
    # builds the bare architecture with unrelated weights and receives the packed state — the ONLY collective of the run
    qnn, qspec = build_quantised_unet(kind, dev) if rank == 0 else build_skeleton(kind, dev)
    nbytes = sampling.broadcast_packed_model(qnn, src=0)

    from qdiff.arch import ldm_unet
    if kind == "sd":
        shape, ctx_shape, guide, evals = (4, 64, 64), (77, 768), 7.5, EVALS_PER_IMAGE_BATCH
        betas = sampling.ldm_betas(0.00085, 0.0120)
        ocfg = ldm_unet.sd_v1_config()
    elif kind == "ldm":
        shape, ctx_shape, guide, evals = (3, 64, 64), None, 1.0, 200
        betas = sampling.ldm_betas(0.0015, 0.0195)
        ocfg = ldm_unet.lsun_beds_config()
    elif kind == "churches":
        shape, ctx_shape, guide, evals = (4, 32, 32), None, 1.0, 400          # README.md:53-55: -c 400 -e 0.0
        betas = sampling.ldm_betas(0.0015, 0.0155)                            # models/ldm/lsun_churches256/config.yaml:5-6
        ocfg = ldm_unet.lsun_churches_config()
    else:
        shape, ctx_shape, guide, evals = (3, 32, 32), None, 1.0, 100
        betas = sampling.ddpm_betas()
        ocfg = dict(ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16], resolution=32)
    # BASELINE configs as the reference states them: LDM-4 LSUN-beds runs DDIM with `-e 1.0` (README.md:47-49) — sigma_t > 0, i.e.
    # a fresh noise tensor per step (ddim.py:216), drawn full-batch-then-sliced on every rank INSIDE the timed
    # step; SD (PLMS, plms.py:216), CIFAR (denoising.py:29, eta 0) and churches (`-e 0.0`, README.md:53-55) have sigma = 0
    eta = 1.0 if kind == "ldm" else 0.0
    table = sampling.StepTable(betas, {"sd": 50, "ldm": 200, "churches": 400}.get(kind, 100), eta=eta)
    gb = n * world
    x = sampling.sharded_noise((gb,) + shape, seed=0, world_size=world, rank=rank, device=dev)
    cond = uncond = None
    if ctx_shape:
        cond = sampling.sharded_noise((gb,) + ctx_shape, 1, world, rank, dev)
        uncond = sampling.sharded_noise((gb,) + ctx_shape, 2, world, rank, dev)
    if not a.no_graph:
        qnn.enable_hip_graphs(True)

    def unet(xx, tt, cc=None):
        return qnn(xx, tt, cc) if cc is not None else qnn(xx, tt.float() if kind == "cifar" else tt)

    state = dict(x=x, old=[], i=0, cond=cond, uncond=uncond, ctx2=None, noise=None)
    if eta > 0.0:
        state["noise"] = sampling.ShardedStepNoise(gb, shape, 4321, world, rank, dev)
    # The conditioning of a sampling run is constant (plms.py:184-187 rebuilds the same torch.cat([uncond, c]) at every step):
    # build it once and let the model compute its cross-attention K / V^T operands ONCE per run (QuantModel.prepare_context,
    # QDIFF_CTX_PIN=0 restores the per-evaluation computation).  The one-off cost is measured here and charged to every step
    # as prepare_ms / evals (one preparation per image batch of `evals` evaluations).
    ctx2, prepare_ms, ctx_prepared = None, 0.0, False
    if ctx_shape:
        ctx2 = state["ctx2"] = torch.cat([uncond, cond])
        with torch.no_grad():
            qnn(torch.cat([x] * 2), torch.full((2 * x.shape[0],), 500, device=dev, dtype=torch.long), ctx2)   # plans, packs, graph of the unprepared shape
            ctx_prepared = bool(qnn.prepare_context(ctx2))
            if ctx_prepared:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    qnn.prepare_context(ctx2)
                torch.cuda.synchronize()
                prepare_ms = 1000.0 * (time.perf_counter() - t0) / 3

    def one_step():
        i = state["i"] % len(table)
        index = len(table) - i - 1
        t = torch.full((state["x"].shape[0],), int(table.timesteps[index]), device=dev, dtype=torch.long)
        e = sampling.guided_eps(unet, state["x"], t, state["cond"], state["uncond"], guide, state["ctx2"])
        old = state["old"]
        if kind == "sd" and len(old) >= 3:
            e_prime = (55 * e - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        else:
            e_prime = e
        noise = state["noise"]() if state["noise"] is not None else None      # eta > 0: the whole batch's draw, this rank's slice
        state["x"], _ = table.update(state["x"], e_prime, index, noise)
        old.append(e)
        if len(old) >= 4:
            old.pop(0)
        state["i"] += 1

    with torch.no_grad():
        for _ in range(max(a.warmup, 4 if kind == "sd" else 1)):
            one_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(state["x"]).all(), "sampler state diverged"

    ms_per_step = 1000.0 * elapsed / a.steps + prepare_ms / evals
    images_per_s = gb / (evals * ms_per_step / 1000.0)

    def retime(warm=4):
        """The timed loop again (same model, whatever stream type / batch is now in force): ms per step, rank-local."""
        state.update(old=[], i=0)
        with torch.no_grad():
            for _ in range(warm):
                one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                one_step()
            torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t1) / a.steps
    out = {
        "metric": "denoising images/sec (whole node), SD-v1.4 W4A8 512x512 50-step PLMS" if kind == "sd" else f"denoising images/sec ({kind})",
        "value": round(images_per_s, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"int8xint4->int32 ({stream_name} residual stream)", "data": "synthetic",
        "config": {"workload": f"{kind} UNet eval batch {2 * n if guide != 1.0 else n} per GPU, {evals} evals per image batch, "
                               f"W{qspec['w_bits']}A8 sm_abit={qspec['sm_abit']}{'' if kind == 'churches' else ' split'}, hip-graph={'off' if a.no_graph else 'on'}"
                               + (f", cross-attention K/V of the run's context prepared once per image batch ({prepare_ms:.2f} ms, charged as /{evals} per step)"
                                  if ctx_prepared else (", cross-attention K/V recomputed per evaluation" if ctx_shape else "")),
                   "images_per_gpu": n, "global_batch": gb, "single_unet_step_ms": round(ms_per_step, 4), "eta": eta,
                   "context_prepare_ms": round(prepare_ms, 3), "context_prepared": ctx_prepared,
                   "parallelism": f"batch-sharded x{world}, quant-state broadcast {nbytes} B"},
    }
    if kind == "cifar" and rank == 0:
        # BASELINE configs[1] as a whole-sampler-step graph (generalized_steps captured like DevicePLMS): the pixel-space UNet
        # is launch-latency-bound (~0.04 of the MFMA peak), so the per-step host work a graph of the evaluation alone leaves
        # is visible; extra key, the headline above stays the evaluation-graph number
        seq = sampling.quad_sequence(1000, 100)
        bt = torch.from_numpy(betas).float().to(dev) if not torch.is_tensor(betas) else betas.float().to(dev)
        smp = sampling.DeviceGeneralizedSteps(qnn, x, seq, bt, use_graph=True)
        with torch.no_grad():
            for _ in range(3):
                smp.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                smp.step()
            torch.cuda.synchronize()
        ws_ms = 1000.0 * (time.perf_counter() - t0) / a.steps
        out["config"]["whole_step_graph_ms"] = round(ws_ms, 4)
        out["config"]["whole_step_graph_images_per_s"] = round(gb / (evals * ws_ms / 1000.0), 3)
    if rank == 0:
        # ---- roofline of the dominant kernel class (live HIP events, eager launches) ----------------
        xb = state["x"]
        tb = torch.full((xb.shape[0],), 500, device=dev, dtype=torch.long)
        if guide != 1.0:
            margs = [torch.cat([xb] * 2), torch.cat([tb] * 2), ctx2]
        else:
            margs = [xb, tb.float() if kind == "cifar" else tb]
        box = box_calibration(dev)                     # what this box sustains, taken right after the timed steps
        # the captured evaluation alone (no guidance / sampler update around it): ms per replay, back to back
        replay_ms = None
        if qnn._graphs:
            g0 = list(qnn._graphs.values())[-1]
            for _ in range(2):
                g0.graph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                g0.graph.replay()
            torch.cuda.synchronize()
            replay_ms = round(1000.0 * (time.perf_counter() - t1) / 10, 4)
        measure_classes(qnn, margs)                    # warm (eager path, caches)
        r = measure_classes(qnn, margs)
        ach = r["ops"] / (r["total_ms"] * 1e-3) / 1e12
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs of this command,
        # tools/final_measure.sh); bench.py cannot collect counters itself, so it reports the committed summary of the
        # same workload and says which file it read
        traffic, traffic_src = None, None
        for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"_{kind}_igemm_hbm_traffic.json")),
                           reverse=True):
            tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            traffic = tj.get("hbm_bytes_per_call_corrected")
            traffic_src = "profiles/" + cand + (f" (measured at commit {tj['commit']})" if tj.get("commit") else " (commit of the measurement not recorded)")
            break
        # whole-step view: every integer op of the evaluation (contractions + attention, SURVEY.md §8d per-sample figures)
        # against the wall clock of the timed sampler step
        per_sample_gop = {"sd": 803.0, "ldm": 202.0, "cifar": 12.5, "churches": 41.8}[kind]
        step_top = per_sample_gop * 1e9 * (xb.shape[0] * (2 if guide != 1.0 else 1)) / (ms_per_step * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                           "frac": round(ach / I8_MFMA_PEAK_TOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "frac_of_ubench_ceiling_4404": round(ach / I8_MFMA_UBENCH_TOPS, 4),
                           "whole_step_achieved": round(step_top, 2), "whole_step_frac": round(step_top / I8_MFMA_PEAK_TOPS, 4),
                           "whole_step_frac_of_4404": round(step_top / I8_MFMA_UBENCH_TOPS, 4),
                           "kernel": "every qd_conv2d_i8 launch of one evaluation: igemm_kernel<MT,NT,WM,WN,..> (+ splitk_finalize_kernel)",
                           "launches_per_eval": r["launches"],
                           "avg_launch_us": round(1000.0 * r["total_ms"] / r["launches"], 2),
                           "igemm_ms_per_eval": round(r["total_ms"], 3), "algorithmic_GOP_per_eval": round(r["ops"] / 1e9, 1),
                           "event_pair_overhead_us": round(r["event_overhead_us"], 2),
                           "isolation": "per-launch times taken with the context K/V branch serialised into the launch stream "
                                        "(QDIFF_CTX_FORK=late); the timed steps run it concurrently with the stem (start)",
                           "by_launch_class": r["classes"],
                           # every library launch of the evaluation by class (ms per evaluation, eager launches back to back on a
                           # parked stream); their sum + other = eval_ms_eager, to be read against ms_per_step (graph replay +
                           # guidance / sampler update glue)
                           "by_class": r["by_class"], "eval_ms_eager": r["eval_ms"], "library_launches_per_eval": r["library_launches"],
                           "graph_replay_eval_ms": replay_ms,
                           "sampler_glue_ms": (round(ms_per_step - prepare_ms / evals - replay_ms, 4) if replay_ms is not None else None),
                           "attention_calls": r["attention_calls"], "producer_entries": r["producer_entries"],
                           "frac_of_box_ubench": (round(ach / box["mfma_ubench_tops"], 4) if box.get("mfma_ubench_tops") else None),
                           "whole_step_frac_of_box_ubench": (round(step_top / box["mfma_ubench_tops"], 4) if box.get("mfma_ubench_tops") else None)}
        out["box"] = box
        if world == 1 and not a.no_denominators:
            # same UNet, same batch, same GPU, same run: the denominators of north_star's ">= 4x the reference fp32
            # PyTorch-ROCm UNet" target
            den = gpu_denominators(qnn, margs)
            den["speedup_vs_fp32_unet"] = round(den["fp32_unet_ms"] / ms_per_step, 2)
            den["speedup_vs_fake_quant_sim"] = round(den["fake_quant_sim_ms"] / ms_per_step, 2)
            den["note"] = ("ms per UNet evaluation at the bench batch; fp32_unet = quantisation off (org_weight path, MIOpen/rocBLAS "
                           "fp32); fake_quant_sim = the reference's fp32 simulation arithmetic on the GPU (fake-quantised weights cached)")
            out["gpu_denominators"] = den
        if world == 1 and not a.no_cpu_baseline:
            dt, cpu_threads, cpu_sweep = cpu_baseline(qnn, qspec, kind, ocfg)
            # dt = one sample through one UNet evaluation; an image needs `evals` evaluations of 2 samples (CFG) or 1
            per_image = evals * (2 if guide != 1.0 else 1) * dt
            out["cpu_baseline"] = {"value": round(1.0 / per_image, 6), "unit": "images/s", "cores": cpu_threads,
                                   "kind": "port", "what": "oracle/unet_ref.py — the CPU PORT of the reference's fake-quant forward (pinned bit for bit to the "
                                                           "reference's outputs; /root/reference does not exist on the GPU box), at BATCH 1, at the best "
                                                           "thread count of a sweep; the unmodified reference on 8 build-container cores: profiles/r02_cpu_reference_time.json",
                                   "sample": f"UNet evaluations of one sample on the host cores: 1 warm-up, one evaluation per "
                                                             f"thread count {cpu_sweep} (seconds), then 2 timed at the best ({cpu_threads} of "
                                                             f"{torch.get_num_threads()} threads): {dt:.1f} s each; extrapolated to {evals} evaluations x "
                                                             f"{2 if guide != 1.0 else 1} samples per image"}
        if a.decode and kind in ("sd", "ldm", "churches"):
            out["first_stage_decode"] = first_stage_decode(kind, n)
        if a.extra_batch and 0 < a.extra_batch < n and world == 1:
            # the same loop at another batch — BASELINE configs[2] is quoted at `-n 10` (README.md:47-49), configs[3] at
            # `--n_samples 5` = evaluation batch 10 (README.md:59-61): extra key, same process, same model
            keep = {k: state[k] for k in ("x", "cond", "uncond", "ctx2", "noise")}
            try:
                nb = a.extra_batch
                state["x"] = keep["x"][:nb].clone()
                prep_b = 0.0
                if ctx_shape:
                    state["cond"], state["uncond"] = keep["cond"][:nb].clone(), keep["uncond"][:nb].clone()
                    state["ctx2"] = torch.cat([state["uncond"], state["cond"]])
                    with torch.no_grad():
                        qnn(torch.cat([state["x"]] * 2), torch.full((2 * nb,), 500, device=dev, dtype=torch.long), state["ctx2"])
                        if qnn.prepare_context(state["ctx2"]):
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            for _ in range(3):
                                qnn.prepare_context(state["ctx2"])
                            torch.cuda.synchronize()
                            prep_b = 1000.0 * (time.perf_counter() - t1) / 3
                if eta > 0.0:
                    state["noise"] = sampling.ShardedStepNoise(nb, shape, 4321, 1, 0, dev)
                ms_b = retime() + prep_b / evals
                nsamp = nb * (2 if guide != 1.0 else 1)
                out["config"]["extra_batch"] = {"images_per_gpu": nb, "eval_batch": nsamp, "ms_per_step": round(ms_b, 4),
                                                "images_per_s": round(nb / (evals * ms_b / 1000.0), 4),
                                                "whole_step_frac": round(per_sample_gop * 1e9 * nsamp / (ms_b * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, 4)}
            except Exception as exc:  # noqa: BLE001 - never lose the line over an extra
                out["config"]["extra_batch"] = {"error": repr(exc)[:200]}
            finally:
                state.update(keep)
                if ctx_shape:
                    with torch.no_grad():
                        qnn.prepare_context(state["ctx2"])
        if kind == "sd" and world == 1 and not a.no_extras:
            other = {}
            if engine.STREAM_DTYPE == torch.float32:
                # the same workload on the opt-in fp16 activation stream (the reference scripts' own precision, txt2img.py:231-236):
                # same process, same packed model; the context is prepared again (its operands depend on the stream's rows)
                try:
                    engine.set_stream_dtype(torch.float16)
                    with torch.no_grad():
                        qnn(torch.cat([state["x"]] * 2), torch.full((2 * state["x"].shape[0],), 500, device=dev, dtype=torch.long), ctx2)
                        qnn.prepare_context(ctx2)
                    ms16 = retime() + prepare_ms / evals
                    measure_igemm(qnn, margs)
                    r16 = measure_igemm(qnn, margs)
                    ach16 = r16["ops"] / (r16["total_ms"] * 1e-3) / 1e12
                    other["sd_fp16_stream"] = {"ms_per_step": round(ms16, 4), "value": round(gb / (evals * ms16 / 1000.0), 4), "unit": "images/s",
                                               "dtype": "int8xint4->int32 (fp16 residual stream)", "vs_fp32_stream_ms": round(ms16 - ms_per_step, 4),
                                               "igemm_ms_per_eval": round(r16["total_ms"], 3), "igemm_frac": round(ach16 / I8_MFMA_PEAK_TOPS, 4),
                                               "by_launch_class": r16["classes"], "envelope": _committed("_fp16_envelope.json")}
                except Exception as exc:  # noqa: BLE001
                    other["sd_fp16_stream"] = {"error": repr(exc)[:300]}
                finally:
                    engine.set_stream_dtype(torch.float32)
            other.update(extra_lines(a))
            out["other_configs"] = other
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
