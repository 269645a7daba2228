"""Whole-UNet parity of the HIP engine (through qdiff.QuantModel, the drop-in boundary) against the
real reference's outputs (tests/golden/model_*.pt) and against the CPU oracle.

Tolerance (tier T2, SURVEY.md §7 "Parity definition"): the engine accumulates exact
integers where the reference accumulates fp32 products, so activations entering the NEXT quantiser
differ by ~1e-7 relative and occasionally flip a round() tie; a quantised network amplifies such
flips.  The reference is subject to the same effect: evaluating the SAME fake-quant network in fp64
(oracle tier T2x, QuantCkpt64) moves its output by 2.5e-2 .. 7e-2 of range.  Stated bound for
(weight+act) quantised UNets:
    max|engine - ref_fp64| <= 1.25 * max|ref_fp32 - ref_fp64| + 1e-3 * range     (the engine is at least as close to
                                                 exact arithmetic as the reference's own fp32 run; measured 0.8x .. 1.0x)
    max|engine - ref_fp32| <= 1.5  * max|ref_fp32 - ref_fp64| + 1e-3 * range     (two realisations of a chaotic map can
                                                 be up to the sum of their distances apart; measured 0.9x .. 1.3x)
    cosine(engine, ref_fp32) >= 0.995, cosine(engine, ref_fp64) >= cosine(ref_fp32, ref_fp64) - 1e-3
(round 1 allowed 2x; the discriminating per-block bounds are in tests/test_block_parity.py);
weights-only and fp states run plain fp32 library convolutions: max|diff| <= 1e-3 * max|ref|.
"""
import os
import sys
import tempfile

import pytest
import torch

from golden_util import build_ckpt, build_engine_model, fixture_inputs, load_fixture, quant_params

pytestmark = pytest.mark.gpu

TINY = ["cifar_tiny", "ldm_tiny", "sd_tiny"]
FULL = ["cifar_full", "ldm_full", "sd_full"]


def _resume(fx, dev):
    import qdiff
    from qdiff.utils import resume_cali_model
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    model = build_engine_model(spec).to(dev)
    qnn = qdiff.QuantModel(model, wq, aq, sm_abit=spec["sm_abit"]).to(dev).eval()
    cal = tuple(a for a in fixture_inputs(fx, "cal") if a is not None)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt.pth")
        torch.save(build_ckpt(fx), path)
        resume_cali_model(qnn, path, cal, quant_act=True, cond=spec["ctx"] is not None)
    return qnn


def _run(qnn, fx, dev):
    x, t, c = fixture_inputs(fx, "test")
    with torch.no_grad():
        y = qnn(x.to(dev), t.to(dev), c.to(dev)) if c is not None else qnn(x.to(dev), t.to(dev))
    torch.cuda.synchronize()
    return y.float().cpu()


def _oracle64(fx):
    """fp64 evaluation of the same fake-quant network (oracle tier T2x), or a stored copy."""
    if "out_wa_oracle64" in fx:
        return fx["out_wa_oracle64"].double()
    if fx["name"].endswith("_full") and fx["name"] != "cifar_full":
        return None
    from golden_util import oracle_cfg
    from oracle import unet_ref as U
    spec = fx["spec"]
    Q = U.QuantCkpt64(build_ckpt(fx), spec["w_bits"], spec["a_bits"], spec["a_sym"], spec["sm_abit"])
    x, t, c = fixture_inputs(fx, "test")
    with torch.no_grad():
        if spec["family"] == "cifar":
            return U.cifar_forward(Q, oracle_cfg(spec), x.double(), t, split_shortcut=spec["split"])
        return U.ldm_forward(Q, oracle_cfg(spec), x.double(), t, None if c is None else c.double(), split=spec["split"])


def _metrics(y, ref):
    d = (y - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(y.flatten(), ref.flatten(), dim=0).item()
    return d, cos, ref.abs().max().item()


# ldm_updown_tiny / churches_full: resblock_updown + use_scale_shift_norm (LSUN-Churches LDM-8, models/ldm/lsun_churches256);
# collected late: added after the round's last GPU run
LATE = ["ldm_updown_tiny", "churches_full"]


@pytest.mark.parametrize("name", TINY + FULL + LATE)
def test_quantised_unet_matches_reference(cuda, name):
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    # every QuantModule must actually be on the integer path
    import qdiff
    mods = [m for m in qnn.modules() if isinstance(m, qdiff.QuantModule)]
    assert len(mods) == fx["n_quant_modules"]
    assert all(m.int_ready() for m in mods)
    y = _run(qnn, fx, cuda)
    assert all(m._plan is not None for m in mods), "a QuantModule did not run the integer kernel"
    d, cos, mx = _metrics(y, fx["out_wa"])
    print(f"\n[{name}] W+A vs reference fp32: max|diff|={d:.3e} ({d / mx:.2e} of range), cosine={cos:.7f}")
    y64 = _oracle64(fx)
    assert y64 is not None, "fixture lacks the fp64-oracle envelope (tools/add_oracle64.py)"
    d64, cos64, _ = _metrics(y.double(), y64)
    dself, cosself, _ = _metrics(fx["out_wa"].double(), y64)
    print(f"[{name}] engine vs fp64 oracle: {d64 / mx:.2e} (cos {cos64:.7f}) | reference fp32 vs fp64 oracle: "
          f"{dself / mx:.2e} (cos {cosself:.7f})")
    # bound relative to the reference's own rounding-noise envelope (module docstring)
    assert d64 <= 1.25 * dself + 1e-3 * mx, f"{name}: engine is {d64 / mx:.3e} of range from the fp64 evaluation, the reference {dself / mx:.3e}"
    assert d <= 1.5 * dself + 1e-3 * mx, f"{name}: {d / mx:.3e} of range vs envelope {dself / mx:.3e}"
    assert cos >= 0.995 and cos64 >= cosself - 1e-3
    if name in TINY + ["cifar_full", "ldm_updown_tiny"]:
        qnn.set_quant_state(True, False)
        d, cos, mx = _metrics(_run(qnn, fx, cuda), fx["out_w"])
        print(f"[{name}] W-only: max|diff|={d:.3e}")
        assert d <= 1e-3 * mx
        qnn.set_quant_state(False, False)
        d, cos, mx = _metrics(_run(qnn, fx, cuda), fx["out_fp"])
        print(f"[{name}] fp: max|diff|={d:.3e}")
        assert d <= 1e-3 * mx


@pytest.mark.parametrize("name", TINY + FULL)
def test_fp16_activation_stream_envelope(cuda, name):
    """Opt-in fp16 storage of the inter-kernel activations (engine.set_stream_dtype(torch.float16): the precision the
    reference's scripts run at by default, `--precision autocast`; every kernel still computes exact integers / fp32): the
    whole-UNet output against the reference's fp32 golden and against the fp64 evaluation of the same network, with the
    envelope test of test_quantised_unet_matches_reference at the (wider) bounds stated below."""
    from qdiff import engine
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)                                       # quantisers initialised / resumed in fp32
    y32 = _run(qnn, fx, cuda)
    engine.set_stream_dtype(torch.float16)
    try:
        y = _run(qnn, fx, cuda)
    finally:
        engine.set_stream_dtype(torch.float32)
    assert y.dtype == torch.float32 and torch.isfinite(y).all()
    d, cos, mx = _metrics(y, fx["out_wa"])
    y64 = _oracle64(fx)
    d64, cos64, _ = _metrics(y.double(), y64)
    dself, cosself, _ = _metrics(fx["out_wa"].double(), y64)
    d3264, _, _ = _metrics(y32.double(), y64)
    print(f"\n[{name}] fp16 stream vs reference fp32: {d / mx:.2e} of range (cos {cos:.7f}); vs fp64 oracle: {d64 / mx:.2e} "
          f"(fp32 stream: {d3264 / mx:.2e}; the reference's own fp32: {dself / mx:.2e})")
    # Stated bound of the fp16 stream: 2x (fp32 stream: 1.25x / 1.5x) the reference's own fp32-vs-fp64 distance, from the fp64
    # evaluation and from the reference's fp32 output.  Measured round 3 (distance from fp64 / the reference's own): cifar_tiny
    # 1.80x, sd_tiny 1.45x, ldm_tiny 1.04x, ldm_full 1.15x, cifar_full 0.87x, sd_full 0.93x — two tiny fixtures sit outside
    # the fp32-stream bounds, and the SD evaluation is not faster (22.0 vs 21.8 ms): fp32 stays the default and the headline.
    assert d64 <= 2.0 * dself + 1e-3 * mx, f"{name}: fp16 stream {d64 / mx:.3e} of range from the fp64 evaluation, the reference {dself / mx:.3e}"
    assert d <= 2.0 * dself + 1e-3 * mx
    assert cos >= 0.995 and cos64 >= cosself - 2e-3


@pytest.mark.parametrize("name", TINY)
def test_state_dict_schema_matches_reference(cuda, name):
    """Key names and shapes of the saved checkpoint are the reference's."""
    from qdiff.utils import export_cali_state_dict
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    sd = export_cali_state_dict(qnn)
    want = {k: tuple(s) for k, s in fx["keys"]}
    got = {k: tuple(v.shape) for k, v in sd.items()}
    assert got == want
    # and the values survived the resume round trip
    ck = build_ckpt(fx)
    for k, v in sd.items():
        assert torch.equal(v.cpu().float(), ck[k].float()), k
    # attribute types after resume (reference utils.py:443-457)
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    for m in qnn.modules():
        if isinstance(m, AdaRoundQuantizer):
            assert torch.is_tensor(m.delta) and not isinstance(m.delta, torch.nn.Parameter)
        elif isinstance(m, UniformAffineQuantizer) and m.inited:
            assert isinstance(m.zero_point, int) and isinstance(m.delta, torch.nn.Parameter)


def test_quant_state_can_flip_between_forwards(cuda):
    """set_quant_state may be toggled at any time: results are stable."""
    fx = load_fixture("model_cifar_tiny.pt")
    qnn = _resume(fx, cuda)
    y1 = _run(qnn, fx, cuda)
    qnn.set_quant_state(False, False)
    _run(qnn, fx, cuda)
    qnn.set_quant_state(True, True)
    assert torch.equal(_run(qnn, fx, cuda), y1)


@pytest.mark.parametrize("name", ["sd_tiny", "cifar_tiny"])
def test_packed_checkpoint_round_trip_on_gpu(cuda, name, tmp_path):
    """save_packed_ckpt -> load_packed_ckpt into a model with different fp32 weights: identical output on the GPU
    (the packed codes are what the kernels read; nothing is re-quantised) and the fp32 weights are released."""
    import qdiff
    from qdiff.utils import load_packed_ckpt, save_packed_ckpt
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    y0 = _run(qnn, fx, cuda)
    path = str(tmp_path / "packed.pth")
    save_packed_ckpt(qnn, path)
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    other = build_engine_model(spec).to(cuda)
    with torch.no_grad():
        for p in other.parameters():
            p.add_(torch.randn_like(p) * 0.5)
    q2 = qdiff.QuantModel(other, wq, aq, sm_abit=spec["sm_abit"]).to(cuda).eval()
    load_packed_ckpt(q2, path)
    y1 = _run(q2, fx, cuda)
    assert torch.equal(y0, y1)
    assert all(m.weight.numel() == 0 for m in q2.modules() if isinstance(m, qdiff.QuantModule))
    # ... and that output is the REFERENCE's (golden made by the real reference from the fp32 checkpoint), inside the
    # envelope of the whole-UNet test: the packed file carries everything the reference-format checkpoint determined
    ref = fx["out_wa"]
    d = (y1 - ref).abs().max().item()
    dself = (ref.double() - _oracle64(fx)).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(y1.flatten(), ref.flatten(), dim=0).item()
    assert d <= 1.5 * dself + 1e-3 * ref.abs().max().item() and cos >= 0.995, (d, dself, cos)


@pytest.mark.parametrize("name", ["ldm_tiny", "ldm_full", "cifar_full"])
def test_ldm_attention_qkv_operand_projections_on_gpu(cuda, name, monkeypatch, tmp_path):
    """The LDM AttentionBlock's qkv conv1d as three GEMMs with attention-operand epilogues (QuantModule.head_plans,
    QuantAttentionBlock._forward_heads; reference quant_block.py:163-187): the same quantiser inputs (I*scale + bias in one
    fma, times the q / k prescale) as the fp32 projection + qd_quantize_heads route, so the UNet output is the same bit for
    bit; the route is taken where the token count is a multiple of 128 and survives a packed checkpoint when the heads are
    whole 32-row tiles (LDM-4: 32 channels per head).  cifar_full: the q / k / v convolutions of the DDIM AttnBlock (int8 weights,
    one 256-channel head, 256 tokens; reference quant_block.py:354-386) through the same epilogues."""
    import qdiff
    from qdiff import hip, quant_block
    from qdiff.utils import load_packed_ckpt, save_packed_ckpt
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    calls = {"heads": 0, "float": 0}
    real_conv, real_qh = hip.conv2d_i8, hip.quantize_heads

    def counting_conv(cc, acc_out=None):
        calls["heads"] += cc.epilogue in (hip.EPI_HEADS_I8, hip.EPI_HEADS_T_I8) and (cc.heads["H"] > 1 or cc.wbits == 8)
        return real_conv(cc, acc_out)

    def counting_qh(*a, **k):
        calls["float"] += 1
        return real_qh(*a, **k)
    real_group = hip.conv2d_i8_group

    def counting_group(ccs):                               # q / k / v as ONE grouped launch: three projections
        calls["heads"] += sum(cc.epilogue in (hip.EPI_HEADS_I8, hip.EPI_HEADS_T_I8) and (cc.heads["H"] > 1 or cc.wbits == 8) for cc in ccs)
        return real_group(ccs)
    monkeypatch.setattr(hip, "conv2d_i8", counting_conv)
    monkeypatch.setattr(hip, "conv2d_i8_group", counting_group)
    monkeypatch.setattr(hip, "quantize_heads", counting_qh)
    qnn.enable_hip_graphs(False)
    monkeypatch.setattr(quant_block, "QKV_HEADS", False)
    want = _run(qnn, fx, cuda)
    assert calls["heads"] == 0 and calls["float"] > 0
    n_float = calls["float"]
    monkeypatch.setattr(quant_block, "QKV_HEADS", True)
    calls.update(heads=0, float=0)
    got = _run(qnn, fx, cuda)
    assert calls["heads"] > 0 and calls["heads"] + calls["float"] == n_float, calls
    assert torch.equal(got, want)
    if name == "ldm_full":
        n_heads = calls["heads"]
        path = str(tmp_path / "packed.pth")
        save_packed_ckpt(qnn, path)
        spec = fx["spec"]
        wq, aq = quant_params(spec)
        q2 = qdiff.QuantModel(build_engine_model(spec).to(cuda), wq, aq, sm_abit=spec["sm_abit"]).to(cuda).eval()
        load_packed_ckpt(q2, path)
        q2.enable_hip_graphs(False)
        calls.update(heads=0, float=0)
        assert torch.equal(_run(q2, fx, cuda), want)
        assert calls["heads"] == n_heads, calls


@pytest.mark.parametrize("name", ["sd_tiny", "ldm_tiny"])
def test_hip_graph_replay_equals_eager(cuda, name):
    """bench.py measures HIP-graph replay: the replayed evaluation must equal the eager one bit for bit, for inputs
    different from the ones the graph was captured with (shared operand buffers, split-K scratch, statistics buffers)."""
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume(fx, cuda)
    x, t, c = (a.to(cuda) if a is not None else None for a in fixture_inputs(fx, "test"))
    g = torch.Generator(device=cuda).manual_seed(5)
    x2 = torch.randn(x.shape, device=cuda, generator=g)
    t2 = (t + 37) % 1000
    c2 = torch.randn(c.shape, device=cuda, generator=g) if c is not None else None
    run = lambda a, b, cc: (qnn(a, b, cc) if cc is not None else qnn(a, b)).clone()
    with torch.no_grad():
        e1, e2 = run(x, t, c), run(x2, t2, c2)
        qnn.enable_hip_graphs(True)
        g1 = run(x, t, c)            # captures
        g2 = run(x2, t2, c2)         # replays with new inputs
        g1b = run(x, t, c)
        qnn.enable_hip_graphs(False)
    torch.cuda.synchronize()
    assert torch.equal(e1, g1) and torch.equal(e2, g2) and torch.equal(g1, g1b)


def test_prepared_context_under_the_fp16_stream(cuda):
    """the preparation runs outside an evaluation, where engine._EFFECTIVE is unset — it must quantise the rows the
    evaluation's own branch would (this model's stream verdict), so that prepared == per-evaluation, bit for bit, also with
    QDIFF_STREAM=fp16."""
    from qdiff import engine, quant_block as qb
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume(fx, cuda)
    qnn.enable_hip_graphs(False)
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    engine.set_stream_dtype(torch.float16)
    try:
        with torch.no_grad():
            qb._CTX_AUTO, auto = False, qb._CTX_AUTO
            try:
                want = qnn(x, t, c).clone()               # per-evaluation branch
            finally:
                qb._CTX_AUTO = auto
            ckv = qnn.__dict__["_ctx_kv"]
            assert qnn.prepare_context(c) and engine._EFFECTIVE[0] is None
            runs = ckv.chain_runs
            got = qnn(x, t, c.clone()).clone()
            assert ckv.chain_runs == runs
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    finally:
        engine.set_stream_dtype(torch.float32)


def test_prepared_context_changes_nothing(cuda, monkeypatch):
    """QuantModel.prepare_context on the GPU: the evaluation of a prepared context (cross-attention K / V^T operands and
    key-term tables computed once, reference quant_block.py:193-195 recomputes them per evaluation) equals the per-evaluation
    computation bit for bit — eager and as a HIP graph, for two different contexts (two slots since round 5: each with the
    graph captured for it), and with an unprepared tensor falling back to the ordinary graph (automatic preparation off)."""
    from qdiff import quant_block as qb
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume(fx, cuda)
    monkeypatch.setattr(qb, "_CTX_AUTO", False)           # explicit preparation is the subject
    qnn.enable_hip_graphs(False)
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    g = torch.Generator(device=cuda).manual_seed(11)
    c2 = torch.randn(c.shape, device=cuda, generator=g)
    x2 = torch.randn(x.shape, device=cuda, generator=g)
    run = lambda a, cc: qnn(a, t, cc).clone()
    with torch.no_grad():
        want = {k: run(*k_args) for k, k_args in (("x,c", (x, c)), ("x2,c", (x2, c)), ("x,c2", (x, c2)), ("x2,c2", (x2, c2)))}
        ckv = qnn.__dict__["_ctx_kv"]
        assert qnn.prepare_context(c)
        runs = ckv.chain_runs
        got_eager = run(x, c)
        qnn.enable_hip_graphs(True)
        g1 = run(x, c)                       # captures the prepared evaluation (slot 0)
        g2 = run(x2, c)                      # replays it
        assert ckv.chain_runs == runs, "a prepared context ran the to_k / to_v chain"
        u1 = run(x, c2)                      # c2 is not prepared: the ordinary graph (context copied in, chain inside)
        assert ckv.chain_runs > runs
        assert qnn.prepare_context(c2)       # second slot
        runs = ckv.chain_runs
        g3 = run(x, c2)                      # the prepared evaluation of slot 1: its own graph
        g4 = run(x2, c2)
        assert ckv.chain_runs == runs
        u2 = run(x2, c.clone())              # c is still prepared (slot 0), recognised by value: its graph replays
        assert ckv.chain_runs == runs and len(qnn._graphs) == 3
        qnn.enable_hip_graphs(False)
    torch.cuda.synchronize()
    which = lambda y: [k for k, w in want.items() if torch.equal(y, w)]
    got = {"eager": which(got_eager), "g1": which(g1), "g2": which(g2), "u1": which(u1), "g3": which(g3), "g4": which(g4), "u2": which(u2)}
    assert got == {"eager": ["x,c"], "g1": ["x,c"], "g2": ["x2,c"], "u1": ["x,c2"], "g3": ["x,c2"], "g4": ["x2,c2"], "u2": ["x2,c"]}, got


def test_default_graph_replay_and_value_matched_context(cuda, monkeypatch):
    """the path bench.py measures is the path an UNMODIFIED caller takes.  No enable_hip_graphs(), no
    prepare_context(): the model is called with a FRESH context tensor every time (plms.py:184-187 builds torch.cat([uncond, c])
    per step).  Expected: the context chain runs once (first sight), every later call recognises the bytes, the evaluation is
    captured on the second sight of the call signature and replayed from then on — and every output equals the all-eager,
    per-evaluation-chain output bit for bit; an identity-matched and a value-matched context give identical bytes; a second
    prompt takes the second slot and its own graph; a re-assigned quantiser parameter drops graphs and prepared contexts."""
    from qdiff import quant_block as qb
    fx = load_fixture("model_sd_tiny.pt")
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    g = torch.Generator(device=cuda).manual_seed(21)
    c2 = torch.randn(c.shape, device=cuda, generator=g)
    xs = [x] + [torch.randn(x.shape, device=cuda, generator=g) for _ in range(4)]
    ref = _resume(fx, cuda)
    ref.enable_hip_graphs(False)
    monkeypatch.setattr(qb, "_CTX_AUTO", False)
    with torch.no_grad():
        want = [ref(xx, t, c).clone() for xx in xs]
        want2 = [ref(xx, t, c2).clone() for xx in xs]
    monkeypatch.setattr(qb, "_CTX_AUTO", True)
    qnn = _resume(fx, cuda)                                # defaults: graph replay on (second sight), contexts prepared on first sight
    ckv = qnn.__dict__["_ctx_kv"]
    assert qnn._graphs == {} and not ckv._pins
    with torch.no_grad():
        runs = ckv.chain_runs
        got = [qnn(xx, t, c.clone()).clone() for xx in xs]          # a fresh context tensor per call
        assert ckv.chain_runs == runs + 1 and ckv.value_matches >= len(xs) - 1 and len(qnn._graphs) == 1
        y_id = qnn(xs[1], t, c).clone()                              # the same bytes through another object ...
        y_id2 = qnn(xs[1], t, c).clone()                             # ... which is then recognised by identity
        got2 = [qnn(xx, t, c2.clone()).clone() for xx in xs]        # second prompt: second slot, its own graph
        assert ckv.chain_runs == runs + 2 and len(qnn._graphs) == 2 and len(ckv._pins) == 2
        back = qnn(xs[2], t, c.clone()).clone()                      # the first prompt is still prepared
        assert ckv.chain_runs == runs + 2
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, want)) and all(torch.equal(a, b) for a, b in zip(got2, want2))
        assert torch.equal(y_id, want[1]) and torch.equal(y_id2, want[1]) and torch.equal(back, want[2])
        # a quantiser parameter re-assigned behind the model's back (what resume_cali_model does): nothing stale survives
        mod = next(m for m in qnn.modules() if isinstance(m, qb.QuantBasicTransformerBlock)).attn2.to_k
        for mm in (mod, next(m for m in ref.modules() if isinstance(m, qb.QuantBasicTransformerBlock)).attn2.to_k):
            mm.act_quantizer.delta = torch.nn.Parameter(mm.act_quantizer.delta.detach() * 1.5)
        tok = qnn.state_token()
        y_new = qnn(xs[0], t, c.clone()).clone()
        assert qnn.state_token() == tok and len(qnn._graphs) <= 1 and ckv.chain_runs == runs + 3
        monkeypatch.setattr(qb, "_CTX_AUTO", False)
        w_new = ref(xs[0], t, c).clone()
        torch.cuda.synchronize()
        assert torch.equal(y_new, w_new) and not torch.equal(w_new, want[0])


def test_hooks_registered_after_capture_keep_the_model_eager(cuda):
    """a replayed graph runs no Python, so a forward hook registered on a sub-module AFTER a graph of the
    call signature exists (the reference's calibration capture, reference qdiff/utils.py:190-255; any recorder) must switch the
    evaluation back to the eager walk — and the graph is used again once the hook is gone."""
    from qdiff.quant_block import QuantBasicTransformerBlock
    fx = load_fixture("model_sd_tiny.pt")
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    qnn = _resume(fx, cuda)
    with torch.no_grad():
        ys = [qnn(x, t, c).clone() for _ in range(3)]
        assert len(qnn._graphs) == 1
        g = next(iter(qnn._graphs.values()))
        replays = []
        orig = g.graph.replay
        g.graph.replay = lambda: (replays.append(1), orig())[1]
        fired = []
        # (a reconstruction unit: the fused integer walk calls blocks through __call__, single layers inside them it drives directly)
        mod = next(m for m in qnn.modules() if isinstance(m, QuantBasicTransformerBlock))
        h = mod.register_forward_hook(lambda _m, _a, _o: fired.append(1))
        y_hooked = qnn(x, t, c).clone()
        assert fired and not replays and len(qnn._graphs) == 1
        h.remove()
        y_back = qnn(x, t, c).clone()
        assert replays and len(fired) == 1
    torch.cuda.synchronize()
    assert all(torch.equal(y, ys[0]) for y in ys[1:] + [y_hooked, y_back])


def test_repreparing_a_locked_context_keeps_the_handle(cuda):
    """ContextKV.pin of bytes that are already prepared refreshes the entry IN PLACE — the handle lock_context
    handed out earlier stays the live entry, unlock_context releases it, and no slot is leaked."""
    fx = load_fixture("model_sd_tiny.pt")
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    qnn = _resume(fx, cuda)
    ckv = qnn.__dict__["_ctx_kv"]
    with torch.no_grad():
        qnn(x, t, c)
        h = qnn.lock_context(c)
        assert h is not None and h["locked"] == 1
        gen = h["gen"]
        assert qnn.prepare_context(c.clone())                      # the same prompt again (plms_sample / a second sampler)
        assert len(ckv._pins) == 1 and ckv._pins[0] is h and h["locked"] == 1 and h["gen"] == gen + 1
        qnn.unlock_context(h)
        assert ckv._pins[0]["locked"] == 0
        for i in range(4):                                        # unlocked entries are evicted again: the slots do not pile up
            qnn.prepare_context(torch.randn_like(c))
        assert len(ckv._pins) <= 2
    torch.cuda.synchronize()


def test_two_whole_step_samplers_share_one_model(cuda):
    """a DevicePLMS whole-step graph reads the prepared cross-attention operands of ITS conditioning.  Two
    samplers with different prompts on one model, stepped alternately, and a third prompt sampled in between: each run
    reproduces its own eager trajectory bit for bit (the entries are locked per sampler; the third prompt takes a third slot)."""
    from qdiff import sampling
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume(fx, cuda)
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    g = torch.Generator(device=cuda).manual_seed(31)
    rnd = lambda ref: torch.randn(ref.shape, device=cuda, generator=g)
    prompts = [(c, rnd(c)), (rnd(c), rnd(c)), (rnd(c), rnd(c))]
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 5, eta=0.0)
    with torch.no_grad():
        want = [sampling.plms_sample(qnn, x, table, cond=cc, uncond=uu, scale=3.0).clone() for cc, uu in prompts]
        qnn.release_context()
        s1 = sampling.DevicePLMS(qnn, table, x, cond=prompts[0][0], uncond=prompts[0][1], scale=3.0, use_graph=True)
        s2 = sampling.DevicePLMS(qnn, table, x, cond=prompts[1][0], uncond=prompts[1][1], scale=3.0, use_graph=True)
        ckv = qnn.__dict__["_ctx_kv"]
        assert sum(e["locked"] for e in ckv._pins) == 2
        third = None
        for k in range(s1.total):
            s1.step(k)
            if k == 2:                                     # another prompt on the same model, mid-run
                third = sampling.plms_sample(qnn, x, table, cond=prompts[2][0], uncond=prompts[2][1], scale=3.0).clone()
            s2.step(k)
        torch.cuda.synchronize()
        assert torch.equal(s1.x, want[0]) and torch.equal(s2.x, want[1]) and torch.equal(third, want[2])
        assert len(ckv._pins) == 3
        s1.close(); s2.close()
        assert sum(e["locked"] for e in ckv._pins) == 0


def test_whole_step_generalized_sampler_graph_equals_eager(cuda):
    """DeviceGeneralizedSteps (BASELINE configs[1]: CIFAR-10 DDIM, denoising.py:10-32) as one HIP graph per step == the eager
    generalized_steps loop on the quantised pixel-space UNet, bit for bit."""
    from qdiff import sampling
    fx = load_fixture("model_cifar_tiny.pt")
    qnn = _resume(fx, cuda)
    x, t, _ = fixture_inputs(fx, "test")
    x = x.to(cuda)
    betas = torch.from_numpy(sampling.ddpm_betas()).float().to(cuda)
    seq = sampling.quad_sequence(1000, 6)
    with torch.no_grad():
        want = sampling.generalized_steps(lambda xx, tt: qnn(xx, tt), x, seq, betas, eta=0.0)
        got = sampling.DeviceGeneralizedSteps(qnn, x, seq, betas, use_graph=True).run()
    torch.cuda.synchronize()
    assert torch.isfinite(want).all() and torch.equal(got, want)


def test_whole_step_graph_plms_equals_eager_sampler(cuda):
    """One HIP graph per PLMS step (UNet on the CFG batch + guidance + multistep update, DevicePLMS) reproduces the eager
    plms_sample loop bit for bit on a quantised SD-style UNet."""
    from qdiff import sampling
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume(fx, cuda)
    x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
    g = torch.Generator(device=cuda).manual_seed(9)
    uc = torch.randn(c.shape, device=cuda, generator=g)
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 6, eta=0.0)
    unet = lambda xx, tt, cc=None: qnn(xx, tt, cc)
    with torch.no_grad():
        want = sampling.plms_sample(unet, x, table, cond=c, uncond=uc, scale=3.0)
        got = sampling.DevicePLMS(unet, table, x, cond=c, uncond=uc, scale=3.0, use_graph=True).run()
    torch.cuda.synchronize()
    assert torch.isfinite(want).all() and torch.equal(got, want)


def test_device_plms_on_gpu_matches_reference_sampler_golden(cuda):
    """DevicePLMS (device-resident step state, one HIP graph per PLMS step) driving the deterministic stub eps-model ON THE
    GPU vs the trajectory the real reference's PLMSSampler produced with the same stub (tests/golden/samplers.pt): the
    sampler arithmetic is identical; tanh / division on the GPU differ from the CPU's in the last ulps, hence 2e-5 of range
    over 50 steps instead of bit equality (the CPU test test_sampling_sharding.py is bit-exact)."""
    from qdiff import sampling
    from test_sampling_sharding import stub_eps
    fx = load_fixture("samplers.pt")["plms"]
    calls = []

    def unet(x, t, c=None):
        calls.append(x.shape[0])
        return stub_eps(x, t, c)
    table = sampling.StepTable(sampling.ldm_betas(fx["ls"], fx["le"]), fx["steps"], eta=0.0)
    with torch.no_grad():
        got = sampling.DevicePLMS(unet, table, fx["xT"].to(cuda), cond=fx["c"].to(cuda), uncond=fx["uc"].to(cuda),
                                  scale=fx["scale"], use_graph=True).run()
    torch.cuda.synchronize()
    want = fx["out"]
    assert (got.cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


def _foreign_reference_packages(monkeypatch):
    """Stand-ins for the reference's `ldm` / `ddim` packages (the GPU box has no reference tree): every class
    qdiff.quant_block.reference_classes() looks up exists as a DISTINCT type — a subclass of this repo's class whose own
    forward counts its calls — so `type(m) is ref[...]` matches exactly as it does on the real reference classes."""
    import types
    from qdiff import quant_block
    from qdiff.arch import ddim_unet, ldm_unet
    calls = {"foreign_forward": 0}

    def foreign(base):
        def forward(self, *a, **k):
            calls["foreign_forward"] += 1
            return base.forward(self, *a, **k)
        return type(base.__name__, (base,), {"forward": forward, "__module__": "foreign"})
    oai = types.ModuleType("ldm.modules.diffusionmodules.openaimodel")
    att = types.ModuleType("ldm.modules.attention")
    for name in ("ResBlock", "AttentionBlock", "QKMatMul", "SMVMatMul", "TimestepBlock", "Upsample", "Downsample", "UNetModel",
                 "TimestepEmbedSequential"):
        setattr(oai, name, foreign(getattr(ldm_unet, name)))
    for name in ("BasicTransformerBlock", "SpatialTransformer"):
        setattr(att, name, foreign(getattr(ldm_unet, name)))
    dd = types.ModuleType("ddim.models.diffusion")
    for name in ("ResnetBlock", "AttnBlock", "Model", "Upsample", "Downsample"):
        setattr(dd, name, foreign(getattr(ddim_unet, name)))
    pk = {"ldm": types.ModuleType("ldm"), "ldm.modules": types.ModuleType("ldm.modules"),
          "ldm.modules.diffusionmodules": types.ModuleType("ldm.modules.diffusionmodules"),
          "ldm.modules.diffusionmodules.openaimodel": oai, "ldm.modules.attention": att,
          "ddim": types.ModuleType("ddim"), "ddim.models": types.ModuleType("ddim.models"), "ddim.models.diffusion": dd}
    pk["ldm"].modules = pk["ldm.modules"]
    pk["ldm.modules"].diffusionmodules = pk["ldm.modules.diffusionmodules"]
    pk["ldm.modules"].attention = att
    pk["ldm.modules.diffusionmodules"].openaimodel = oai
    pk["ddim"].models = pk["ddim.models"]
    pk["ddim.models"].diffusion = dd
    for k, v in pk.items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setattr(quant_block, "_REF_CACHE", {})
    return oai, att, dd, calls


@pytest.mark.parametrize("name", ["sd_tiny", "cifar_tiny"])
def test_foreign_model_classes_run_this_repos_forwards_on_the_gpu(cuda, monkeypatch, name):
    """VERDICT r02 missing #6, narrowed as far as a box without the reference tree allows: a UNet whose modules are instances
    of FOREIGN classes (found through `ldm.*` / `ddim.*` the way the reference's are) is wrapped by QuantModel; in the
    (True, True) state (a) the forwards bound onto the foreign SpatialTransformer / Upsample / Downsample / UNet walk are the
    very function objects this repo's own arch classes run — the ones the teacher-forced GPU tests verify —, (b) no foreign
    forward is entered on the integer path, (c) the output equals this repo's own classes' bit for bit and reproduces the
    reference golden inside the envelope."""
    import qdiff
    from qdiff.arch import ddim_unet, ldm_unet
    from qdiff.utils import resume_cali_model
    fx = load_fixture(f"model_{name}.pt")
    spec = fx["spec"]
    want = _run(_resume(fx, cuda), fx, cuda)                      # this repo's own classes
    oai, att, dd, calls = _foreign_reference_packages(monkeypatch)
    model = build_engine_model(spec).to(cuda)
    remap = {ldm_unet.ResBlock: oai.ResBlock, ldm_unet.AttentionBlock: oai.AttentionBlock, ldm_unet.QKMatMul: oai.QKMatMul,
             ldm_unet.SMVMatMul: oai.SMVMatMul, ldm_unet.Upsample: oai.Upsample, ldm_unet.Downsample: oai.Downsample,
             ldm_unet.UNetModel: oai.UNetModel, ldm_unet.TimestepEmbedSequential: oai.TimestepEmbedSequential,
             ldm_unet.BasicTransformerBlock: att.BasicTransformerBlock, ldm_unet.SpatialTransformer: att.SpatialTransformer,
             ddim_unet.ResnetBlock: dd.ResnetBlock, ddim_unet.AttnBlock: dd.AttnBlock, ddim_unet.Model: dd.Model,
             ddim_unet.Upsample: dd.Upsample, ddim_unet.Downsample: dd.Downsample}
    n_foreign = 0
    for m in model.modules():
        if type(m) in remap:
            m.__class__ = remap[type(m)]
            n_foreign += 1
    assert n_foreign > 5
    wq, aq = quant_params(spec)
    qnn = qdiff.QuantModel(model, wq, aq, sm_abit=spec["sm_abit"]).to(cuda).eval()
    cal = tuple(a for a in fixture_inputs(fx, "cal") if a is not None)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt.pth")
        torch.save(build_ckpt(fx), path)
        resume_cali_model(qnn, path, cal, quant_act=True, cond=spec["ctx"] is not None)
    own = {att.SpatialTransformer: ldm_unet.SpatialTransformer.forward, oai.Upsample: ldm_unet.Upsample.forward,
           oai.Downsample: ldm_unet.Downsample.forward, dd.Upsample: ddim_unet.Upsample.forward, dd.Downsample: ddim_unet.Downsample.forward}
    bound = 0
    for m in qnn.model.modules():
        if type(m) in own and getattr(m, "dims", 2) == 2:
            assert getattr(m.forward, "__func__", None) is own[type(m)], type(m).__name__
            bound += 1
    assert bound >= 2
    calls["foreign_forward"] = 0
    got = _run(qnn, fx, cuda)
    assert calls["foreign_forward"] == 0, "a foreign (reference-side) forward ran in the integer state"
    assert torch.equal(got, want)
    ref = fx["out_wa"].float()
    rng = ref.abs().max().item()
    dself = (ref.double() - _oracle64(fx)).abs().max().item() if _oracle64(fx) is not None else 0.05 * rng
    assert (got - ref).abs().max().item() <= 1.5 * dself + 1e-3 * rng
