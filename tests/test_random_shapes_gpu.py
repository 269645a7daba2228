"""Seeded random shapes through the contraction and the attention entry points (GPU).

The parametrised tests of test_hip_kernels.py pin the shapes the three UNets launch; this file walks the space between them:
every draw is checked against the CPU oracle (exact int32 accumulators; fp32 rows within the stated bound) AND across the
launch variants the library chooses between (four-wave block vs two K-groups, split-K vs one pass, register-fed vs LDS-staged
attention), which must agree bit for bit.  Seeds are fixed: a failure names its draw.
"""
import random
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn.functional as F

from oracle import quant_ref as R
from test_hip_kernels import _aq, _codes, _weight_quantizer

pytestmark = pytest.mark.gpu


def _conv_draws(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        k = rng.choice([1, 1, 3, 3, 3])
        wbits = rng.choice([4, 4, 8])
        H = rng.choice([4, 7, 8, 12, 16, 16, 24, 32])
        B = rng.choice([1, 2, 3, 4, 8])
        Cin = 8 * rng.randint(1, 96) if rng.random() < 0.6 else rng.choice([128, 192, 224, 256, 320, 448, 640, 960, 1280])
        Cout = rng.choice([8 * rng.randint(1, 60), 64, 128, 160, 224, 256, 320, 448, 640])
        if B * H * H * Cin * k * k > 6_000_000:          # keep the oracle's integer convolution in seconds
            B, H = 1, min(H, 16)
        stride = 2 if (k == 3 and H % 2 == 0 and rng.random() < 0.2) else 1
        out.append((i, B, Cin, H, Cout, k, wbits, rng.random() < 0.3, rng.random() < 0.5, rng.random() < 0.5, stride))
    return out


CONV_DRAWS = _conv_draws(150, 20260930)


@pytest.mark.parametrize("draw", CONV_DRAWS, ids=[f"conv{d[0]}_B{d[1]}_C{d[2]}_H{d[3]}_N{d[4]}_k{d[5]}_w{d[6]}_s{d[10]}" for d in CONV_DRAWS])
def test_random_conv_shapes(cuda, draw):
    from qdiff import engine, hip
    i, B, Cin, H, Cout, k, wbits, a_sym, with_rowbias, with_residual, stride = draw
    g = torch.Generator().manual_seed(1000 + i)
    x = torch.randn(B, Cin, H, H, generator=g)
    x = x if a_sym else F.silu(x)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    bias = torch.randn(Cout, generator=g)
    q = _weight_quantizer(w, wbits, True, g)
    d, z = R.uaq_init_scale(x, 8, a_sym, False, "max")
    aq = _aq(d, z, 8, a_sym)
    pack = engine.pack_module_weights(w.to(cuda), [q], 0)
    plan = engine.build_conv_plan(pack, [aq], k, k, stride, k // 2, bias.to(cuda))
    Ho, _ = engine.conv_out_hw(H, H, plan)
    M = B * Ho * Ho
    xq = engine.quantize_rows(x.to(cuda), plan, B, Cin, H * H, (Cin * H * H, H * H, 1))
    rowbias = torch.randn(B, Cout, generator=g).to(cuda) if with_rowbias else None
    residual = torch.randn(M, Cout, generator=g).to(cuda) if with_residual else None

    # 1. exact int32 accumulators against the integer oracle
    acc = torch.zeros((M, Cout), dtype=torch.int32, device=cuda)
    engine.conv_forward(plan, xq, B, H, H, acc_out=acc)
    torch.cuda.synchronize()
    xc = R.uaq_codes(x, aq.delta, aq.zero_point, 8, a_sym)
    want_acc = R.int_conv_exact(xc, int(z), _codes(w, q), q.zero_point.reshape(-1).long(), "conv2d", dict(stride=stride, padding=k // 2))
    got_acc = acc.cpu().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2).long()
    assert torch.equal(got_acc, want_acc), f"accumulators: max |diff| = {(got_acc - want_acc).abs().max().item()}"

    # 2. fp32 rows: every launch variant the library may pick produces the same bytes ...
    outs = {}
    try:
        for kg in (1, 0):
            for sk in (None, False):
                hip.conv_config(kgroups=kg)
                kw = {} if sk is None else dict(splitk=False)
                o = engine.conv_forward(plan, xq, B, H, H, rowbias=rowbias, residual=residual, **kw)
                torch.cuda.synchronize()
                outs[(kg, sk)] = o.clone()
    finally:
        hip.conv_config(kgroups=1)
    ref = outs[(0, False)]
    assert torch.isfinite(ref).all()
    for key, o in outs.items():
        assert torch.equal(o, ref), key          # K-groups: integer adds commute; split-K: int32 partials, the finalise pass runs the epilogue's float sequence

    # ... 3. and they are the reference's fake-quantised layer within the fp32 bound of test_conv_fp32_matches_fake_quant
    want = R.quant_module_forward(x, w, bias, "conv2d", dict(stride=stride, padding=k // 2),
                                  [dict(delta=q.delta, zero_point=q.zero_point, alpha=q.alpha, n_levels=q.n_levels)],
                                  [dict(delta=aq.delta, zero_point=z, n_bits=8, sym=a_sym)])
    if rowbias is not None:
        want = want + rowbias.cpu()[:, :, None, None]
    if residual is not None:
        want = want + residual.cpu().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    got = ref.cpu().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-6


def _attn_draws(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        d = rng.choice([8, 24, 40, 40, 48, 56, 72, 80])
        smb = rng.choice([16, 16, 8])
        T = rng.choice([32, 77, 96, 130, 256, 300, 512])
        S = rng.choice([19, 32, 77, 160, 237, 512, 640, 1024])
        out.append((i, rng.choice([1, 2]), rng.choice([1, 2, 4]), T, S, d, smb, rng.choice([0.3, 1.0, 3.0]), rng.random() < 0.25))
    return out


ATTN_DRAWS = _attn_draws(60, 930)


@pytest.mark.parametrize("draw", ATTN_DRAWS, ids=[f"attn{d[0]}_T{d[3]}_S{d[4]}_d{d[5]}_p{d[6]}" for d in ATTN_DRAWS])
def test_random_attention_shapes(cuda, draw):
    """Register-fed and LDS-staged kernels, table and constant-operand key terms: the same bytes; and those bytes are the integer
    oracle's within the bound of test_attention_fused."""
    from qdiff import engine, hip
    i, B, H, T, S, d, smb, sharp, qpos = draw
    g = torch.Generator().manual_seed(5000 + i)
    C = H * d
    q, k, v = (torch.randn(B, L, C, generator=g) for L in (T, S, S))
    q = q * sharp
    if qpos:
        q = q.abs()

    def mk(t, n_bits=8, always_zero=False):
        dd, zz = R.uaq_init_scale(t, n_bits, False, False, "max", always_zero)
        return dict(delta=dd, zero_point=zz, n_bits=n_bits, sym=False)
    heads = lambda t, L: t.view(B, L, H, d).permute(0, 2, 1, 3).reshape(B * H, L, d)
    scale = d ** -0.5
    p = (torch.einsum("bid,bjd->bij", heads(q, T), heads(k, S)) * scale).softmax(-1)
    aq_q, aq_k, aq_v, aq_w = mk(q), mk(k), mk(v), mk(p, smb, True)
    ns = lambda a: NS(delta=a["delta"], zero_point=a["zero_point"], n_bits=a["n_bits"], sym=a["sym"])
    ap = engine.build_attn_plan(ns(aq_q), ns(aq_k), ns(aq_v), ns(aq_w), scale, 1.0, cuda)
    Tp, Sp, dp = engine.pad32(T), engine.pad32(S), engine.pad32(d)
    q8 = torch.zeros((B * H, Tp, dp), dtype=torch.int8, device=cuda)
    k8 = torch.zeros((B * H, Sp, dp), dtype=torch.int8, device=cuda)
    v8 = torch.zeros((B * H, dp, Sp), dtype=torch.int8, device=cuda)
    vsum = torch.zeros((B * H, dp), dtype=torch.int32, device=cuda)
    for which, (t, L, buf) in enumerate(((q, T, q8), (k, S, k8), (v, S, v8))):
        engine.heads_from_float(ap, which, t.to(cuda), B, L, H, d, (L * C, C, d, 1), buf, vsum)
    outs = {}
    try:
        for mode in (0, 3):
            for ktab in (1, 0):
                hip.attn_config(pipe_mode=mode, ktab=ktab, lean=3 if d >= 64 else 1)
                o = engine.attention_codes(ap, q8, k8, v8, vsum, B, T, S, H, d)
                torch.cuda.synchronize()
                outs[(mode, ktab)] = o.clone()
    finally:
        hip.attn_config(pipe_mode=2, ktab=1, lean=1)
    ref = outs[(0, 1)]
    assert torch.isfinite(ref).all() and ref.abs().max() > 0
    for key, o in outs.items():
        assert torch.equal(ref, o), (key, (ref - o).abs().max().item())
    want, _ = R.attention_int(heads(q, T), heads(k, S), heads(v, S), scale, aq_q, aq_k, aq_v, aq_w, pre_scale=1.0)
    got = ref.cpu().view(B, T, H, d).permute(0, 2, 1, 3).reshape(B * H, T, d)
    rng_ = want.abs().max().item()
    diff = (got.double() - want).abs()
    assert (diff > 2e-4 * rng_).float().mean().item() <= 1e-2
    assert diff.max().item() <= 2e-2 * rng_


def _heads_draws(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        wbits = rng.choice([4, 4, 4, 8])
        H = rng.choice([1, 2, 4, 8, 14])
        d = 4 * rng.randint(2, 48)
        if wbits == 8 and H * d <= 64:
            d = 72
        if H * d > 1344:
            H = max(1, 1344 // d)
        out.append((i, 128 * rng.randint(1, 4), H * d, 8 * rng.randint(4, 160), H, wbits))
    return out


HEADS_DRAWS = _heads_draws(40, 61)


@pytest.mark.parametrize("draw", HEADS_DRAWS, ids=[f"heads{d[0]}_T{d[1]}_N{d[2]}_K{d[3]}_H{d[4]}_w{d[5]}" for d in HEADS_DRAWS])
def test_random_head_layout_epilogues(cuda, draw):
    """q / k / v projections with the attention operand bytes written from the epilogue (single launches and the grouped launch
    of round 6) against the fp32 projection + qd_quantize_heads, on head counts / head dims / K between the UNets' own."""
    from test_hip_kernels import _heads_epilogue_case
    _, T, N, K, H, wbits = draw
    _heads_epilogue_case(cuda, T, N, K, H, wbits)


def _temb_draws(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        w_bits = rng.choice([4, 8])
        B = rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 16, 20, 33, 64, 100, 130, 257])
        K = rng.choice([128, 224, 512, 1280, 16 * rng.randint(4, 80)])          # (multiples of 16: the kernel's contract, the engine falls back otherwise)
        widths = tuple(rng.choice([64, 128, 224, 256, 320, 448, 640, 1280, 8 * rng.randint(4, 100)]) for _ in range(rng.randint(1, 6)))
        out.append((i, w_bits, B, K, widths))
    return out


TEMB_DRAWS = _temb_draws(30, 17)


@pytest.mark.parametrize("draw", TEMB_DRAWS, ids=[f"temb{d[0]}_w{d[1]}_B{d[2]}_K{d[3]}_L{len(d[4])}" for d in TEMB_DRAWS])
def test_random_temb_mlp_shapes(cuda, draw):
    """qd_temb_mlp (row groups on grid.y since round 6: how many depends on the layer count and the batch) against the
    per-layer generic path, on batches / widths / layer counts around the policy's break points."""
    from test_hip_kernels import test_temb_mlp_equals_generic_path
    _, w_bits, B, K, widths = draw
    test_temb_mlp_equals_generic_path(cuda, w_bits, B, K, widths)


def _misc_draws(seed):
    rng = random.Random(seed)
    splitk, fp16, gnstat, gn, ups = [], [], [], [], []
    while len(splitk) < 16:
        k = rng.choice([1, 3])
        H, W = rng.choice([(1, 1), (1, 77), (4, 4), (8, 8), (7, 9), (16, 16)])
        B, Cin, N = rng.choice([1, 2, 4, 16]), 32 * rng.randint(4, 60), rng.choice([64, 128, 160, 224, 320, 640, 1280, 8 * rng.randint(4, 80)])
        bn = 160 if N % 160 == 0 else 224 if N % 224 == 0 else 128 if N > 64 else 64
        blocks = -(-B * H * W // 128) * -(-N // bn)
        if blocks <= 64 and k * k * -(-Cin // 64) >= 32:          # the shapes the library contracts in K slices (choose_splitk)
            splitk.append((f"rand{len(splitk)}", B, Cin, H, W, N, k))
    for i in range(16):
        fp16.append((rng.choice([1, 2, 3]), 8 * rng.randint(2, 80), rng.choice([8, 16, 32]), rng.choice([64, 128, 160, 196, 200, 224, 320, 448, 640, 8 * rng.randint(3, 80)]), rng.choice([1, 3])))
    for i in range(12):
        H = rng.choice([16, 32])                                  # H * H must be a multiple of 128 rows
        gnstat.append((rng.choice([1, 2, 3]), 8 * rng.randint(2, 60), H, 32 * rng.randint(1, 20), rng.choice([1, 3])))
    for i in range(16):
        gn.append((rng.random() < 0.5, 32 * rng.randint(1, 60), rng.choice([1, 16, 64, 100, 144, 256, 1024])))
    for i in range(10):
        ups.append((rng.choice([1, 2, 3]), 8 * rng.randint(2, 80), rng.choice([4, 8, 16]), rng.choice([64, 128, 160, 224, 320, 640, 8 * rng.randint(4, 60)]), rng.choice([4, 4, 8])))
    return splitk, fp16, gnstat, gn, ups


SPLITK_DRAWS, FP16_DRAWS, GNSTAT_DRAWS, GN_DRAWS, UPS_DRAWS = _misc_draws(424242)


@pytest.mark.parametrize("case", SPLITK_DRAWS, ids=[f"B{c[1]}_C{c[2]}_{c[3]}x{c[4]}_N{c[5]}_k{c[6]}" for c in SPLITK_DRAWS])
def test_random_splitk_shapes(cuda, case):
    """Small M, long K: whatever number of K slices the library picks, not a bit changes against the one-pass schedule."""
    from test_hip_kernels import test_conv_splitk_is_bit_identical_to_unsplit
    test_conv_splitk_is_bit_identical_to_unsplit(cuda, case)


@pytest.mark.parametrize("shape", FP16_DRAWS, ids=[f"B{c[0]}_C{c[1]}_H{c[2]}_N{c[3]}_k{c[4]}" for c in FP16_DRAWS])
def test_random_fp16_stream_shapes(cuda, shape):
    from test_hip_kernels import test_conv_fp16_stream_is_the_rounded_fp32_epilogue
    test_conv_fp16_stream_is_the_rounded_fp32_epilogue(cuda, shape)


@pytest.mark.parametrize("shape", GNSTAT_DRAWS, ids=[f"B{c[0]}_C{c[1]}_H{c[2]}_N{c[3]}_k{c[4]}" for c in GNSTAT_DRAWS])
def test_random_groupnorm_statistics_from_the_epilogue(cuda, shape):
    from test_hip_kernels import test_conv_emits_groupnorm_statistics
    test_conv_emits_groupnorm_statistics(cuda, *shape)


@pytest.mark.parametrize("shape", GN_DRAWS, ids=[f"silu{int(c[0])}_C{c[1]}_S{c[2]}" for c in GN_DRAWS])
def test_random_groupnorm_shapes(cuda, shape):
    from test_hip_kernels import test_groupnorm_silu_quant
    test_groupnorm_silu_quant(cuda, *shape)


@pytest.mark.parametrize("shape", UPS_DRAWS, ids=[f"B{c[0]}_C{c[1]}_h{c[2]}_N{c[3]}_w{c[4]}" for c in UPS_DRAWS])
def test_random_upsampling_fold_shapes(cuda, shape):
    from test_hip_kernels import test_conv_folds_nearest_upsampling
    test_conv_folds_nearest_upsampling(cuda, *shape)


def _geglu_draws(n, seed):
    rng = random.Random(seed)
    return [(i, rng.choice([1, 50, 128, 200, 384, 1000]), 32 * rng.randint(1, 40), 8 * rng.randint(4, 160)) for i in range(n)]


GEGLU_DRAWS = _geglu_draws(16, 4242)


@pytest.mark.parametrize("draw", GEGLU_DRAWS, ids=[f"geglu{d[0]}_M{d[1]}_F{d[2]}_K{d[3]}" for d in GEGLU_DRAWS])
def test_random_geglu_epilogue_shapes(cuda, draw):
    """The GEGLU projection with value * gelu(gate) -> the next Linear's act quantiser in its epilogue (engine.conv_forward_geglu,
    O_GEGLU; reference ldm/modules/attention.py:42-44 + quant_layer.py:248-279) against the unfused route on the same library:
    fp32 projection, then qd_geglu_quant — the same fma, the same erf polynomial, the same quotient: the same codes — and both
    against the oracle's GEGLU of the fake-quantised projection (a code apart on <= 0.1 % of elements: exact round() ties)."""
    from qdiff import engine, hip
    i, M, Fdim, K = draw
    g = torch.Generator().manual_seed(7000 + i)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(2 * Fdim, K, generator=g) * 0.05
    bias = torch.randn(2 * Fdim, generator=g) * 0.1
    q = _weight_quantizer(w, 4, True, g)
    dx, zx = R.uaq_init_scale(x, 8, False, False, "max")
    aq = _aq(dx, zx)
    want_h = R.quant_module_forward(x, w, bias, "linear", {},
                                    [dict(delta=q.delta, zero_point=q.zero_point, alpha=q.alpha, n_levels=q.n_levels)],
                                    [dict(delta=aq.delta, zero_point=zx, n_bits=8, sym=False)])
    y = R.geglu(want_h)
    dy, zy = R.uaq_init_scale(y, 8, False, False, "max")
    nxt_w = torch.randn(64, Fdim, generator=g) * 0.05
    nxt = engine.build_conv_plan(engine.pack_module_weights(nxt_w.to(cuda), [_weight_quantizer(nxt_w, 4, True, g)], 0), [_aq(dy, zy)], 1, 1, 1, 0, None)
    plain = engine.build_conv_plan(engine.pack_module_weights(w.to(cuda), [q], 0), [aq], 1, 1, 1, 0, bias.to(cuda))
    fused = engine.build_conv_plan(engine.pack_module_weights(w.to(cuda), [q], 0, row_perm=engine.geglu_row_perm(Fdim, cuda)), [aq], 1, 1, 1, 0, bias.to(cuda))
    assert fused.pack.tiled and fused.pack.wbits == 4
    xq = engine.quantize_rows(x.to(cuda), plain, 1, K, M, (0, 1, K))
    h = engine.conv_forward(plain, xq, 1, 1, M, 1, M, splitk=False)
    a = torch.zeros((M, nxt.ldx), dtype=torch.int8, device=cuda)
    hip.geglu_quant(h, M, Fdim, 2 * Fdim, nxt.qparams[0], nxt.grids[0], a, nxt.ldx)
    b = engine.conv_forward_geglu(fused, xq, M, nxt)
    torch.cuda.synchronize()
    assert torch.equal(a[:, :Fdim], b[:, :Fdim]), (a[:, :Fdim] != b[:, :Fdim]).float().mean().item()
    want = (R.uaq_codes(y, dy, zy, 8, False) - 128).to(torch.int8)
    diff = (b[:, :Fdim].cpu().int() - want.int()).abs()
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() <= 2e-3


def _ln_draws(n, seed):
    rng = random.Random(seed)
    return [(i, rng.choice([1, 7, 64, 70, 128, 300, 1024, 4096]), rng.choice([320, 640, 1280, 448, 672, 896, 16 * rng.randint(1, 96)]), rng.randint(1, 3))
            for i in range(n)]


LN_DRAWS = _ln_draws(24, 808)


@pytest.mark.parametrize("draw", LN_DRAWS, ids=[f"ln{d[0]}_M{d[1]}_C{d[2]}_n{d[3]}" for d in LN_DRAWS])
def test_random_layernorm_quant_shapes(cuda, draw):
    """qd_layernorm_quant (one to three consumers: the q / k / v or GEGLU inputs of a transformer block; every row-count / width
    form of the kernel) against torch's LayerNorm + the oracle's quantiser: a code apart on <= 0.1 % of elements."""
    from qdiff import engine, hip
    from test_hip_kernels import _code_mismatch
    i, M, C, n = draw
    g = torch.Generator().manual_seed(9000 + i)
    x = torch.randn(M, C, generator=g) * 1.7 + 0.2
    ln = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g))
        ln.bias.copy_(torch.randn(C, generator=g))
        y = ln(x)
    params = [(0.031, 120), (0.02, 133), (0.05, 100)][:n]
    outs = [torch.empty((M, C), dtype=torch.int8, device=cuda) for _ in params]
    hip.layernorm_quant(x.to(cuda), M, C, C, ln.eps, ln.weight.data.to(cuda), ln.bias.data.to(cuda),
                        [torch.tensor([d, float(z)], device=cuda) for d, z in params], [engine.act_grid(8, False)] * n, outs, C)
    torch.cuda.synchronize()
    for o, (d, z) in zip(outs, params):
        mx, frac = _code_mismatch(o.cpu(), R.uaq_codes(y, torch.tensor(d), z, 8, False) - 128)
        assert mx <= 1 and frac <= 1e-3


@pytest.mark.parametrize("C", [856, 1600])
def test_layernorm_widths_outside_the_kernel_fall_back(cuda, C):
    """engine.layernorm_quant on a width qd_layernorm_quant does not take (not a multiple of 16 / above 1536): torch's LayerNorm
    + one quantiser launch per consumer — the same codes as the oracle's quantiser on torch's output."""
    from qdiff import engine
    g = torch.Generator().manual_seed(12)
    M = 40
    x = torch.randn(M, C, generator=g)
    ln = torch.nn.LayerNorm(C).to(cuda)
    w = torch.randn(32, C, generator=g) * 0.05
    plans = [engine.build_conv_plan(engine.pack_module_weights(w.to(cuda), [_weight_quantizer(w, 4, True, g)], 0), [_aq(d, z)], 1, 1, 1, 0, None)
             for d, z in ((0.031, 120), (0.02, 133))]
    outs = engine.layernorm_quant(x.to(cuda), M, C, ln, plans)
    torch.cuda.synchronize()
    y = torch.nn.functional.layer_norm(x.to(cuda), (C,), ln.weight, ln.bias, ln.eps).cpu()
    for o, (d, z) in zip(outs, ((0.031, 120), (0.02, 133))):
        assert torch.equal(o[:, :C].cpu().int(), (R.uaq_codes(y, torch.tensor(d), z, 8, False) - 128).int())
