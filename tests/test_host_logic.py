"""CPU tests (no GPU): the C-ABI library loads and exports what include/qdiff_hip.h declares; the
host logic of the engine (plans, packing-mode decisions, fused block wiring, checkpoint resume /
export, attribute juggling) runs end to end on the ABI emulator (tests/abi_emulator.py) and matches
the real reference's golden outputs; the product path refuses to run without the GPU."""
import ctypes
import os
import re
import sys
import tempfile

import pytest
import torch

import abi_emulator
from golden_util import build_ckpt, build_engine_model, fixture_inputs, load_fixture, quant_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    import importlib.util
    spec = importlib.util.spec_from_file_location("qdiff_build", os.path.join(ROOT, "q-diffusion_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = ctypes.CDLL(mod.build())                       # hipcc cross-compiles gfx950 without a GPU
    header = open(os.path.join(ROOT, "include", "qdiff_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(qd_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 12
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in qdiff_hip.h but not exported"
    from qdiff import hip
    assert sorted(hip.EXPORTS) == declared
    lib.qd_abi_version.restype = ctypes.c_int
    assert lib.qd_abi_version() == 20
    assert lib.qd_device_ok() in (0, 1)                  # no compute calls without a GPU


def test_conv_desc_layout_matches_header():
    """ctypes mirror of qd_conv_desc / qd_conv_seg has the C layout (sizes from the header's field list)."""
    from qdiff import hip
    assert ctypes.sizeof(hip.ConvSeg) == 4 * 4 + 6 * 8
    assert ctypes.sizeof(hip.ConvDesc) == 6 * 8 + 5 * 8 + 16 * 4 + 2 * ctypes.sizeof(hip.ConvSeg) + 8 + 4 * 4 + 16 + 5 * 4 + 4 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(hip.RawSeg) == 6 * 4 + 8 and ctypes.sizeof(hip.RawQuant) == 8 + 8 + 4 + 4 + 2 * ctypes.sizeof(hip.RawSeg)


def test_conv_desc_offsets_match_the_compiled_header(tmp_path):
    """Compile include/qdiff_hip.h with gcc and compare sizeof/offsetof of every field with the ctypes mirror."""
    import shutil
    import subprocess
    from qdiff import hip
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "qdiff_hip.h"', 'int main(void){',
             'printf("qd_conv_seg %zu\\n", sizeof(qd_conv_seg));', 'printf("qd_conv_desc %zu\\n", sizeof(qd_conv_desc));']
    for name, _ in hip.ConvSeg._fields_:
        lines.append(f'printf("seg.{name} %zu\\n", offsetof(qd_conv_seg, {name}));')
    for name, _ in hip.ConvDesc._fields_:
        lines.append(f'printf("desc.{name} %zu\\n", offsetof(qd_conv_desc, {name}));')
    lines += ['printf("qd_raw_seg %zu\\n", sizeof(qd_raw_seg));', 'printf("qd_raw_quant %zu\\n", sizeof(qd_raw_quant));']
    for name, _ in hip.RawSeg._fields_:
        lines.append(f'printf("rseg.{name} %zu\\n", offsetof(qd_raw_seg, {name}));')
    for name, _ in hip.RawQuant._fields_:
        lines.append(f'printf("raw.{name} %zu\\n", offsetof(qd_raw_quant, {name}));')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["qd_conv_seg"]) == ctypes.sizeof(hip.ConvSeg)
    assert int(got["qd_conv_desc"]) == ctypes.sizeof(hip.ConvDesc)
    for name, _ in hip.ConvSeg._fields_:
        assert int(got[f"seg.{name}"]) == getattr(hip.ConvSeg, name).offset, name
    for name, _ in hip.ConvDesc._fields_:
        assert int(got[f"desc.{name}"]) == getattr(hip.ConvDesc, name).offset, name
    assert int(got["qd_raw_seg"]) == ctypes.sizeof(hip.RawSeg) and int(got["qd_raw_quant"]) == ctypes.sizeof(hip.RawQuant)
    for name, _ in hip.RawSeg._fields_:
        assert int(got[f"rseg.{name}"]) == getattr(hip.RawSeg, name).offset, name
    for name, _ in hip.RawQuant._fields_:
        assert int(got[f"raw.{name}"]) == getattr(hip.RawQuant, name).offset, name


def test_ctypes_signatures_match_the_header_prototypes():
    """Every `argtypes` list of qdiff/hip.py against the prototype include/qdiff_hip.h declares for that symbol: the same number
    of parameters, each of the same kind (pointer / 32-bit int / 64-bit int / float) in the same position — a shifted argument
    in a ctypes call corrupts silently, and no CPU test would otherwise see it."""
    import ctypes
    import re
    from qdiff import hip
    lib = hip.load()
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qdiff_hip.h")).read(), flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|const char\*|void)\s+(qd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S)
    assert len(protos) == len(hip.EXPORTS) and {n for n, _ in protos} == set(hip.EXPORTS)

    def header_kind(t):
        t = t.strip()
        if t in ("void", ""):
            return None
        if "*" in t:
            return "ptr"
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "float": "f32"}[t.replace("const", "").split()[0]]

    def ctypes_kind(c):
        if c in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(c, type) and issubclass(c, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_int64: "i64", ctypes.c_long: "i64", ctypes.c_float: "f32"}[c]

    for name, args in protos:
        want = [k for k in (header_kind(a) for a in args.split(",")) if k]
        at = getattr(lib, name).argtypes
        if at is None:
            assert not want, f"{name}: the header declares {len(want)} parameters, hip.py sets no argtypes"
            continue
        assert [ctypes_kind(c) for c in at] == want, name


def test_modulated_groupnorm_entry_validates_its_arguments_on_the_host():
    """qd_groupnorm_mod_silu_quant (ABI v15): the argument checks that precede any device work are reachable without a GPU and
    tell that `mod` / `mod_ld` arrive in the positions the ctypes binding puts them (fake non-null pointers are never
    dereferenced on this path)."""
    from qdiff import hip
    lib = hip.load()
    args = [1, 0, 1, 8, 16, 16, 4, 1e-5, None, None, 1, None, 0, 0, 0, None, 0, 8, 16, 16, None, 0, 0]
    assert lib.qd_groupnorm_mod_silu_quant(*args, 24, 5, None) != 0
    assert "mod_ld >= 2 C" in lib.qd_last_error().decode()
    assert lib.qd_groupnorm_mod_silu_quant(*args, None, 64, None) != 0
    assert "null modulation rows" in lib.qd_last_error().decode()


def test_integer_path_refuses_to_run_on_the_host():
    """No silent fallback: with (True, True) quantisation and CPU tensors the engine raises."""
    import qdiff
    from qdiff.hip import HipEngineError
    m = qdiff.QuantModule(torch.nn.Conv2d(16, 16, 3, padding=1), dict(n_bits=8, channel_wise=True, scale_method="max"),
                          dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True))
    m.set_quant_state(True, True)
    with torch.no_grad(), pytest.raises(HipEngineError):
        m(torch.randn(1, 16, 8, 8))


# ------------------------------------------------------------------------------------------------
# host logic on the emulator
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def emu(monkeypatch):
    abi_emulator.install(monkeypatch)


def _resume_cpu(fx):
    import qdiff
    from qdiff.utils import resume_cali_model
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    cal = tuple(a for a in fixture_inputs(fx, "cal") if a is not None)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt.pth")
        torch.save(build_ckpt(fx), path)
        resume_cali_model(qnn, path, cal, quant_act=True, cond=spec["ctx"] is not None)
    return qnn


def test_quant_module_kinds_on_emulator(emu):
    """conv2d / strided asym-pad conv / split 1x1 / conv1d / linear through QuantModule vs the real
    reference's outputs (ops.pt 'modules'); tolerance 2e-5 of range (integer vs fp32 accumulation)."""
    import torch.nn as nn
    import qdiff
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    ops = load_fixture("ops.pt")
    for c in ops["modules"]:
        w = c["weight"]
        if c["kind"] == "conv2d":
            org = nn.Conv2d(w.shape[1], w.shape[0], w.shape[2], stride=c["kw"]["stride"], padding=c["kw"]["padding"])
        elif c["kind"] == "conv1d":
            org = nn.Conv1d(w.shape[1], w.shape[0], 1)
        else:
            org = nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            org.weight.copy_(w)
            org.bias.copy_(c["bias"])
        m = qdiff.QuantModule(org, dict(n_bits=c["w_bits"], channel_wise=True, scale_method="max"),
                              dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True, symmetric=c["a_sym"]))
        m.set_quant_state(True, True)
        x = torch.nn.functional.pad(c["x"], (0, 1, 0, 1)) if c.get("asym_pad") else c["x"]
        with torch.no_grad():
            y0 = m(x, split=c["split"]) if c["split"] else m(x)          # data-dependent init, as the reference
            assert (y0 - c["y_uniform"]).abs().max() <= 2e-5 * c["y_uniform"].abs().max()
            # the initialisation reproduced the reference's scales exactly
            aqs = [m.act_quantizer] + ([m.act_quantizer_0] if c["split"] else [])
            for q, d, z in zip(aqs, c["a_delta"], c["a_zp"]):
                assert torch.equal(q.delta.detach(), d) and float(q.zero_point) == float(z)
            wqs = [m.weight_quantizer] + ([m.weight_quantizer_0] if c["split"] else [])
            for q, d, z in zip(wqs, c["w_delta"], c["w_zp"]):
                assert torch.equal(q.delta, d) and torch.equal(q.zero_point, z)
            slices = [m.org_weight] if not c["split"] else [m.org_weight[:, :c["split"]], m.org_weight[:, c["split"]:]]
            names = ["weight_quantizer", "weight_quantizer_0"]
            for nm, q, ws, al in zip(names, wqs, slices, c["alphas"]):
                ada = AdaRoundQuantizer(q, ws, "learned_hard_sigmoid")
                ada.alpha.data.copy_(al)
                setattr(m, nm, ada)
            y = m(x)
        assert (y - c["y"]).abs().max() <= 2e-5 * c["y"].abs().max(), (c["kind"], c["w_bits"], c["a_sym"], c["split"])


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_tiny_unets_on_emulator(emu, name):
    """resume_cali_model + fused integer blocks end to end on CPU.  Bound: the reference's own
    fp32-vs-fp64 envelope — max|diff| <= 0.1 * range and cosine >= 0.998."""
    import qdiff
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume_cpu(fx)
    mods = [m for m in qnn.modules() if isinstance(m, qdiff.QuantModule)]
    assert len(mods) == fx["n_quant_modules"] and all(m.int_ready() for m in mods)
    x, t, c = fixture_inputs(fx, "test")
    with torch.no_grad():
        y = qnn(x, t, c) if c is not None else qnn(x, t)
    assert all(m._plan is not None for m in mods)
    ref = fx["out_wa"]
    d = (y - ref).abs().max().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(y.flatten(), ref.flatten(), dim=0).item()
    assert d <= 0.1 and cos >= 0.998, (d, cos)
    # weights-only and fp states run the fp32 library path on the host and match tightly
    for state, key in (((True, False), "out_w"), ((False, False), "out_fp")):
        qnn.set_quant_state(*state)
        with torch.no_grad():
            y = qnn(x, t, c) if c is not None else qnn(x, t)
        assert (y - fx[key]).abs().max() <= 1e-4 * fx[key].abs().max()


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_checkpoint_schema_and_resume_types(emu, name):
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    from qdiff.utils import export_cali_state_dict
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume_cpu(fx)
    sd = export_cali_state_dict(qnn)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(s) for k, s in fx["keys"]}
    ck = build_ckpt(fx)
    assert all(torch.equal(sd[k].float(), ck[k].float()) for k in ck)
    n_split = 0
    for m in qnn.modules():
        if isinstance(m, AdaRoundQuantizer):
            assert torch.is_tensor(m.delta) and not isinstance(m.delta, torch.nn.Parameter)
            assert not isinstance(m.zero_point, torch.nn.Parameter)
        elif isinstance(m, UniformAffineQuantizer) and m.inited:
            assert isinstance(m.zero_point, int) and isinstance(m.delta, torch.nn.Parameter)
        if getattr(m, "split", 0):
            n_split += 1
    assert (n_split > 0) == bool(fx["spec"]["split"]), "split shortcut engaged / not engaged against the spec"


def test_plan_cache_tracks_quantiser_changes(emu):
    """Packed weights / epilogue constants are rebuilt when delta, zero_point or alpha change —
    by re-assignment or in place (reference utils.py:397-457 does both)."""
    import qdiff
    m = qdiff.QuantModule(torch.nn.Conv2d(16, 8, 1), dict(n_bits=8, channel_wise=True, scale_method="max"),
                          dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True))
    m.set_quant_state(True, True)
    x = torch.randn(2, 16, 4, 4)
    with torch.no_grad():
        y0 = m(x)
        p0 = m._plan
        assert m(x) is not None and m._plan is p0               # cached
        m.act_quantizer.delta.mul_(2.0)                          # in-place change (version counter bumps)
        y1 = m(x)
        assert m._plan is not p0 and not torch.equal(y0, y1)
        p1 = m._plan
        m.act_quantizer.delta.data.mul_(0.5)                     # .data writes are invisible to autograd versions ...
        m.invalidate()                                           # ... so the documented hook is used
        assert torch.equal(m(x), y0) and m._plan is not p1
        pk = m._pack
        m.weight_quantizer.delta = m.weight_quantizer.delta * 1.5   # re-assignment
        m(x)
        assert m._pack is not pk
        m.act_quantizer.zero_point = int(m.act_quantizer.zero_point) + 3
        p2 = m._plan
        m(x)
        assert m._plan is not p2


def test_packing_mode_selection(emu):
    from types import SimpleNamespace as NS
    from qdiff import engine
    w = torch.randn(8, 16, 1, 1)
    mk = lambda bits, zp: NS(delta=torch.full((8, 1, 1, 1), 0.1), zero_point=torch.full((8, 1, 1, 1), float(zp)), n_levels=2 ** bits)
    assert engine.pack_module_weights(w, [mk(4, 7)], 0).mode == 4          # int4 nibbles
    assert engine.pack_module_weights(w, [mk(8, 131)], 0).mode == 8        # u8 codes: W-128 + row-sum correction
    pk = engine.pack_module_weights(w, [mk(6, 30)], 0)
    assert pk.mode == 8 and pk.tiled and pk.wbits == 8                    # tile order always stores W-128
    assert engine.pack_module_weights(w, [mk(4, 200)], 0).mode == 4        # odd zero point: still nibbles (zw lives in the epilogue)
    from qdiff import hip
    with pytest.raises(hip.HipEngineError):                                # outside the epilogue's int32 budget
        engine.pack_module_weights(w, [mk(8, 700)], 0)
    sym = mk(8, 0)
    sym.sym, sym.n_levels = True, 127
    with pytest.raises(hip.HipEngineError):                                # signed weight grids are not packed
        engine.pack_module_weights(w, [sym], 0)


def test_activation_zero_point_outside_the_stored_byte_raises(emu):
    """An activation zero point whose stored form z' = zp - off leaves int8 (possible after EMA range updates with
    x_min > 0, or a learned delta) would wrap silently in the quantiser kernels: the plan builder refuses it."""
    from types import SimpleNamespace as NS
    from qdiff import engine, hip
    w = torch.randn(8, 16, 1, 1)
    wq = NS(delta=torch.full((8, 1, 1, 1), 0.1), zero_point=torch.full((8, 1, 1, 1), 7.0), n_levels=16)
    pack = engine.pack_module_weights(w, [wq], 0)
    ok = NS(delta=torch.tensor(0.05), zero_point=255, n_bits=8, sym=False)
    engine.build_conv_plan(pack, [ok], 1, 1, 1, 0, None)
    for zp in (-1, 256, torch.tensor(300.0)):
        bad = NS(delta=torch.tensor(0.05), zero_point=zp, n_bits=8, sym=False)
        with pytest.raises(hip.HipEngineError):
            engine.build_conv_plan(pack, [bad], 1, 1, 1, 0, None)


def _standalone_matmul_modules(c):
    """QuantQKMatMul / QuantSMVMatMul with the quantiser state of the reference's golden case."""
    import qdiff
    from qdiff.quant_block import QuantQKMatMul, QuantSMVMatMul
    aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True, symmetric=True)
    qk, smv = QuantQKMatMul(aq), QuantSMVMatMul(aq, sm_abit=8)
    qk.scale = c["scale"]
    for mod, table in ((qk, c["qq"]), (smv, c["qs"])):
        for n, (d, z) in table.items():
            qz = getattr(mod, n)
            qz.delta = torch.nn.Parameter(d.clone()) if torch.is_tensor(d) else d
            qz.zero_point = z
            qz.inited = True
    qk.use_act_quant = smv.use_act_quant = True
    return qk, smv


def test_standalone_qk_smv_matmuls_on_emulator(emu):
    """The two attention matmul modules used on their own (reference quant_block.py:114-160) take the integer
    engine too; vs the real reference's outputs (ops.pt 'attention' / ldm_qk_smv), 2e-5 of range."""
    ops = load_fixture("ops.pt")
    c = [a for a in ops["attention"] if a["kind"] == "ldm_qk_smv"][0]
    qk, smv = _standalone_matmul_modules(c)
    with torch.no_grad():
        w = qk(c["q"], c["k"])
        assert (w - c["weight"]).abs().max() <= 2e-5 * c["weight"].abs().max()
        a = smv(torch.softmax(c["weight"].float(), dim=-1), c["v"])
    assert (a - c["out"]).abs().max() <= 2e-5 * c["out"].abs().max()


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_packed_checkpoint_round_trip(emu, name, tmp_path):
    """export_packed_ckpt -> load_packed_ckpt into a model with DIFFERENT fp32 weights reproduces the original
    integer-path output bit for bit, the fp32 weights of the quantised layers are released, and the file is several
    times smaller than the reference-format state dict."""
    import qdiff
    from qdiff.utils import export_cali_state_dict, load_packed_ckpt, save_packed_ckpt
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume_cpu(fx)
    x, t, c = fixture_inputs(fx, "test")
    run = lambda m: m(x, t, c) if c is not None else m(x, t)
    with torch.no_grad():
        y0 = run(qnn)
    path = tmp_path / "packed.pth"
    save_packed_ckpt(qnn, path)
    ref_bytes = sum(v.numel() * v.element_size() for v in export_cali_state_dict(qnn).values())
    assert os.path.getsize(path) < 0.45 * ref_bytes

    spec = fx["spec"]
    wq, aq = quant_params(spec)
    other = build_engine_model(spec)
    with torch.no_grad():
        for p in other.parameters():
            p.add_(torch.randn_like(p) * 0.5)                      # nothing of the fp32 weights may matter
    q2 = qdiff.QuantModel(other, wq, aq, sm_abit=spec["sm_abit"]).eval()
    load_packed_ckpt(q2, str(path))
    with torch.no_grad():
        y1 = run(q2)
    assert torch.equal(y0, y1)
    mods = [m for m in q2.modules() if isinstance(m, qdiff.QuantModule)]
    assert mods and all(m.weight.numel() == 0 for m in mods)


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_blocks_teacher_forced_on_emulator(emu, name):
    """The teacher-forced per-block harness of tests/test_block_parity.py (GPU, full shapes) on the CPU ABI emulator:
    every fused block wiring of qdiff/quant_block.py fed with the oracle's block inputs reproduces the oracle's block
    outputs within the per-kind bounds."""
    from block_parity_util import run_block_parity
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume_cpu(fx)
    lines, failures = run_block_parity(qnn, fx, torch.device("cpu"))
    print("\n" + "\n".join(lines))
    assert lines[0].endswith("0.00e+00 of range"), "oracle trace run is not the reference run"
    assert not failures, "\n".join(failures)


def test_transformer_sublayers_teacher_forced_on_emulator(emu):
    """The sub-layer harness of tests/test_block_parity.py (attn1 / attn2 / ff of every transformer block fed the oracle's
    sub-layer input; the attention judged against the exact-integer oracle, the to_out code flips counted) on the ABI
    emulator, sd_tiny."""
    from block_parity_util import run_sublayer_parity
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume_cpu(fx)
    lines, failures = run_sublayer_parity(qnn, fx, torch.device("cpu"))
    print("\n" + "\n".join(lines))
    assert len(lines) >= 5 and not failures, "\n".join(failures)


def test_churches_unet_on_emulator(emu):
    """LSUN-Churches LDM-8 at its real shapes (models/ldm/lsun_churches256/config.yaml: 4 x 32 x 32 latents, 35 residual
    blocks that all modulate their second norm, 4 + 4 of them resampling, 21 eight-head attention blocks with head dims
    24 / 48 / 96 on 1024 .. 4 tokens) through the host logic and the ABI emulator: the whole evaluation inside the reference's
    own fp32-vs-fp64 envelope, and every block teacher-forced with the oracle's inputs within the per-kind bounds."""
    import qdiff
    from block_parity_util import run_block_parity
    fx = load_fixture("model_churches_full.pt")
    qnn = _resume_cpu(fx)
    mods = [m for m in qnn.modules() if isinstance(m, qdiff.QuantModule)]
    assert len(mods) == fx["n_quant_modules"] and all(m.int_ready() for m in mods)
    x, t, _ = fixture_inputs(fx, "test")
    with torch.no_grad():
        y = qnn(x, t)
    assert all(m._plan is not None for m in mods)
    ref = fx["out_wa"]
    d = (y - ref).abs().max().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(y.flatten(), ref.flatten(), dim=0).item()
    assert d <= 0.1 and cos >= 0.998, (d, cos)
    lines, failures = run_block_parity(qnn, fx, torch.device("cpu"))
    print("\n" + "\n".join(lines))
    assert lines[0].endswith("0.00e+00 of range"), "oracle trace run is not the reference run"
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("name", ["sd_tiny", "cifar_tiny"])
def test_running_stat_updates_match_the_simulation_path(emu, name):
    """QuantModel.set_running_stat(True) (the reference's calibration loop, txt2img.py:457-468) in (True, True) state:
    the fused blocks must not freeze their first-batch ranges — every activation quantiser (QuantModules inside blocks
    and the attention q/k/v/w quantisers) ends up with the same EMA-updated delta / zero_point as the pure simulation."""
    import copy
    import qdiff
    from qdiff import engine
    from qdiff.quant_layer import UniformAffineQuantizer
    fx = load_fixture(f"model_{name}.pt")
    qa = _resume_cpu(fx)
    qb = copy.deepcopy(qa)
    x, t, c = fixture_inputs(fx, "test")
    args = (x * 1.7 + 0.3, t) + ((c,) if c is not None else ())      # a batch with a different range than calibration

    def track(q):
        q.set_running_stat(True)
        for m in q.modules():                                         # attention quantisers of every block flavour
            if isinstance(m, UniformAffineQuantizer) and m.leaf_param and m.inited:
                m.running_stat = True
    track(qa)
    track(qb)
    with torch.no_grad():
        qa(*args)                                                     # engine wiring (fused blocks must step aside)
        engine.SIMULATE = True
        try:
            qb(*args)                                                 # pure simulation
        finally:
            engine.SIMULATE = False
    changed = 0
    for (na, ma), (nb, mb) in zip(qa.named_modules(), qb.named_modules()):
        if isinstance(ma, UniformAffineQuantizer) and ma.inited and ma.leaf_param and ma.running_stat:
            # standalone QuantModules still contract on the integer engine (exact accumulators) while tracking, so tensors
            # further down differ from the simulation's by fp32 rounding and the occasional code flip it causes: the tracked
            # ranges agree to within 1e-2 (softmax maxima are the touchiest), not bitwise (the EMA moves them by percents, checked below)
            da, db = float(torch.as_tensor(ma.delta.detach())), float(torch.as_tensor(mb.delta.detach()))
            assert abs(da - db) <= 1e-2 * abs(db), (na, da, db)
            assert abs(float(ma.zero_point) - float(mb.zero_point)) <= 1.0, na
            changed += 1
    assert changed > 20
    # and the ranges did move away from the calibrated ones
    q0 = _resume_cpu(fx)
    moved = sum(1 for (n, m), (_, m0) in zip(qa.named_modules(), q0.named_modules())
                if isinstance(m, UniformAffineQuantizer) and m.inited and m.leaf_param and m.delta.shape == torch.Size([])
                and abs(float(m.delta) - float(m0.delta)) > 5e-3 * float(m0.delta))
    assert moved > 20


def test_prepared_context_skips_the_chain_and_changes_nothing(emu, monkeypatch):
    """Cross-attention K / V^T operands once per sampling run instead of once per evaluation (the reference recomputes to_k /
    to_v of the constant conditioning in every evaluation, quant_block.py:193-195).  The evaluation of a prepared context
    issues NO to_k / to_v GEMM and no head-layout quantiser, hands the attention kernel a key-term table that belongs to the
    prepared keys (the emulator asserts it), and returns the unprepared output bit for bit.  Round 5: a context is recognised by
    VALUE — a fresh tensor with the prepared bytes (plms.py:184-187 builds one per step) — two contexts stay prepared at a time
    (least recently used evicted), and forward() prepares an unseen context by itself (QDIFF_CTX_AUTO), so that the unmodified
    reference samplers run the chain once per run.  Changed bytes, a quant-state flip and QDIFF_CTX_PIN=0 fall back to the
    per-evaluation branch."""
    from qdiff import hip, quant_block as qb, sampling
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume_cpu(fx)
    x, t, c = fixture_inputs(fx, "test")
    g = torch.Generator().manual_seed(3)
    c2, c3 = torch.randn(c.shape, generator=g), torch.randn(c.shape, generator=g)
    monkeypatch.setattr(qb, "_CTX_AUTO", False)           # first: explicit preparation only
    with torch.no_grad():
        want, want2, want3 = qnn(x, t, c), qnn(x, t, c2), qnn(x, t, c3)
        want2x = qnn(x, t, c2 * 2)
    ckv = qnn.__dict__["_ctx_kv"]
    assert not ckv._pins
    calls = {"heads": 0, "kterm_given": 0}
    real_qh, real_attn = hip.quantize_heads, hip.attn_i8

    def counting_qh(*a, **kw):
        calls["heads"] += 1
        return real_qh(*a, **kw)

    def counting_attn(*a, **kw):
        calls["kterm_given"] += kw.get("kterm") is not None
        return real_attn(*a, **kw)

    monkeypatch.setattr(hip, "quantize_heads", counting_qh)
    monkeypatch.setattr(hip, "attn_i8", counting_attn)
    # the library takes a table only on long key axes (LDS-staged kernel); here every eligible head dim does, so that the
    # prepared tables travel through the host code and reach the emulator's staleness check
    monkeypatch.setattr(hip, "attn_uses_keyterm", lambda d, S, asym: bool(asym) and d < 64 and d % 32 != 0)
    nblk = sum(isinstance(m, qb.QuantBasicTransformerBlock) for m in qnn.modules())

    def run(cc, expect):
        calls.update(heads=0, kterm_given=0)
        y = qnn(x, t, cc)
        assert torch.equal(y, expect)
        return calls["heads"]

    with torch.no_grad():
        per_eval = run(c, want)
        assert per_eval >= 2 * nblk                       # k and v^T of every cross-attention (+ ragged self-attention operands)
        prepared = per_eval - 2 * nblk                    # what an evaluation issues when the context chain does not run
        assert qnn.prepare_context(c) is True
        assert run(c, want) == prepared
        assert 0 < calls["kterm_given"] <= nblk, calls      # cross-attentions of eligible head dims got their prepared key-term tables
        # a FRESH tensor holding the prepared bytes — what the reference's samplers hand over at every step
        vm = ckv.value_matches
        assert run(c.clone(), want) == prepared and ckv.value_matches == vm + 1
        assert run(torch.cat([c[:1], c[1:]]), want) == prepared and ckv.value_matches == vm + 2
        # a second context takes the second slot; both stay prepared
        assert qnn.prepare_context(c2) is True
        assert run(c2, want2) == prepared and run(c, want) == prepared and len(ckv._pins) == 2
        assert {e["slot"] for e in ckv._pins} == {0, 1}
        # a third one evicts the least recently used (c2: c was used last)
        assert qnn.prepare_context(c3) is True
        assert run(c3, want3) == prepared and run(c, want) == prepared and len(ckv._pins) == 2
        assert run(c2, want2) == per_eval                  # no longer prepared, automatic preparation is off: per-evaluation branch
        # in-place edits of a prepared tensor: its version moved, so the bytes decide
        c.add_(0.0)
        assert run(c, want) == prepared                    # same bytes
        c2m = c2.clone()
        assert qnn.prepare_context(c2m) is True and run(c2m, want2) == prepared
        c2m.mul_(2)
        assert run(c2m, want2x) == per_eval                # other bytes: not prepared
        # any state change drops the prepared contexts (their bytes were made by the old plans)
        qnn.prepare_context(c)
        qnn.set_quant_state(True, True)
        assert not ckv._pins and run(c, want) == per_eval
        monkeypatch.setattr(qb, "_CTX_PIN", False)
        assert qnn.prepare_context(c) is False and run(c, want) == per_eval
        monkeypatch.setattr(qb, "_CTX_PIN", True)
        # a prepared context handed to a latent batch it was not made for is refused, not read out of bounds
        assert qnn.prepare_context(c) is True
        with pytest.raises(hip.HipEngineError):
            qnn(torch.cat([x, x]), torch.cat([t, t]), c)
        qnn.release_context()
        assert not ckv._pins
        # ---- automatic preparation (the default): the model prepares what it has not seen -----------------------------------
        monkeypatch.setattr(qb, "_CTX_AUTO", True)
        runs = ckv.chain_runs
        assert run(c, want) == per_eval and ckv.chain_runs == runs + 1          # first sight: the chain ran once, into a slot
        assert run(c.clone(), want) == prepared and ckv.chain_runs == runs + 1
        assert run(c2, want2) == per_eval and run(c2.clone(), want2) == prepared and run(c, want) == prepared
        assert ckv.chain_runs == runs + 2
        qnn.release_context()
        # the reference's sampler loop, as plms.py:176-190 writes it: fresh concatenations at every step, no announcement
        table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 4, eta=0.0)
        uc = torch.randn(c.shape, generator=g)
        monkeypatch.setattr(qb, "_CTX_AUTO", False)
        unet = lambda xx, tt, cc=None: qnn(xx, tt, cc)
        ref = sampling.plms_sample(unet, x, table, cond=c, uncond=uc, scale=3.0)            # every evaluation runs the chain
        monkeypatch.setattr(qb, "_CTX_AUTO", True)

        def as_script(xx, tt, cc=None):                   # guided_eps hands over its own ctx2: rebuild it as the reference does
            return qnn(xx, tt, torch.cat([uc, c]))
        calls["heads"] = 0
        got = sampling.plms_sample(as_script, x, table, cond=c, uncond=uc, scale=3.0)
        assert torch.equal(got, ref)
        assert calls["heads"] == 5 * prepared + 2 * nblk, calls                            # 4 steps = 5 evaluations, ONE context chain
        qnn.release_context()
        # the samplers of this package announce the run's conditioning themselves
        calls["heads"] = 0
        got = sampling.plms_sample(qnn, x, table, cond=c, uncond=uc, scale=3.0)
        assert torch.equal(got, ref)
        assert calls["heads"] == 5 * prepared + 2 * nblk, calls


def test_ldm_attention_block_qkv_as_three_operand_projections(emu, monkeypatch):
    """The LDM AttentionBlock's fused qkv conv1d (output channels [head][q | k | v][d], reference quant_block.py:163-187) runs as
    three GEMMs over row subsets of its weight whose epilogues write the attention operand bytes (QuantModule.head_plans,
    quant_block.QuantAttentionBlock._forward_heads): per-output-channel weight quantisers make the subsets exact, so the
    UNet output is the one of the fp32-round-trip route bit for bit; the route is taken where the token count allows it
    (256 tokens at the first level of the tiny model, not the 64 of the second) and not after a packed checkpoint froze the
    layer's weights."""
    from qdiff import engine, hip, quant_block
    fx = load_fixture("model_ldm_tiny.pt")
    qnn = _resume_cpu(fx)
    x, t, _ = fixture_inputs(fx, "test")
    calls = {"heads": 0, "float": 0}
    real_conv, real_qh = hip.conv2d_i8, hip.quantize_heads

    def counting_conv(cc, acc_out=None):
        calls["heads"] += cc.epilogue in (hip.EPI_HEADS_I8, hip.EPI_HEADS_T_I8) and cc.heads["H"] > 1
        return real_conv(cc, acc_out)

    def counting_qh(*a, **k):
        calls["float"] += 1
        return real_qh(*a, **k)
    monkeypatch.setattr(hip, "conv2d_i8", counting_conv)
    monkeypatch.setattr(hip, "quantize_heads", counting_qh)
    blocks = [m for m in qnn.modules() if isinstance(m, quant_block.QuantAttentionBlock)]
    assert blocks
    monkeypatch.setattr(quant_block, "QKV_HEADS", False)
    with torch.no_grad():
        want = qnn(x, t)
    assert calls["heads"] == 0 and calls["float"] == 3 * len(blocks)
    monkeypatch.setattr(quant_block, "QKV_HEADS", True)
    calls.update(heads=0, float=0)
    with torch.no_grad():
        got = qnn(x, t)
    fused = [b for b in blocks if b.qkv.__dict__.get("_heads_cache", [None, None])[1] is not None]
    assert 0 < len(fused) < len(blocks)
    assert calls["heads"] == 3 * len(fused) and calls["float"] == 3 * (len(blocks) - len(fused)), calls
    assert torch.equal(got, want)
    # the three plans hold exactly the rows of their role, head-major
    b = fused[0]
    nh, C = b.num_heads, b.channels
    d = C // nh
    for role, plan in enumerate(b.qkv.head_plans(nh)):
        rows = torch.cat([torch.arange(d) + h * 3 * d + role * d for h in range(nh)])
        assert torch.equal(plan.pack.row_perm.cpu(), rows) and plan.Cout == C
        assert torch.equal(plan.bias.cpu(), b.qkv.bias.detach().float()[rows])
    # a re-assigned quantiser rebuilds them
    before = b.qkv.head_plans(nh)
    b.qkv.act_quantizer.delta = torch.nn.Parameter(b.qkv.act_quantizer.delta.detach() * 1.5)
    assert b.qkv.head_plans(nh) is not before


def test_ddim_attn_block_operand_projections_with_int8_weights(emu, monkeypatch):
    """The pixel-space DDIM AttnBlock (reference quant_block.py:354-386; CIFAR W8A8): q / k / v are 1x1 convolutions with int8
    weights; where the token count is a multiple of 128 and the layer has more than 64 channels their epilogues write the
    attention operand bytes (one head as wide as the layer) and the attention epilogue quantises for proj_out — the block's
    output is the one of the fp32-projection route bit for bit."""
    import qdiff
    from qdiff import hip, quant_block
    from qdiff.arch import ddim_unet

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.in_channels = 128
            self.attn = ddim_unet.AttnBlock(128)

        def forward(self, x, t=None, c=None):
            return self.attn(x)
    torch.manual_seed(11)
    net = Net()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    wq = dict(n_bits=8, channel_wise=True, scale_method="max")
    aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)
    qnn = qdiff.QuantModel(net, wq, aq, sm_abit=8).eval()
    assert isinstance(qnn.model.attn, quant_block.QuantAttnBlock)
    x = torch.randn(2, 128, 16, 8)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        qnn(x)                                            # initialises every quantiser (fp32 simulation)
    calls = {"heads": 0, "float": 0}
    real_conv, real_qh = hip.conv2d_i8, hip.quantize_heads

    def counting_conv(cc, acc_out=None):
        calls["heads"] += cc.epilogue in (hip.EPI_HEADS_I8, hip.EPI_HEADS_T_I8) and cc.wbits == 8
        return real_conv(cc, acc_out)

    def counting_qh(*a, **k):
        calls["float"] += 1
        return real_qh(*a, **k)
    monkeypatch.setattr(hip, "conv2d_i8", counting_conv)
    monkeypatch.setattr(hip, "quantize_heads", counting_qh)
    x2 = torch.randn(2, 128, 16, 8)
    monkeypatch.setattr(quant_block, "QKV_HEADS", False)
    with torch.no_grad():
        want = qnn(x2)
    assert calls == {"heads": 0, "float": 3}
    monkeypatch.setattr(quant_block, "QKV_HEADS", True)
    calls.update(heads=0, float=0)
    with torch.no_grad():
        got = qnn(x2)
    assert calls == {"heads": 3, "float": 0}
    assert torch.equal(got, want)
    # 64 tokens (8 x 8): the fp32-projection route
    calls.update(heads=0, float=0)
    with torch.no_grad():
        qnn(torch.randn(2, 128, 8, 8))
    assert calls == {"heads": 0, "float": 3}


def test_head_plans_gathered_from_the_pack_equal_a_fresh_packing(emu):
    """Heads of a multiple of 32 channels (LDM-4: 32): the q / k / v operands are whole 32-row tiles of the layer's own pack
    (engine.pack_select_tiles) — the same bytes and per-row constants as packing the row subset from the fp32 weight, and
    available when a packed checkpoint froze the layer (the fp32 weight is gone then)."""
    import qdiff
    from qdiff import engine
    torch.manual_seed(3)
    heads, d, cin = 2, 32, 48
    m = qdiff.QuantModule(torch.nn.Conv1d(cin, 3 * heads * d, 1), dict(n_bits=4, channel_wise=True, scale_method="max"),
                          dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True))
    m.set_quant_state(True, True)
    m._init_act_quantizers(torch.randn(2, cin, 128))
    plans = m.head_plans(heads)
    assert plans is not None and len(plans) == 3
    for role, plan in enumerate(plans):
        rows = torch.cat([torch.arange(d) + h * 3 * d + role * d for h in range(heads)])
        fresh = engine.pack_module_weights(m.weight, [m.weight_quantizer], 0, row_perm=rows)
        assert plan.pack.Cout == fresh.Cout == heads * d and plan.pack.ldk == fresh.ldk
        assert torch.equal(plan.pack.wq, fresh.wq)
        for a, b in zip(plan.pack.segs, fresh.segs):
            assert all(a[k] == b[k] for k in engine._PACK_SEG_INTS if k in b)
            assert all(torch.equal(a[k], b[k]) for k in ("wsum", "delta_w", "zw"))
        assert torch.equal(plan.bias, m.bias.detach()[rows])
    # ragged selections are refused, frozen layers still serve tile-aligned heads
    assert engine.pack_select_tiles(m.conv_plan().pack, torch.arange(16)) is None
    assert engine.pack_select_tiles(m.conv_plan().pack, torch.arange(32) + 16) is None
    m.load_packed(engine.pack_from_dict(engine.pack_to_dict(m.conv_plan().pack), "cpu"))
    m.weight.data = torch.empty(0)
    frozen = m.head_plans(heads)
    assert frozen is not None and all(torch.equal(a.pack.wq, b.pack.wq) for a, b in zip(frozen, plans))
    assert m.head_plans(3) is None                        # 192 channels do not divide into 3 x 3 groups


def test_sampling_under_inference_mode_and_value_match_of_inference_tensors(emu):
    """tensors created under torch.inference_mode() carry no version counter (`t._version` raises); the prepared-
    context bookkeeping and the plan caches must not touch it.  An inference tensor is recognised by value only."""
    from qdiff import sampling
    fx = load_fixture("model_sd_tiny.pt")
    qnn = _resume_cpu(fx)
    x, t, c = fixture_inputs(fx, "test")
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.0120), 4, eta=0.0)
    with torch.no_grad():
        uc = torch.randn(c.shape, generator=torch.Generator().manual_seed(5))
        want = sampling.plms_sample(qnn, x, table, cond=c, uncond=uc, scale=2.0)
    qnn.release_context()
    ckv = qnn.__dict__["_ctx_kv"]
    with torch.inference_mode():
        ci, uci = c.clone(), uc.clone()                   # inference tensors
        with pytest.raises(RuntimeError):
            ci._version
        runs = ckv.chain_runs
        got = sampling.plms_sample(qnn, x.clone(), table, cond=ci, uncond=uci, scale=2.0)
        assert ckv.chain_runs == runs + 1                 # prepared once, recognised at every step (by value)
    assert torch.equal(got, want)


@pytest.mark.parametrize("name", ["cifar_tiny", "ldm_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_planned_concatenation_is_a_view_and_changes_nothing(emu, name, monkeypatch):
    """Skip concatenations (openaimodel.py:776, ddim diffusion.py:340) planned through engine.CatSlot + the skip
    connection's int8 rows taken from the GroupNorm pass (qd_raw_quant): from the second evaluation on no `cat` copy runs
    any more, and the output is that of the copying path bit for bit."""
    from qdiff import quant_block as qb
    fx = load_fixture(f"model_{name}.pt")
    qnn = _resume_cpu(fx)
    x, t, c = fixture_inputs(fx, "test")
    args = (x, t) + ((c,) if c is not None else ())
    with torch.no_grad():
        y0 = qnn(*args)                               # first evaluation: records the channel plan, concatenates by copy
    calls = {"cat": 0, "view": 0, "raw": 0}
    real_cat, real_adj, real_gn = torch.cat, qb._adjacent, qb.engine.groupnorm_silu_quant

    def counting_cat(ts, dim=0, **kw):
        if dim == 1 and len(ts) == 2 and ts[0].dim() == 4 and ts[0].is_floating_point():
            calls["cat"] += 1
        return real_cat(ts, dim=dim, **kw)

    def counting_adj(a, b, dim, unit):
        out = real_adj(a, b, dim, unit)
        if out is not None and dim == 1:
            calls["view"] += 1
        return out

    def counting_gn(*a, **kw):
        if kw.get("raw_plan") is not None:
            calls["raw"] += 1
        return real_gn(*a, **kw)

    monkeypatch.setattr(torch, "cat", counting_cat)
    monkeypatch.setattr(qb, "_adjacent", counting_adj)
    monkeypatch.setattr(qb.engine, "groupnorm_silu_quant", counting_gn)
    with torch.no_grad():
        y1 = qnn(*args)                               # planned: views
    n_cat = len(qnn.model.__dict__["_cat_plan"])
    assert calls["view"] == n_cat and calls["cat"] == 0, calls
    assert calls["raw"] > 0, "no skip connection took its rows from the GroupNorm pass"
    assert torch.equal(y0, y1)
    monkeypatch.setattr(qb, "CAT_SLOTS", False)
    monkeypatch.setattr(qb, "_FUSE_SKIP_QUANT", False)
    calls.update(cat=0, view=0, raw=0)
    with torch.no_grad():
        y2 = qnn(*args)
    assert calls["cat"] == n_cat and calls["view"] == 0 and calls["raw"] == 0, calls
    assert torch.equal(y1, y2)


def test_cat_channels_only_takes_the_view_when_it_is_one():
    """quant_block._adjacent must never call two tensors 'the two halves of one buffer' unless concatenating them IS the
    strided view it returns: every near-miss falls back to the copy."""
    from qdiff import engine, quant_block as qb
    B, H, W, C1, C2 = 2, 4, 4, 8, 4
    slot = engine.CatSlot(C1, C2)
    dev = torch.device("cpu")
    a = qb._rows_to_nchw(slot.rows(0, B * H * W, C1, dev), B, H, W)
    b = qb._rows_to_nchw(slot.rows(1, B * H * W, C2, dev), B, H, W)
    slot.buf.copy_(torch.arange(slot.buf.numel(), dtype=torch.float32).view_as(slot.buf))
    v = qb.cat_channels(a, b)
    assert v.data_ptr() == a.data_ptr() and torch.equal(v, torch.cat([a, b], dim=1))
    assert torch.equal(qb._nhwc_rows(v), slot.buf) and qb._nhwc_rows(v).data_ptr() == slot.buf.data_ptr()
    assert torch.equal(qb._nhwc_rows(b), slot.buf[:, C1:]) and qb._nhwc_rows(b).stride() == (C1 + C2, 1)
    # near misses: wrong order, a gap, different buffers, a clone of one half, mismatching shapes / dtypes
    assert qb._adjacent(b, a, 1, 1) is None
    wide = engine.CatSlot(C1, C2 + 4)
    wa = qb._rows_to_nchw(wide.rows(0, B * H * W, C1, dev), B, H, W)
    wb = qb._rows_to_nchw(wide.buf[:, C1 + 4:], B, H, W)                     # same row stride, but a 4-channel gap
    assert qb._adjacent(wa, wb, 1, 1) is None
    other = engine.CatSlot(C1, C2)
    ob = qb._rows_to_nchw(other.rows(1, B * H * W, C2, dev), B, H, W)
    other.rows(0, B * H * W, C1, dev)
    other.buf.fill_(3.0)                   # slot buffers are torch.empty: garbage (possibly NaN) would defeat torch.equal below
    wide.buf.fill_(5.0)
    assert qb._adjacent(a, ob, 1, 1) is None
    assert qb._adjacent(a, b.clone(), 1, 1) is None
    assert qb._adjacent(a[:1], b, 1, 1) is None and qb._adjacent(a, b.double(), 1, 1) is None
    # two plain contiguous NCHW tensors that happen to be neighbours in one allocation are NOT a channel concatenation
    flat = torch.arange(2 * B * C1 * H * W, dtype=torch.float32)
    p, q = flat[:B * C1 * H * W].view(B, C1, H, W), flat[B * C1 * H * W:].view(B, C1, H, W)
    assert qb._adjacent(p, q, 1, 1) is None
    for bad in ((b, a), (a, ob), (p, q)):
        assert torch.equal(qb.cat_channels(*bad), torch.cat(list(bad), dim=1))
    # a slot refuses producers of the wrong width / row count instead of mis-placing them
    assert slot.rows(0, B * H * W, C1 + 1, dev) is None and slot.rows(1, B * H * W + 1, C2, dev) is None


def test_bench_cpu_baseline_leg_runs_on_the_host(monkeypatch):
    """bench.py's `cpu_baseline` leg (the oracle timed on the host cores, thread-count sweep) only runs after the GPU part of the
    driver's command: a bug in it costs the whole bench line (round 3: a loop variable shadowed the timestep tensor).  Run it
    here on the CIFAR model built through the ABI emulator."""
    import abi_emulator
    abi_emulator.install(monkeypatch)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    qnn, qspec = bench.build_quantised_unet("cifar", torch.device("cpu"))
    ocfg = dict(ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16], resolution=32)
    dt, threads, sweep = bench.cpu_baseline(qnn, qspec, "cifar", ocfg, k=1)
    assert dt > 0 and threads >= 1 and str(threads) in sweep


def test_context_branch_fork_points(emu, monkeypatch):
    """Where the cross-attention K / V branch (ContextKV) forks off the main stream — QDIFF_CTX_FORK = late (at the first
    cross-attention, where the fork coincided with the join and the 148 launches of the branch ran alone:
    profiles/r03_sd_eval_timeline.tsv), start (the model's forward pre-hook, before the stem), attn (right before the first
    self-attention kernel) — changes the ORDER of the launches only: one branch per evaluation, the same output bit for bit."""
    from qdiff import engine
    from qdiff import quant_block as qb
    fx = load_fixture("model_sd_tiny.pt")
    x, t, c = fixture_inputs(fx, "test")
    outs, orders = {}, {}
    monkeypatch.setattr(qb, "_CTX_AUTO", False)           # the per-evaluation branch is the subject: no automatic preparation
    for mode in ("late", "start", "attn"):
        monkeypatch.setattr(qb, "_CTX_FORK", mode)
        qnn = _resume_cpu(fx)
        grp = next(m for m in qnn.modules() if isinstance(m, qb.QuantBasicTransformerBlock)).__dict__["_ctx_group"]
        events = []
        prepare0, conv0, attn0 = grp._prepare, engine.conv_forward, engine.attention_codes
        grp._prepare = lambda ctx, _p=prepare0: (events.append("fork" if ctx is c else "fork?"), _p(ctx), events.append("branch done"))[1]
        monkeypatch.setattr(engine, "conv_forward", lambda *a, **k: (events.append("gemm"), conv0(*a, **k))[1])
        monkeypatch.setattr(engine, "attention_codes", lambda *a, **k: (events.append("attention"), attn0(*a, **k))[1])
        with torch.no_grad():
            outs[mode] = qnn(x, t, c)
        monkeypatch.setattr(engine, "conv_forward", conv0)
        monkeypatch.setattr(engine, "attention_codes", attn0)
        assert events.count("fork") == 1 and "fork?" not in events, "one branch per evaluation, for the context the UNet was given"
        orders[mode] = events
    assert torch.equal(outs["late"], outs["start"]) and torch.equal(outs["late"], outs["attn"])
    assert orders["start"][0] == "fork", "before the stem"
    for mode, n_before in (("attn", 0), ("late", 1)):
        ev = orders[mode]
        i = ev.index("fork")
        assert ev[:i].count("attention") == n_before and "gemm" in ev[:i], (mode, ev[:i])
        assert ev[ev.index("branch done") + 1] == "attention"      # attn: the first self-attention kernel; late: the first cross-attention
