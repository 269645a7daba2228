"""Drop-in boundary on the REFERENCE's own model classes (reference scripts/txt2img.py:381-383, 490,
sample_diffusion_ldm.py, sample_diffusion_ddim.py: the scripts build the reference `UNetModel` / `Model`, wrap it with
`qdiff.QuantModel`, resume a calibrated checkpoint and sample).  Here this repo's `qdiff` wraps exactly those classes —
`ldm` / `ddim` are imported from /root/reference, `qdiff` resolves to this repo — and must reproduce the outputs the real
reference produced with ITS qdiff (tests/golden/model_*_tiny.pt):
  * (False, False) and weights-only (True, False): bit for bit on the CPU (same ATen calls on the same operands);
  * (True, True): the integer engine (on the CPU ABI emulator here) inside the tiny models' envelope, with every
    QuantModule on the integer path, the reference SpatialTransformer / Upsample / AttentionBlock running this engine's
    fused forwards and the GroupNorm statistics surviving the reference's `th.cat` of skip connections.
Build container only (the GPU box has no /root/reference): skipped when the reference tree is absent."""
import os
import sys
import tempfile
import types

import pytest
import torch

import abi_emulator
from golden_util import build_ckpt, fixture_inputs, load_fixture, quant_params

REF = "/root/reference"
sys.dont_write_bytecode = True          # the reference tree is read-only for this project: no __pycache__ next to its sources
if not os.path.isdir(os.path.join(REF, "ldm")):
    pytest.skip("reference tree not present (GPU box)", allow_module_level=True)

# `ldm` / `ddim` from the reference, `qdiff` from this repo (conftest put it first on sys.path)
if REF not in sys.path:
    sys.path.append(REF)
_oc, _lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")
_lc.ListConfig = type("ListConfig", (list,), {})
_oc.listconfig = _lc
sys.modules.setdefault("omegaconf", _oc)
sys.modules.setdefault("omegaconf.listconfig", _lc)


def _reference_fp_model(spec):
    from types import SimpleNamespace as NS
    from qdiff import synthetic
    if spec["family"] == "cifar":
        from ddim.models.diffusion import Model
        model = NS(type="simple", in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8],
                   dropout=0.1, resamp_with_conv=True)
        cfg = NS(model=model, data=NS(image_size=16, channels=3), diffusion=NS(num_diffusion_timesteps=1000),
                 split_shortcut=spec["split"])
        m = Model(cfg)
    else:
        from ldm.modules.diffusionmodules.openaimodel import UNetModel
        m = UNetModel(**spec["unet"])
        m.split = bool(spec["split"])
    m.load_state_dict(synthetic.fill_state_dict(m.state_dict(), seed=0))
    return m.eval()


def _wrap_and_resume(fx):
    import qdiff
    import qdiff.quant_model
    from qdiff.utils import resume_cali_model
    assert qdiff.quant_model.__file__.startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "wrong qdiff"
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    model = _reference_fp_model(spec)
    assert type(model).__module__.startswith(("ldm.", "ddim.")), "the model under test must be the reference's class"
    qnn = qdiff.QuantModel(model, wq, aq, sm_abit=spec["sm_abit"]).eval()
    cal = tuple(a for a in fixture_inputs(fx, "cal") if a is not None)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt.pth")
        torch.save(build_ckpt(fx), path)
        resume_cali_model(qnn, path, cal, quant_act=True, cond=spec["ctx"] is not None)
    return qnn


@pytest.mark.parametrize("name", ["sd_tiny", "ldm_tiny", "cifar_tiny", "ldm_updown_tiny"])
def test_reference_classes_wrapped_by_this_qdiff(monkeypatch, name):
    import qdiff
    from qdiff import hip, quant_block
    abi_emulator.install(monkeypatch)
    with_part = []
    gn_emul = hip.groupnorm_silu_quant

    def gn_spy(*a, **k):
        with_part.append(k.get("part") is not None)
        return gn_emul(*a, **k)
    monkeypatch.setattr(hip, "groupnorm_silu_quant", gn_spy)
    fx = load_fixture(f"model_{name}.pt")
    qnn = _wrap_and_resume(fx)
    mods = [m for m in qnn.modules() if isinstance(m, qdiff.QuantModule)]
    assert len(mods) == fx["n_quant_modules"] and all(m.int_ready() for m in mods)
    x, t, c = fixture_inputs(fx, "test")
    run = lambda: (qnn(x, t, c) if c is not None else qnn(x, t))
    with torch.no_grad():
        with_part.clear()
        y = run()
    assert all(m._plan is not None for m in mods), "a QuantModule of the reference model did not take the integer path"
    ref = fx["out_wa"]
    d = (y - ref).abs().max().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(y.flatten(), ref.flatten(), dim=0).item()
    assert d <= 0.1 and cos >= 0.998, (d, cos)
    blocks = [m for m in qnn.modules() if isinstance(m, quant_block.BaseQuantBlock)]
    assert blocks and all(type(b).__name__.startswith("Quant") for b in blocks)
    if fx["spec"]["family"] == "ldm":
        # the statistics written by the producing GEMMs reached the GroupNorms across the reference's th.cat
        # (the tiny maps are too small for 128-row chunks on some levels, hence "some", not "all")
        res = [b for b in blocks if isinstance(b, quant_block.QuantResBlock)]
        assert all(isinstance(b, quant_block.reference_classes()["TimestepBlock"]) for b in res)
        att = [b for b in blocks if isinstance(b, quant_block.QuantAttentionBlock)]
        if att:
            assert all(type(b.attention.qkv_matmul) is quant_block.QuantQKMatMul for b in att)
    # the floating-point states are the reference's own arithmetic: bit for bit
    for state, key in (((True, False), "out_w"), ((False, False), "out_fp")):
        qnn.set_quant_state(*state)
        with torch.no_grad():
            assert torch.equal(run(), fx[key]), (name, key)


def test_groupnorm_statistics_survive_the_reference_cat(monkeypatch):
    """SD-sized channel counts on a 16x16 map (>= 128 rows per sample): the decoder ResBlocks of the REFERENCE UNetModel —
    whose forward concatenates skip connections with plain th.cat (openaimodel.py:776) — receive the first-level GroupNorm
    statistics of both producers (QuantModel's shadow stack), i.e. their first GroupNorm skips its statistics pass."""
    import qdiff
    from qdiff import hip, synthetic
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    abi_emulator.install(monkeypatch)
    calls = []
    gn_emul = hip.groupnorm_silu_quant

    def gn_spy(x, B, S, C, *a, **k):
        calls.append((C, k.get("part") is not None))
        return gn_emul(x, B, S, C, *a, **k)
    monkeypatch.setattr(hip, "groupnorm_silu_quant", gn_spy)
    cfg = dict(image_size=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[], num_res_blocks=1,
               channel_mult=[1, 2], num_heads=4, use_spatial_transformer=True, transformer_depth=1, context_dim=16,
               use_checkpoint=False, legacy=False)
    m = UNetModel(**cfg)
    m.load_state_dict(synthetic.fill_state_dict(m.state_dict(), seed=3))
    m.split = True
    wq = dict(n_bits=4, channel_wise=True, scale_method="max")
    aq = dict(n_bits=8, channel_wise=False, scale_method="max", leaf_param=True)
    qnn = qdiff.QuantModel(m.eval(), wq, aq, sm_abit=16).eval()
    qnn.set_quant_state(True, True)
    g = torch.Generator().manual_seed(0)
    x, t, c = torch.randn(2, 4, 16, 16, generator=g), torch.tensor([10, 500]), torch.randn(2, 5, 16, generator=g)
    with torch.no_grad():
        qnn(x, t, c)                      # data-dependent initialisation
        calls.clear()
        qnn(x, t, c)
    # decoder ResBlocks at 16x16 (256 rows per sample) see concatenated inputs of 64 / 96 channels
    cat_inputs = [(C, part) for C, part in calls if C in (64, 96)]
    assert cat_inputs and any(part for _, part in cat_inputs), cat_inputs


@pytest.mark.parametrize("name", ["sd_tiny", "ldm_tiny", "cifar_tiny", "ldm_updown_tiny"])
def test_reference_unet_takes_the_planned_concatenation_walk(monkeypatch, name):
    """On the REFERENCE's UNetModel the integer state runs this repo's walk (QuantModel._adopt_reference_walk): from the
    second evaluation on every skip concatenation is a view (no `cat` copy of activations), and the output equals the one
    obtained through the reference's own forward + th.cat (QDIFF_REF_WALK=1) bit for bit."""
    from qdiff import quant_block as qb
    abi_emulator.install(monkeypatch)
    fx = load_fixture(f"model_{name}.pt")
    x, t, c = fixture_inputs(fx, "test")
    run = lambda q: (q(x, t, c) if c is not None else q(x, t))
    monkeypatch.setenv("QDIFF_REF_WALK", "1")
    q_ref = _wrap_and_resume(fx)
    with torch.no_grad():
        want = run(q_ref)
    monkeypatch.delenv("QDIFF_REF_WALK")
    qnn = _wrap_and_resume(fx)
    with torch.no_grad():
        y0 = run(qnn)                                   # records the channel plan
    views = {"n": 0, "cat": 0}
    real_adj, real_cat = qb._adjacent, torch.cat

    def adj(a, b, dim, unit):
        out = real_adj(a, b, dim, unit)
        if out is not None and dim == 1:
            views["n"] += 1
        return out

    def cat(ts, dim=0, **kw):
        if dim == 1 and len(ts) == 2 and ts[0].dim() == 4 and ts[0].is_floating_point():
            views["cat"] += 1
        return real_cat(ts, dim=dim, **kw)
    monkeypatch.setattr(qb, "_adjacent", adj)
    monkeypatch.setattr(torch, "cat", cat)
    with torch.no_grad():
        y1 = run(qnn)
    assert views["n"] == len(qnn.model.__dict__["_cat_plan"]) and views["cat"] == 0, views
    assert torch.equal(y0, want) and torch.equal(y1, want)
