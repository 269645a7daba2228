"""N2 — calibration (block / layer reconstruction) against fixtures produced by the REAL reference's calibration code
(tools/make_golden_recon.py: the reference's block_recon.py / layer_recon.py / utils.py driven by the `recon_model` walk of
its scripts, same seeds, same synthetic weights, a few iterations per unit, CPU).  The simulation graph autograd
differentiates here is this repo's (`_forward_sim` compositions, QuantModule's fp32 path) — the same operations as the
reference's, so the trained parameters agree to fp32 rounding; Adam's normalised steps (|step| = lr whatever the
gradient's size) turn a sign flip of a ~0 gradient into a 2*lr difference, hence the small per-element allowance."""
import numpy as np
import pytest
import torch

from golden_util import build_engine_model, load_fixture, quant_params


def _inputs(spec, batch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch,) + tuple(spec["x"]), generator=g)
    t = torch.randint(0, 1000, (batch,), generator=g)
    if spec["family"] == "cifar":
        t = t.float()
    c = torch.randn((batch,) + tuple(spec["ctx"]), generator=g) if spec["ctx"] else None
    return x, t, c


def _summary(t):
    f = t.detach().flatten().double()
    step = max(1, f.numel() // 64)
    return dict(numel=f.numel(), n_up=int((f >= 0).sum()), sum=float(f.sum()), l2=float(f.norm()), sample=f[::step][:64].float())


@pytest.mark.parametrize("name", ["cifar_tiny", "sd_tiny", "ldm_tiny", "ldm_updown_tiny"])
def test_calibration_matches_the_reference(name):
    _calibration_vs_reference_fixture(name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cifar_tiny", "sd_tiny", "ldm_updown_tiny"])
def test_calibration_matches_the_reference_on_gpu(cuda, name):
    """The same fixtures — produced by the REFERENCE's own calibration code on the CPU — with the model, the cached unit
    inputs / outputs and the fused fake-quant kernels on the MI355X (mini-batch indices come from the host generator, so
    the batch sequence is the reference's): the same bounds as on the CPU."""
    _calibration_vs_reference_fixture(name, cuda, on_gpu=True)


def _calibration_vs_reference_fixture(name, dev, on_gpu=False):
    import qdiff
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.calibrate import recon_model
    from qdiff.quant_layer import UniformAffineQuantizer
    fx = load_fixture(f"recon_{name}.pt")
    spec = fx["spec"]
    cond = spec["ctx"] is not None
    wq, aq = quant_params(spec)
    xs, ts, cs = _inputs(spec, fx["n_cal"], fx["cal_seed"])
    cali = (xs, ts, cs) if cond else (xs, ts)                # calibration data stays on the host, as in the scripts
    test = tuple(a.to(dev) for a in _inputs(spec, 2, fx["test_seed"]) if a is not None)
    qnn = qdiff.QuantModel(build_engine_model(spec).to(dev), wq, aq, sm_abit=spec["sm_abit"]).to(dev).eval()
    on_dev = lambda batch: tuple(a.to(dev) for a in batch)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(*on_dev(cali))
    torch.manual_seed(fx["seed"])
    np.random.seed(fx["seed"])
    if fx["iters_w"]:                                     # (the LDM fixture covers the activation phase only: see the tool)
        recon_model(qnn, cali_data=cali, batch_size=fx["batch"], iters=fx["iters_w"], weight=0.01, asym=True, b_range=(20, 2),
                    warmup=0.2, act_quant=False, opt_mode='mse', cond=cond)
    qnn.set_quant_state(True, False)
    got = {k: _summary(m.alpha) for k, m in qnn.named_modules() if isinstance(m, AdaRoundQuantizer)}
    assert set(got) == set(fx["alphas"]), "the set of AdaRound quantisers differs from the reference's"
    lr = 1e-3                                             # Adam default: the size of one step
    flips = total = 0
    worst_frac = 0.0
    for k, want in fx["alphas"].items():
        g = got[k]
        assert g["numel"] == want["numel"], k
        d = (g["sample"].cpu() - want["sample"]).abs()
        assert d.max().item() <= 2 * lr * fx["iters_w"] + 1e-5, (k, d.max().item())     # never further than every step reversed
        worst_frac = max(worst_frac, (d > 1e-4).float().mean().item())
        if not on_gpu:
            # same library, same summation order as the run that produced the fixture: only Adam's sign noise on ~0 gradients
            assert (d > 1e-4).float().mean().item() <= 0.05, (k, (d > 1e-4).float().mean().item())
        flips += abs(g["n_up"] - want["n_up"])
        total += want["numel"]
        assert abs(g["l2"] - want["l2"]) <= (5e-3 if on_gpu else 1e-3) * want["l2"] + 1e-4, (k, g["l2"], want["l2"])
    # On the GPU the fp32 convolutions of the simulation (MIOpen) sum in another order than the CPU's: most AdaRound
    # gradients of these few-iteration fixtures are noise-level, Adam normalises them to +-lr steps, so the element-wise
    # agreement above is not available; what is: no element further than the reachable distance, the norms, the rounding
    # DECISIONS (sign of alpha) and the network output after the weight phase.
    print(f"\n[{name}{' gpu' if on_gpu else ''}] AdaRound: worst fraction of sampled elements beyond 1e-4 = {worst_frac:.3f}, "
          f"rounding decisions that differ = {flips} of {total} ({flips / max(total, 1):.2e})")
    # GPU, measured round 3: 1.5e-4 (cifar_tiny), 0 (sd_tiny)
    assert flips <= (5e-4 if on_gpu else 1e-4) * total, f"{flips} of {total} rounding decisions differ"
    qnn.eval()
    with torch.no_grad():
        y = qnn(*test)
    assert (y.cpu() - fx["out_w"]).abs().max().item() <= 2e-3 * fx["out_w"].abs().max().item()
    # activation phase
    from qdiff import engine
    qnn.set_quant_state(True, True)
    with torch.no_grad(), engine.simulation():
        inds = torch.as_tensor(np.random.choice(xs.shape[0], 4, replace=False))
        qnn(*(a[inds].to(dev) for a in cali))
    recon_model(qnn, cali_data=cali, batch_size=fx["batch"], iters=fx["iters_a"], act_quant=True, opt_mode='mse', lr=4e-4, p=2.4,
                cond=cond)
    qnn.set_quant_state(True, True)
    deltas = {k: m.delta.detach() for k, m in qnn.named_modules()
              if isinstance(m, UniformAffineQuantizer) and getattr(m, "leaf_param", False) and m.inited and torch.is_tensor(m.delta)}
    assert set(deltas) == set(fx["deltas"])
    moved = 0
    for k, want in fx["deltas"].items():
        g = deltas[k]
        assert torch.allclose(g.reshape(-1).cpu(), want.reshape(-1), rtol=5e-3, atol=2 * 4e-4 * fx["iters_a"]), (k, g, want)
        moved += 1
    assert moved > 20
    qnn.eval()
    with torch.no_grad(), engine.simulation():
        y = qnn(*test)
    # a fake-quantised network amplifies the 1e-3-level step-size differences above through round() ties
    y = y.cpu()
    cos = torch.nn.functional.cosine_similarity(y.flatten(), fx["out_wa"].flatten(), dim=0).item()
    assert (y - fx["out_wa"]).abs().max().item() <= 0.12 * fx["out_wa"].abs().max().item() and cos >= 0.99, cos


def test_temperature_schedule_and_loss_terms():
    """rounding_temperature / ReconstructionObjective against closed forms (reference block_recon.py:169-252)."""
    from qdiff.recon import ReconstructionObjective, rounding_temperature
    td = lambda t: rounding_temperature(t, 100, hold=0.2, b_first=20, b_last=2)
    assert td(0) == 20 and td(19) == 20 and td(100) == 2
    assert abs(td(60) - (2 + 18 * (1 - 40 / 80))) < 1e-12

    class Q(torch.nn.Module):
        def get_soft_targets(self):
            return torch.tensor([0.0, 0.25, 0.5, 1.0])
    import qdiff
    m = qdiff.QuantModule(torch.nn.Linear(4, 4), dict(n_bits=4, channel_wise=True, scale_method="max"),
                          dict(n_bits=8, channel_wise=False, scale_method="max"))
    m.weight_quantizer = Q()
    lf = ReconstructionObjective(m, 10, weight=0.5, kind='mse', b_range=(20, 2), warmup=0.2, p=2.0)
    pred, tgt = torch.ones(2, 3), torch.zeros(2, 3)
    assert float(lf(pred, tgt)) == 3.0                               # call 1 < warm-up: reconstruction term only
    lf.calls = 5
    b = rounding_temperature(6, 10, 0.2, 20, 2)
    want = 3.0 + 0.5 * float((1 - ((torch.tensor([0.0, 0.25, 0.5, 1.0]) - .5).abs() * 2).pow(b)).sum())
    assert abs(float(lf(pred, tgt)) - want) < 1e-6
    assert float(ReconstructionObjective(m, 10, regularise=False)(pred, tgt)) == 3.0     # activation phase: no regulariser


def test_capture_hooks_see_both_block_inputs():
    """save_inp_oup_data returns [xs, embs] for units called with two tensors and the full-precision output; with `asym`
    the inputs come from the quantised network (they differ from the fp inputs after the first quantised layer)."""
    import qdiff
    from qdiff.quant_block import QuantResnetBlock
    from qdiff.utils import save_inp_oup_data
    fx = load_fixture("recon_cifar_tiny.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    xs, ts, _ = _inputs(spec, 8, 300)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(xs, ts)
    blocks = [m for m in qnn.modules() if isinstance(m, QuantResnetBlock)]
    inp_fp, out_fp = save_inp_oup_data(qnn, blocks[2], (xs, ts), asym=False, act_quant=False, batch_size=4, keep_gpu=False)
    inp_q, out_q = save_inp_oup_data(qnn, blocks[2], (xs, ts), asym=True, act_quant=False, batch_size=4, keep_gpu=False)
    assert isinstance(inp_fp, list) and inp_fp[0].shape[0] == 8 and inp_fp[1].shape[0] == 8 and out_fp.shape[0] == 8
    assert torch.equal(out_fp, out_q)                       # outputs always come from the full-precision network
    assert not torch.equal(inp_fp[0], inp_q[0]) and not torch.equal(inp_fp[1], inp_q[1])   # (the time embedding is quantised too)
    assert qnn.training and blocks[2].use_weight_quant and not blocks[0].use_weight_quant


def test_output_gradient_tap_is_the_hand_written_kl_gradient():
    """save_grad_data (Fisher weights, reference utils.py:152-180, 271-310): the backward tap on a unit returns the gradient of
    KL(quantised up to the unit || full precision) w.r.t. the unit's output — compared with autograd.grad on a retained
    output of the same two passes; the hook is gone and the hand-back state is the reference's afterwards."""
    import qdiff
    from qdiff import engine
    from qdiff.quant_block import QuantResnetBlock
    from qdiff.utils import quantize_up_to, save_grad_data
    fx = load_fixture("recon_cifar_tiny.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    xs, ts, _ = _inputs(spec, 4, 300)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(xs, ts)
    unit = [m for m in qnn.modules() if isinstance(m, QuantResnetBlock)][1]
    with engine.simulation():
        got = save_grad_data(qnn, unit, (xs, ts), act_quant=False, batch_size=4)
        assert got.shape[0] == 4 and float(got.min()) >= 1.0 and not unit._backward_hooks
        assert qnn.training and unit.use_weight_quant
        # the same quantity by hand: the LAST backward through the unit is the full-precision pass (autograd runs the later
        # pass first), whose output gradient is what the reference's hook keeps
        kept = []
        h = unit.register_forward_hook(lambda _m, _a, out: kept.append(out))
        qnn.eval()
        with torch.enable_grad():
            qnn.set_quant_state(False, False)
            tgt = torch.nn.functional.softmax(qnn(xs, ts), dim=1)
            quantize_up_to(qnn, unit, False)
            loss = torch.nn.functional.kl_div(torch.nn.functional.log_softmax(qnn(xs, ts), dim=1), tgt, reduction='batchmean')
            want = torch.autograd.grad(loss, kept[0])[0]
        h.remove()
    assert torch.allclose(got, want.abs() + 1.0, rtol=1e-5, atol=1e-7)


def test_calibrate_then_resume_round_trip(tmp_path):
    """calibrate_model -> reference-format checkpoint -> resume_cali_model on a fresh model reproduces the calibrated
    model's output (the producer and the consumer of the checkpoint agree on every key)."""
    import qdiff
    from qdiff import engine
    from qdiff.calibrate import calibrate_model
    from qdiff.utils import resume_cali_model
    fx = load_fixture("recon_cifar_tiny.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    xs, ts, _ = _inputs(spec, 8, 300)
    test = tuple(a for a in _inputs(spec, 2, 200) if a is not None)
    qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    torch.manual_seed(7)
    np.random.seed(7)
    seen = []
    sd = calibrate_model(qnn, (xs, ts), cond=False, quant_act=True, cali_batch_size=4, cali_iters=2, cali_iters_a=2,
                         act_init_batch=4, on_unit=lambda n, u: seen.append(n))
    assert len(seen) > 10 and any(k.endswith("weight_quantizer.alpha") for k in sd) and any(k.endswith("act_quantizer.delta") for k in sd)
    qnn.eval()
    with torch.no_grad(), engine.simulation():
        want = qnn(*test)
    path = tmp_path / "ckpt.pth"
    torch.save(sd, path)
    q2 = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    with engine.simulation():
        resume_cali_model(q2, str(path), (xs[:1], ts[:1]), quant_act=True, cond=False)
        with torch.no_grad():
            got = q2(*test)
    assert torch.equal(got, want)


def _dp_worker(rank, world, port, out_dir):
    """Data-parallel calibration: each rank reconstructs the same unit on ITS half of the calibration samples; gradients
    are averaged with an all-reduce every iteration (recon.reconstruct(multi_gpu=True)), so the trained parameters stay
    identical on all ranks although the data differ."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qdiff
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_block import QuantResnetBlock
    fx = load_fixture("recon_cifar_tiny.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    xs, ts, _ = _inputs(spec, 16, 300)
    qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        qnn(xs[:8], ts[:8])                               # same initialisation batch on every rank
    sel = slice(rank * 8, rank * 8 + 8)                   # ... but each rank calibrates on its own shard
    blk = [m for m in qnn.modules() if isinstance(m, QuantResnetBlock)][1]
    torch.manual_seed(11)                                 # same mini-batch index sequence (indices into different shards)
    qdiff.block_reconstruction(qnn, blk, (xs[sel], ts[sel]), batch_size=4, iters=4, weight=0.01, asym=True, warmup=0.2,
                               act_quant=False, opt_mode='mse', multi_gpu=True)
    alphas = {k: m.alpha.detach().clone() for k, m in blk.named_modules() if isinstance(m, AdaRoundQuantizer)}
    torch.save(alphas, os.path.join(out_dir, f"alphas_{rank}.pt"))
    # the whole sequence, every rank on ITS shard from the very first (initialisation) batch on: the quantiser ranges of rank 0
    # are adopted everywhere (calibrate.sync_quantisers), the gradients averaged -> one checkpoint on all ranks
    from qdiff.calibrate import calibrate_model
    q2 = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
    torch.manual_seed(5)
    np.random.seed(5)
    sel4 = slice(rank * 8, rank * 8 + 8)
    torch.set_num_threads(2)
    sd = calibrate_model(q2, (xs[sel4], ts[sel4]), cond=False, quant_act=True, cali_batch_size=4, cali_iters=1, cali_iters_a=1,
                         init_batch=4, act_init_batch=4, multi_gpu=True)
    torch.save(sd, os.path.join(out_dir, f"sd_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_calibration_two_ranks_gloo(tmp_path):
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0 = torch.load(os.path.join(tmp_path, "alphas_0.pt"))
    a1 = torch.load(os.path.join(tmp_path, "alphas_1.pt"))
    assert len(a0) >= 2 and set(a0) == set(a1)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k               # averaged gradients -> identical Adam trajectories
    s0, s1 = torch.load(os.path.join(tmp_path, "sd_0.pt")), torch.load(os.path.join(tmp_path, "sd_1.pt"))
    assert set(s0) == set(s1) and len(s0) > 100
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.gpu
def test_calibration_runs_on_the_gpu(cuda, tmp_path):
    """The whole calibration sequence on the MI355X (HBM-resident unit caches, fused fake-quant kernels in the step-size
    phase), then the checkpoint it wrote drives the INTEGER engine: keys as produced on the CPU, finite values, and the
    integer evaluation of the calibrated model close to its fp32 simulation (bound: the envelope of DESIGN.md §6)."""
    import qdiff
    from qdiff import engine
    from qdiff.calibrate import calibrate_model
    from qdiff.utils import resume_cali_model
    fx = load_fixture("recon_cifar_tiny.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    xs, ts, _ = _inputs(spec, 8, 300)
    test = tuple(a.to(cuda) for a in _inputs(spec, 2, 200) if a is not None)
    qnn = qdiff.QuantModel(build_engine_model(spec).to(cuda), wq, aq, sm_abit=spec["sm_abit"]).to(cuda).eval()
    torch.manual_seed(7)
    np.random.seed(7)
    sd = calibrate_model(qnn, (xs, ts), cond=False, quant_act=True, cali_batch_size=4, cali_iters=2, cali_iters_a=2, act_init_batch=4)
    assert any(k.endswith("weight_quantizer.alpha") for k in sd) and any(k.endswith("act_quantizer.delta") for k in sd)
    assert all(torch.isfinite(v.float()).all() for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())
    qnn.eval()
    with torch.no_grad(), engine.simulation():
        sim = qnn(*test).float().cpu()
    path = tmp_path / "ckpt.pth"
    torch.save({k: v.cpu() for k, v in sd.items()}, path)
    q2 = qdiff.QuantModel(build_engine_model(spec).to(cuda), wq, aq, sm_abit=spec["sm_abit"]).to(cuda).eval()
    resume_cali_model(q2, str(path), (xs[:1], ts[:1]), quant_act=True, cond=False)
    with torch.no_grad():
        got = q2(*test).float().cpu()                       # integer engine
    assert torch.isfinite(got).all()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), sim.flatten(), dim=0).item()
    assert (got - sim).abs().max().item() <= 0.15 * sim.abs().max().item() and cos >= 0.99, cos


@pytest.mark.parametrize("n_bits,shape", [(4, (48, 32, 3, 3)), (8, (40, 96)), (4, (16, 8, 1, 1))])
def test_vectorised_mse_range_search_matches_the_per_channel_loop(n_bits, shape):
    """UniformAffineQuantizer._init_mse_channelwise (all channels at once) vs the reference's per-channel loop
    (quant_layer.py:138-140,162-177, reproduced by init_quantization_scale on the host): the same (delta, zero_point) for
    every channel — except that a channel whose two best shrink factors score within fp32 summation noise may take the
    neighbouring factor (allowed on <= 2 % of the channels, and then by exactly one 1 % step)."""
    from qdiff import quant_layer as ql
    g = torch.Generator().manual_seed(17)
    w = torch.randn(shape, generator=g) * torch.rand(shape[0], *([1] * (len(shape) - 1)), generator=g)
    q = ql.UniformAffineQuantizer(n_bits=n_bits, symmetric=False, channel_wise=True, scale_method="mse")
    d_loop, z_loop = q.init_quantization_scale(w, channel_wise=True)
    d_vec, z_vec = q._init_mse_channelwise(w)
    assert d_loop.shape == d_vec.shape and z_loop.shape == z_vec.shape
    same = (d_loop == d_vec).flatten() & (z_loop == z_vec).flatten()
    assert same.float().mean().item() >= 0.98, same.float().mean().item()
    off = ~same
    if off.any():
        ratio = (d_vec.flatten()[off] / d_loop.flatten()[off])
        assert ((ratio - 1).abs() <= 0.015).all(), ratio


def test_split_layer_reconstruction_default_is_the_references(monkeypatch):
    """A SPLIT QuantModule reconstructed as a single layer (reference layer_recon.py:50-58: soft targets for
    `weight_quantizer` only, so `weight_quantizer_0.alpha` never receives a gradient) against a fixture produced by the
    reference's own layer_reconstruction (tools/make_golden_recon.py split_layer): by DEFAULT this package does the same —
    the first half within Adam's sign noise, the second half bit-identical (it is the untouched initialisation);
    QDIFF_TRAIN_BOTH_SPLIT_HALVES=1 is the opt-in that trains both."""
    import qdiff
    from qdiff import recon
    from qdiff.layer_recon import layer_reconstruction
    fx = load_fixture("recon_split_layer.pt")
    spec = fx["spec"]
    wq, aq = quant_params(spec)
    xs, ts, _ = _inputs(spec, fx["n_cal"], fx["cal_seed"])
    cali = (xs, ts)

    def run(both):
        monkeypatch.setattr(recon, "TRAIN_BOTH_SPLIT_HALVES", both)
        qnn = qdiff.QuantModel(build_engine_model(spec), wq, aq, sm_abit=spec["sm_abit"]).eval()
        qnn.set_quant_state(True, False)
        with torch.no_grad():
            qnn(*cali)
        layer = dict(qnn.named_modules())[fx["layer"]]
        assert layer.split == fx["split"]
        torch.manual_seed(fx["seed"])
        np.random.seed(fx["seed"])
        layer_reconstruction(qnn, layer, cali_data=cali, batch_size=fx["batch"], iters=fx["iters_w"], weight=0.01, asym=True,
                             b_range=(20, 2), warmup=0.2, act_quant=False, opt_mode='mse', cond=False)
        return layer.weight_quantizer.alpha.detach().clone(), layer.weight_quantizer_0.alpha.detach().clone()

    a, a0 = run(False)
    lr = 1e-3
    d = (a - fx["alpha"]).abs()
    assert d.max().item() <= 2 * lr * fx["iters_w"] + 1e-5 and (d > 1e-4).float().mean().item() <= 0.05
    assert torch.equal(a0, fx["alpha_0"]), "the second half of a split layer must stay untrained, as in the reference"
    b, b0 = run(True)
    assert (b0 - fx["alpha_0"]).abs().max().item() > 1e-4, "QDIFF_TRAIN_BOTH_SPLIT_HALVES=1 did not train weight_quantizer_0"
    assert (b0 - fx["alpha_0"]).abs().max().item() <= 2 * lr * fx["iters_w"] + 1e-5
