"""Samplers vs trajectories of the reference's OWN samplers (tests/golden/samplers.pt, produced by
ldm/models/diffusion/{plms,ddim}.py and ddim/functions/denoising.py driving a stub eps-model), and the
batch-sharded launcher on world_size-2 gloo (CPU): a sharded run reproduces the single-process samples
bit for bit and the quant-state broadcast delivers rank 0's tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from golden_util import load_fixture


def stub_eps(x, t, c=None):
    tt = t.float().view(-1, 1, 1, 1)
    out = torch.tanh(0.3 * x + 1e-3 * tt) + 0.1 * torch.roll(x, 1, dims=-1)
    if c is not None:
        out = out + 0.05 * c.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    return out


def test_plms_matches_reference_sampler():
    from qdiff import sampling
    fx = load_fixture("samplers.pt")["plms"]
    calls = []

    def unet(x, t, c=None):
        calls.append(x.shape[0])
        return stub_eps(x, t, c)
    table = sampling.StepTable(sampling.ldm_betas(fx["ls"], fx["le"]), fx["steps"], eta=0.0)
    out = sampling.plms_sample(unet, fx["xT"], table, cond=fx["c"], uncond=fx["uc"], scale=fx["scale"])
    assert len(calls) == fx["calls"] == 51 and all(b == 6 for b in calls)      # CFG doubles the batch (App. E 11)
    assert torch.equal(out, fx["out"])


def test_ddim_matches_reference_sampler():
    from qdiff import sampling
    fx = load_fixture("samplers.pt")["ddim"]
    table = sampling.StepTable(sampling.ldm_betas(fx["ls"], fx["le"]), fx["steps"], eta=0.0)
    assert list(table.timesteps[:3]) == [1, 51, 101]
    out = sampling.ddim_sample(lambda x, t, c=None: stub_eps(x, t, c), fx["xT"], table)
    assert torch.equal(out, fx["out"])


def test_generalized_steps_matches_reference():
    from qdiff import sampling
    fx = load_fixture("samplers.pt")["generalized"]
    assert sampling.quad_sequence(1000, 20) == fx["seq"]
    betas = torch.from_numpy(sampling.ddpm_betas()).float()
    out = sampling.generalized_steps(lambda x, t: stub_eps(x, t), fx["x"], fx["seq"], betas, eta=0.0)
    assert torch.equal(out, fx["out"])


def test_device_resident_plms_matches_reference_sampler():
    """DevicePLMS (step counter, coefficients and multistep history on the device: one capturable step) == the reference's
    PLMSSampler trajectory, bit for bit."""
    from qdiff import sampling
    fx = load_fixture("samplers.pt")["plms"]
    calls = []

    def unet(x, t, c=None):
        calls.append(x.shape[0])
        return stub_eps(x, t, c)
    table = sampling.StepTable(sampling.ldm_betas(fx["ls"], fx["le"]), fx["steps"], eta=0.0)
    out = sampling.DevicePLMS(unet, table, fx["xT"], cond=fx["c"], uncond=fx["uc"], scale=fx["scale"]).run()
    assert len(calls) == 51 and torch.equal(out, fx["out"])


@pytest.mark.parametrize("key", ["dpm10", "dpm20"])
def test_dpm_solver_matches_reference_sampler(key):
    """DPM-Solver++(2M) as txt2img.py --dpm_solver runs it (DPMSolverSampler.sample): bit for bit, S model calls on the
    CFG-doubled batch with float timestep labels; 10 steps exercise lower_order_final."""
    from qdiff import sampling
    fx = load_fixture("samplers.pt")[key]
    calls = []

    def unet(x, t, c=None):
        assert t.dtype == torch.float32
        calls.append(x.shape[0])
        return stub_eps(x, t, c)
    acp = torch.tensor(np.cumprod(1.0 - sampling.ldm_betas(fx["ls"], fx["le"]), axis=0), dtype=torch.float32)
    out = sampling.dpm_solver_sample(unet, fx["xT"], acp, fx["steps"], cond=fx["c"], uncond=fx["uc"], scale=fx["scale"])
    assert len(calls) == fx["calls"] == fx["steps"] and all(b == 6 for b in calls)
    assert torch.equal(out, fx["out"])


def test_shard_bounds_cover_batch():
    from qdiff.sampling import shard_bounds
    for gb in (1, 7, 8, 64, 65):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(gb, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gb, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace as NS
    from qdiff import sampling
    dev = torch.device("cpu")
    # --- quant-state broadcast: rank 0 holds the "calibrated" tensors, others hold garbage ---------
    torch.manual_seed(100 + rank)
    plan = NS(pack=NS(wq=torch.randint(0, 255, (64,), dtype=torch.uint8),
                      segs=[dict(wsum=torch.randint(-9, 9, (4,), dtype=torch.int32), delta_w=torch.rand(4), zw=None, wzp=None)]),
              segs=[dict(scale=torch.rand(4), zc=torch.randint(-9, 9, (4,), dtype=torch.int32), zfill=None)],
              qparams=[torch.rand(2)], bias=torch.rand(4))
    from qdiff.quant_layer import QuantModule
    mod = QuantModule(torch.nn.Linear(4, 4))
    mod._plan = plan
    qnn = NS(model=torch.nn.Sequential(mod))
    before = [t.clone() for t in sampling.quant_state_tensors(qnn)]
    nbytes = sampling.broadcast_quant_state(qnn, src=0)
    after = sampling.quant_state_tensors(qnn)
    torch.save(dict(before=before, after=[t.clone() for t in after], nbytes=nbytes), os.path.join(out_dir, f"state_{rank}.pt"))
    # --- sharded sampling --------------------------------------------------------------------------
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.012), 10, eta=0.0)
    shape = (gb, 4, 8, 8)
    x = sampling.sharded_noise(shape, 0, world, rank, dev)
    c = sampling.sharded_noise((gb, 5, 6), 1, world, rank, dev)
    uc = sampling.sharded_noise((gb, 5, 6), 2, world, rank, dev)
    out = sampling.plms_sample(lambda xx, tt, cc=None: stub_eps(xx, tt, cc), x, table, cond=c, uncond=uc, scale=7.5)
    full = sampling.gather_samples(out, gb)
    if rank == 0:
        torch.save(full, os.path.join(out_dir, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gb", [6, 7])
def test_sharded_sampling_two_ranks_gloo(tmp_path, gb):
    from qdiff import sampling
    port = _free_port()
    mp.spawn(_worker, args=(2, port, gb, str(tmp_path)), nprocs=2, join=True)
    # single-process reference run
    table = sampling.StepTable(sampling.ldm_betas(0.00085, 0.012), 10, eta=0.0)
    dev = torch.device("cpu")
    x = sampling.sharded_noise((gb, 4, 8, 8), 0, 1, 0, dev)
    c = sampling.sharded_noise((gb, 5, 6), 1, 1, 0, dev)
    uc = sampling.sharded_noise((gb, 5, 6), 2, 1, 0, dev)
    want = sampling.plms_sample(lambda xx, tt, cc=None: stub_eps(xx, tt, cc), x, table, cond=c, uncond=uc, scale=7.5)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    assert got.shape == want.shape and torch.equal(got, want)
    s0 = torch.load(os.path.join(tmp_path, "state_0.pt"))
    s1 = torch.load(os.path.join(tmp_path, "state_1.pt"))
    assert s0["nbytes"] == s1["nbytes"] > 0
    for b0, a0, b1, a1 in zip(s0["before"], s0["after"], s1["before"], s1["after"]):
        assert torch.equal(a0, b0)                   # the source rank is unchanged
        assert torch.equal(a1, b0)                   # the other rank received rank 0's values
    assert any(not torch.equal(b0, b1) for b0, b1 in zip(s0["before"], s1["before"]))
