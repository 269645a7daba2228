"""debug probe (GPU): what between two graph captures of one model corrupts the first graph's replay"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
mode = sys.argv[1]
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
g = torch.Generator(device=cuda).manual_seed(11)
c2 = torch.randn(c.shape, device=cuda, generator=g)
with torch.no_grad():
    w_c = qnn(x, t, c).clone(); w_c2 = qnn(x, t, c2).clone()
    w1 = qnn(x[:1], t[:1], c[:1]).clone()
    if mode.startswith("pin"):
        assert qnn.prepare_context(c)
    qnn.enable_hip_graphs(True)
    a1 = qnn(x, t, c).clone(); print("A first replay ok", torch.equal(a1, w_c))
    ckv = qnn.__dict__["_ctx_kv"]
    if mode == "pin_eager":            # eager unpinned evaluation in between
        e = qnn.model(x, t, c2).clone(); print("eager c2 ok", torch.equal(e, w_c2))
    if mode == "pin_chain":
        ckv._work(c2)
    if mode == "pin_chain_c":
        ckv._work(c)
    if mode == "pin_eager_c":          # eager unpinned evaluation of the SAME context values (a copy: not the pinned object)
        e = qnn.model(x, t, c.clone()).clone(); print("eager c-copy ok", torch.equal(e, w_c))
    if mode == "pin_empty":
        torch.cuda.synchronize(); torch.cuda.empty_cache()
    if mode in ("pin_B", "pin_Bnorep"):
        from qdiff.graph import GraphedUNet
        gb = GraphedUNet(qnn, x, t, c2)
        if mode == "pin_B":
            print("B replay ok", torch.equal(gb(x, t, c2), w_c2))
    if mode == "unpinned_two":         # two ordinary graphs (different batch) of one model
        b1 = qnn(x[:1], t[:1], c[:1]).clone(); print("B(batch 1) ok", torch.equal(b1, w1))
    if True:
        a2 = qnn(x, t, c).clone(); torch.cuda.synchronize(); print("A replay after ok", torch.equal(a2, w_c), float((a2 - w_c).abs().max()))
