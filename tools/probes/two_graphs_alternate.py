"""debug probe (GPU): two live graphs of one model, replayed alternately"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
from qdiff.graph import GraphedUNet
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
mode = sys.argv[1]
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
g = torch.Generator(device=cuda).manual_seed(11)
c2 = torch.randn(c.shape, device=cuda, generator=g)
with torch.no_grad():
    w_c = qnn(x, t, c).clone(); w_c2 = qnn(x, t, c2).clone()
    pa, pb = mode[0] == "p", mode[1] == "p"
    if pa or pb:
        assert qnn.prepare_context(c)
    A = GraphedUNet(qnn, x, t, c, pinned=pa)
    B = GraphedUNet(qnn, x, t, c if pb else c2, pinned=pb)
    wb = w_c if pb else w_c2
    for i in range(3):
        ra = A(x, t, c).clone(); rb = B(x, t, c if pb else c2).clone()
        torch.cuda.synchronize()
        print(mode, "round", i, "A ok", torch.equal(ra, w_c), "B ok", torch.equal(rb, wb))
