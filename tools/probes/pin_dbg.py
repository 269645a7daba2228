"""debug probe (GPU): prepared-context buffers after re-preparation vs the per-evaluation chain's"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
g = torch.Generator(device=cuda).manual_seed(11)
c2 = torch.randn(c.shape, device=cuda, generator=g)
ckv = qnn.__dict__["_ctx_kv"]
snap = lambda tag: {k: tuple(b.clone() for b in v) for k, v in ckv.__dict__.get("_bufs", {}).items() if k[0] == tag and isinstance(v, tuple)}
def same(a, b):
    ka = {k[1:]: v for k, v in a.items()}; kb = {k[1:]: v for k, v in b.items()}
    return {k: [bool(torch.equal(p, q)) for p, q in zip(ka[k], kb[k])] for k in ka}
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
with torch.no_grad():
    w_c = qnn(x, t, c).clone(); ev_c = snap("eval")
    w_c2 = qnn(x, t, c2).clone(); ev_c2 = snap("eval")
    assert qnn.prepare_context(c); torch.cuda.synchronize()
    print("pin(c) == eval(c):", set(map(tuple, same(snap("pin"), ev_c).values())))
    if mode != "nograph":
        qnn.enable_hip_graphs(True)
        g1 = qnn(x, t, c).clone(); print("g1 ok", torch.equal(g1, w_c))
        if mode == "full":
            u1 = qnn(x, t, c2).clone(); print("u1 ok", torch.equal(u1, w_c2))
    assert qnn.prepare_context(c2); torch.cuda.synchronize()
    print("pin(c2) == eval(c2):", same(snap("pin"), ev_c2))
    g3 = qnn(x, t, c2).clone(); torch.cuda.synchronize()
    print("g3 ok", torch.equal(g3, w_c2), "== w_c", torch.equal(g3, w_c), float((g3 - w_c2).abs().max()))
    print("pin after g3 == eval(c2):", set(map(tuple, same(snap("pin"), ev_c2).values())))
    qnn.enable_hip_graphs(False)
    e3 = qnn(x, t, c2).clone(); print("eager pinned c2 ok", torch.equal(e3, w_c2), "pinned?", ckv.pinned(c2))
