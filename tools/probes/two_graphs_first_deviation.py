"""debug probe (GPU): two prepared-context graphs replayed alternately — the first module whose output changes"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
from qdiff.graph import GraphedUNet
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
store, order, which = [{}, {}], [], [0]
def hook(name):
    def f(_m, _a, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point():
                key = f"{name}#{i}"
                st = store[which[0]]
                if key not in st:
                    st[key] = torch.empty(o.shape, dtype=o.dtype, device=o.device)
                    if key not in order:
                        order.append(key)
                st[key].copy_(o)
    return f
with torch.no_grad():
    w_c = qnn(x, t, c).clone()
    assert qnn.prepare_context(c)
    for n, m in qnn.model.named_modules():
        if n:
            m.register_forward_hook(hook(n))
    which[0] = 0; qnn.model(x, t, c)          # allocate the stores outside any capture
    which[0] = 1; qnn.model(x, t, c)
    which[0] = 0; A = GraphedUNet(qnn, x, t, c, pinned=True)
    which[0] = 1; B = GraphedUNet(qnn, x, t, c, pinned=True)
    ra = A(x, t, c).clone(); torch.cuda.synchronize()
    snap = {k: v.clone() for k, v in store[0].items()}
    rb = B(x, t, c).clone(); torch.cuda.synchronize()
    ra2 = A(x, t, c).clone(); torch.cuda.synchronize()
    print("A first ok", torch.equal(ra, w_c), "B ok", torch.equal(rb, w_c), "A second ok", torch.equal(ra2, w_c))
    bad = [k for k in order if k in snap and not torch.equal(store[0][k], snap[k])]
    print(len(order), "outputs;", len(bad), "changed; first:", bad[:10])
    badb = [k for k in order if k in snap and k in store[1] and not torch.equal(store[1][k], snap[k])]
    print("B vs A-first:", len(badb), badb[:6])
