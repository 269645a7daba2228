"""debug probe (GPU): first module whose output changes when freed memory is overwritten between two replays of the
prepared-context graph.  Forward hooks copy every module output into persistent buffers (the copies are captured)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
store, order = {}, []
def hook(name):
    def f(_m, _a, out):
        if torch.is_tensor(out) and out.is_floating_point():
            if name not in store:
                store[name] = torch.empty_like(out, memory_format=torch.contiguous_format)
            if out.shape == store[name].shape:
                store[name].copy_(out)
                if name not in order:
                    order.append(name)
    return f
with torch.no_grad():
    w_c = qnn(x, t, c).clone()
    assert qnn.prepare_context(c)
    for n, m in qnn.model.named_modules():
        if n:
            m.register_forward_hook(hook(n))
    qnn(x, t, c)                    # eager: allocates the stores
    qnn.enable_hip_graphs(True)
    a1 = qnn(x, t, c).clone(); print("first replay ok", torch.equal(a1, w_c))
    snap = {k: v.clone() for k, v in store.items()}
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.full((64 << 20,), float("nan"), device=cuda) for _ in range(8)]
    torch.cuda.synchronize()
    a2 = qnn(x, t, c).clone(); torch.cuda.synchronize()
    print("replay after junk ok", torch.equal(a2, w_c))
    bad = [n for n in order if not torch.equal(store[n], snap[n])]
    print(len(order), "module outputs,", len(bad), "changed; first:", bad[:12])
