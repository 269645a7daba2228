"""debug probe (GPU): what is (re)built while a prepared-context evaluation is being captured?"""
import os, sys, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
from qdiff import engine, hip
from qdiff.graph import GraphedUNet
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
calls = []
def wrap(mod, name):
    real = getattr(mod, name)
    def f(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            calls.append((name, "".join(traceback.format_stack(limit=6)[:-1])))
        return real(*a, **k)
    setattr(mod, name, f)
for mod, names in ((engine, ["build_conv_plan", "pack_module_weights", "build_attn_plan", "qparams_of", "head_buffers"]),
                   (hip, ["make_qparams", "pack_weights_t4", "pack_weights_t8", "pack_weights", "quantize_heads", "attn_keyterm"])):
    for n in names:
        if hasattr(mod, n):
            wrap(mod, n)
real_zeros, real_empty = torch.zeros, torch.empty
with torch.no_grad():
    w_c = qnn(x, t, c).clone()
    assert qnn.prepare_context(c)
    for pinned in (True, False):
        calls.clear()
        G = GraphedUNet(qnn, x, t, c if pinned else c.clone(), pinned=pinned)
        seen = {}
        for n, st in calls:
            seen.setdefault(n, st)
        print("pinned" if pinned else "unpinned", "capture:", {n: sum(1 for m, _ in calls if m == n) for n in seen})
        for n, st in seen.items():
            if n not in ("quantize_heads",):
                print("  first", n, "from\n", st)
