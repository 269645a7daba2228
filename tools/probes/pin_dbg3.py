"""debug probe (GPU): does a captured evaluation hold a dangling pointer into memory that was allocated outside the capture?
after the capture: empty the allocator cache and overwrite everything the allocator hands out with NaN, then replay."""
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
with torch.no_grad():
    w_c = qnn(x, t, c).clone()
    if mode.startswith("pin"):
        assert qnn.prepare_context(c)
    qnn.enable_hip_graphs(True)
    a1 = qnn(x, t, c).clone(); print("first replay ok", torch.equal(a1, w_c))
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.full((64 << 20,), float("nan"), device=cuda) for _ in range(8)]      # 2 GB of NaN over whatever was freed
    torch.cuda.synchronize()
    a2 = qnn(x, t, c).clone(); torch.cuda.synchronize()
    print("replay after junk ok", torch.equal(a2, w_c), "nan" if torch.isnan(a2).any() else float((a2 - w_c).abs().max()))
