"""debug probe (GPU): which kernel operand of the captured (prepared-context) evaluation is dead after the capture?
Every tensor whose pointer goes to the library during the capture is recorded with a weak reference to its storage."""
import os, sys, weakref, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "q-diffusion_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from golden_util import fixture_inputs, load_fixture
import test_engine_models as T
from qdiff import hip
cuda = torch.device("cuda:0")
fx = load_fixture("model_sd_tiny.pt")
qnn = T._resume(fx, cuda)
x, t, c = (a.to(cuda) for a in fixture_inputs(fx, "test"))
rec, on = [], [False]
real = hip._ptr
def spy(tn, name="tensor"):
    if on[0] and tn is not None and tn.is_cuda and torch.cuda.is_current_stream_capturing():
        st = tn.untyped_storage()
        rec.append((weakref.ref(st), st.data_ptr(), st.nbytes(), tuple(tn.shape), str(tn.dtype), "".join(traceback.format_stack(limit=7)[:-1])))
    return real(tn, name)
hip._ptr = spy
with torch.no_grad():
    w_c = qnn(x, t, c).clone()
    assert qnn.prepare_context(c)
    live_before = None
    qnn.enable_hip_graphs(True)
    on[0] = True
    a1 = qnn(x, t, c).clone()
    on[0] = False
    import gc; gc.collect()
    dead = [r for r in rec if r[0]() is None]
    print(len(rec), "operands recorded during capture,", len(dead), "dead now")
    # dead ones allocated INSIDE the capture belong to the graph's private pool (fine); the suspicious ones are outside it
    g = list(qnn._graphs.values())[0]
    seen = set()
    for r in dead:
        key = (r[3], r[4], r[5])
        if key in seen:
            continue
        seen.add(key)
    print(len(seen), "distinct dead operand sites")
    # which of them lie outside the graph's private pool (allocated before the capture, freed since)?
    segs = torch.cuda.memory_snapshot()
    priv = [(sg["address"], sg["address"] + sg["total_size"]) for sg in segs if sg.get("segment_pool_id", (0, 0)) != (0, 0)]
    print(len(segs), "segments,", len(priv), "in private pools")
    hit = set()
    for r in rec:
        inside = any(a <= r[1] < b for a, b in priv)
        alive = r[0]() is not None
        if not inside and not alive:
            k = (r[3], r[4])
            if k not in hit:
                hit.add(k)
                print("DANGLING operand (outside the graph pool, storage freed)", r[3], r[4], "bytes", r[2], "\n", r[5])
    print(len(hit), "dangling operand shapes")
