#!/bin/bash
# Round 6, GPU call 12: per-query code shift + signed hi bytes + per-tile hi skip in attn_pv_kernel (hi + lo): attention tests
# (bit-identity with the lean kernel, oracle), then A/B against the previous commit's attention, same call.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c12
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
OLD=$PWD/q-diffusion_amd/lib/libqdiff_hip_prev.so
for rep in 1 2; do
  for flat in 0 1; do
    echo "== previous flat=$flat rep=$rep" >> $O/attn_ab.txt; BENCH_ATTN_FLAT=$flat QDIFF_HIP_LIB=$OLD timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
    echo "== shifted flat=$flat rep=$rep" >> $O/attn_ab.txt; BENCH_ATTN_FLAT=$flat timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
  done
done
cat $O/attn_ab.txt
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  echo "== sd previous rep=$rep" >> $O/ab.log; QDIFF_HIP_LIB=$OLD timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
  echo "== sd shifted rep=$rep" >> $O/ab.log; timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c12/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "attention", (r.get("by_class") or {}).get("attention",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
