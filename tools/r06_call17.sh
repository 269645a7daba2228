#!/bin/bash
# Round 6, GPU call 17: P.V MFMAs deferred by one tile and spread over the next tile's probability chain (attn_pv_kernel, hi + lo):
# attention tests on the variant library, then A/B against the product library in the same call.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c17
mkdir -p $O
VAR=$PWD/q-diffusion_amd/lib/libqdiff_hip_pvdefer.so
QDIFF_HIP_LIB=$VAR timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $O/pytest_variant.log 2>&1; echo "pytest rc=$?" >> $O/pytest_variant.log
tail -8 $O/pytest_variant.log
for rep in 1 2 3; do
  echo "== product rep=$rep" >> $O/attn_ab.txt; timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
  echo "== deferred rep=$rep" >> $O/attn_ab.txt; QDIFF_HIP_LIB=$VAR timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
done
cat $O/attn_ab.txt
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  echo "== sd product rep=$rep" >> $O/ab.log; timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
  echo "== sd deferred rep=$rep" >> $O/ab.log; QDIFF_HIP_LIB=$VAR timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c17/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "attention", (r.get("by_class") or {}).get("attention",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
