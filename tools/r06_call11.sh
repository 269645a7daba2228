#!/bin/bash
# Round 6, GPU call 11: what would a per-tile skip of the hi-byte P.V MFMAs be worth?  Ablation library (hi MFMAs never issued,
# the test on the packed hi bytes kept: WRONG results, timing only) against the product library, same call.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c11
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
ABL=$PWD/q-diffusion_amd/lib/libqdiff_hip_hiskip.so
for rep in 1 2; do
  echo "== product rep=$rep" >> $O/attn_ab.txt; timeout 300 python tools/bench_attn.py 10 "self 64x64" >> $O/attn_ab.txt 2>&1
  echo "== hi MFMAs skipped rep=$rep" >> $O/attn_ab.txt; QDIFF_HIP_LIB=$ABL timeout 300 python tools/bench_attn.py 10 "self 64x64" >> $O/attn_ab.txt 2>&1
done
cat $O/attn_ab.txt
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  echo "== sd product rep=$rep" >> $O/ab.log; timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
  echo "== sd hi skipped rep=$rep" >> $O/ab.log; QDIFF_HIP_LIB=$ABL timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c11/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "attention", (r.get("by_class") or {}).get("attention",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
