#!/bin/bash
# Round 6, GPU call 19: deferred P.V MFMAs as the product path: attention tests, whole-model tests, SD A/B/A/B against the previous
# commit's attention.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c19
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -m gpu -q -k "attention or golden or sd or tiny" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
OLD=$PWD/q-diffusion_amd/lib/libqdiff_hip_prev.so
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2 3; do
  echo "== sd previous rep=$rep" >> $O/ab.log; QDIFF_HIP_LIB=$OLD timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
  echo "== sd deferred rep=$rep" >> $O/ab.log; timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c19/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "attention", (r.get("by_class") or {}).get("attention",{}).get("ms"), "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
