#!/bin/bash
# Round 6, GPU call 16: (1) GPU tests of the modules the knob pruning touched; (2) how much of the step is the erf of the GEGLU
# epilogue?  Ablation library (erf replaced by its argument: WRONG results, timing only) against the product library.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c16
mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_models.py tests/test_block_parity.py tests/test_first_stage_hip.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
ABL=$PWD/q-diffusion_amd/lib/libqdiff_hip_noerf.so
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  echo "== sd product rep=$rep" >> $O/ab.log; timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
  echo "== sd no erf rep=$rep" >> $O/ab.log; QDIFF_HIP_LIB=$ABL timeout 600 python bench.py $X >> $O/ab.log 2>> $O/ab.err
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c16/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm", (r.get("by_class") or {}).get("igemm",{}).get("ms"), "geglu", [c for c in (r.get("classes") or []) if "geglu" in str(c).lower()][:1], "box", (d.get("box") or {}).get("mfma_ubench_tops"))
PY
cat $O/ab_summary.txt
tail -3 $O/ab.err
