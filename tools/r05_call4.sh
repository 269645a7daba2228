#!/bin/bash
# Round 5, GPU call 4: packed byte quantisers in the GroupNorm / LayerNorm producers + explicit fma in the epilogues — kernel and
# model tests, A/B against the library built from the previous commit (QDIFF_HIP_LIB), both streams.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c4
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -q -m gpu > $O/pytest_subset.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_subset.log; tail -4 $O/pytest_subset.log
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
PRE=$PWD/q-diffusion_amd/lib/libqdiff_hip_pre.so
for rep in 1 2; do
  one "fp32 new rep$rep"   env $B
  one "fp32 pre rep$rep"   env QDIFF_HIP_LIB=$PRE $B
done
one "fp16 new" env $B --stream fp16
one "fp16 pre" env QDIFF_HIP_LIB=$PRE $B --stream fp16
one "fp32 new geglu-mt2" env QD_GEGLU_MT=2 $B
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c4/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl)
PY
cat $O/ab_summary.txt
