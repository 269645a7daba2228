#!/bin/bash
# Round 5, GPU call 5: head-layout epilogue with 8 codes per lane — kernel tests + whole-UNet parity, A/B by knob (QD_HEADS8).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c5
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -q -m gpu -x \
  -k "heads or residual or attention_fused or attention_quantised or quantised_unet_matches_reference or prepared_context or default_graph" > $O/pytest_subset.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_subset.log; tail -4 $O/pytest_subset.log
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for rep in 1 2; do
  one "fp32 heads8 rep$rep"   env $B
  one "fp32 heads4 rep$rep"   env QD_HEADS8=0 $B
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c5/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl)
PY
cat $O/ab_summary.txt
