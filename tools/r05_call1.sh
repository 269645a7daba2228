#!/bin/bash
# Round 5, GPU call 1: the new/changed GPU tests, then A/B of the fp32 / fp16 activation streams (new library, the round-4
# library built from 4447426, the 4-halves-per-lane fp16 form, the always-even pairing) and the as-script line.
# Everything lands in gpurun_out/r05_c1/.
set -u
O=gpurun_out/r05_c1
mkdir -p $O
export TMPDIR=/tmp
python - <<'PY' > $O/env.txt 2>&1
import torch, subprocess
print(torch.__version__, torch.cuda.get_device_name(0))
PY
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -q -m gpu > $O/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $O/pytest_subset.log
B="python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
run() { # name, env..., -- args
  name=$1; shift
  ( env "$@" > /dev/null 2>&1 ) # no-op
}
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
R04=$PWD/q-diffusion_amd/lib/libqdiff_hip_r04.so
NOODD=$PWD/q-diffusion_amd/lib/libqdiff_hip_noodd.so
for rep in 1 2; do
  one "fp32 new rep$rep"        env $B
  one "fp32 r04lib rep$rep"     env QDIFF_HIP_LIB=$R04 $B
  one "fp16 lines rep$rep"      env $B --stream fp16
  one "fp16 4-wide rep$rep"     env QD_F16_LINES=0 $B --stream fp16
  one "fp16 noodd rep$rep"      env QDIFF_HIP_LIB=$NOODD $B --stream fp16
done
one "as-script" env python bench.py --as-script
one "as-script no-auto (graphs off, per-evaluation chain)" env QDIFF_HIP_GRAPH=0 QDIFF_CTX_AUTO=0 python bench.py --as-script
grep -E '^==|"ms_per_step"' $O/ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"by_launch_class": (\{.*\})\}.*/ms_per_step \1 \2/' > $O/ab_summary.txt
python - <<'PY' >> $O/ab_summary.txt
import json,re
name=None
for ln in open("gpurun_out/r05_c1/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln)
        r=d.get("roofline",{})
        cl={k:v["ms"] for k,v in r.get("by_launch_class",{}).items()}
        print(name, "ms_per_step", d.get("ms_per_step"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), cl, {k:d[k] for k in ("context_chain_runs_in_run","contexts_recognised_by_value","graphs_captured") if k in d})
PY
