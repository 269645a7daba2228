#!/bin/bash
# Round 6, GPU call 1: the attention path as three launches (statistics / lo-only P.V / hi+lo P.V) — parity of the attention tests,
# then A/B/A/B against the round-5 library (_ab/r05: the tree at 88ad505 with its own build) on ONE box: the 4096-token call
# alone (flat and peaked rows) and the whole SD step.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c1
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
tail -3 $O/pytest_attn.log
for rep in 1 2; do
  for flat in 1 0; do
    echo "== old flat=$flat rep=$rep" >> $O/attn_ab.txt; (cd _ab/r05 && BENCH_ATTN_FLAT=$flat timeout 300 python tools/bench_attn.py 10 "sd self 64x64") >> $O/attn_ab.txt 2>> $O/attn_ab.err
    echo "== new flat=$flat rep=$rep" >> $O/attn_ab.txt; BENCH_ATTN_FLAT=$flat timeout 300 python tools/bench_attn.py 10 "sd self 64x64" >> $O/attn_ab.txt 2>> $O/attn_ab.err
  done
done
cat $O/attn_ab.txt
X="--no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  echo "== old rep=$rep" >> $O/sd_ab.log; (cd _ab/r05 && timeout 600 python bench.py $X) >> $O/sd_ab.log 2>> $O/sd_ab.err
  echo "== new rep=$rep" >> $O/sd_ab.log; timeout 600 python bench.py $X >> $O/sd_ab.log 2>> $O/sd_ab.err
done
python - <<'PY' > $O/sd_ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r06_c1/sd_ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"))
PY
cat $O/sd_ab_summary.txt
tail -5 $O/sd_ab.err
