#!/bin/bash
# Round 6, GPU call 3: attention tweaks (ring-turn unroll, +128 bias, early V^T reads) — parity, A/B vs round 5, whole SD step A/B,
# then the full default bench line (new by_class / box fields, eta = 1 LDM line, SD extra batch).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c3
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -m gpu -x -q -k "attention or sd_tiny or hooks_registered or repreparing or samplers_share" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
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
for ln in open("gpurun_out/r06_c3/sd_ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"),
              "by_class", {k: v.get("ms") for k, v in (r.get("by_class") or {}).items()}, "box", d.get("box"))
PY
cat $O/sd_ab_summary.txt
tail -5 $O/sd_ab.err
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_c3/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "whole", r["whole_step_frac"])
print("by_class", json.dumps(r["by_class"]))
print("box", d.get("box"), "frac_of_box", r.get("frac_of_box_ubench"), "eval_ms_eager", r.get("eval_ms_eager"), "launches", r.get("library_launches_per_eval"))
print("attention_calls", json.dumps(r.get("attention_calls")))
print("extra_batch", d["config"].get("extra_batch"))
oc=d.get("other_configs",{})
for k,v in oc.items(): print(k, json.dumps(v)[:700])
PY
