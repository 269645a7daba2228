#!/bin/bash
# First GPU call of the next round: what round 3 could not re-run after its GPU budget ended.
#  1. bench.py --decode twice (the library decode faulted once at exactly 2^31 bytes per activation; decode_first_stage now
#     chunks by the true largest activation: expect chunks of 4 and no fault) + decode timing with the 256-thread
#     GroupNorm finalise (round 3 measured 3.77 ms per image before it, chunks of 8)
#  2. the first-stage tests, then the whole GPU suite and smoke
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04a; mkdir -p $out
for i in 1 2; do
  timeout 500 python bench.py --steps 2 --warmup 1 --decode --no-cpu-baseline --no-denominators > $out/bench_decode$i.json 2> $out/bench_decode$i.err
  echo "bench --decode run $i rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$out/bench_decode$i.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("first_stage_decode")))
except Exception as ex:
    print("no JSON line:", ex)
PY
done
timeout 400 python -m pytest tests/test_first_stage.py tests/test_first_stage_hip.py -m gpu -q -s 2>&1 | tail -8
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
