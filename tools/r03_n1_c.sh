#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03n1c; mkdir -p $out
timeout 400 python -m pytest tests/test_first_stage_hip.py -m gpu -q -s 2>&1 | tail -8
timeout 400 python bench.py --steps 2 --warmup 1 --decode --no-cpu-baseline --no-denominators > $out/bench_decode.json 2> $out/bench_decode.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03n1c/bench_decode.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("first_stage_decode")))
PY
