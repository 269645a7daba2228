#!/bin/bash
# Round 4, GPU call 6: decoder graph replay (parity + timing), engine-model tests after the host-side changes, and the
# DEFAULT bench command exactly as the driver runs it (headline + cifar / ldm / decode child runs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c6; mkdir -p $out
timeout 600 python -m pytest tests/test_first_stage_hip.py -m gpu -q -k "graph or golden or sd_shape" 2>&1 | tail -3
for e in "QDIFF_DECODER_GRAPH=0 hip" "QDIFF_DECODER_GRAPH=1 hip" "QDIFF_DECODER_GRAPH=1 hip_bf16"; do set -- $e
  env $1 timeout 300 python bench.py --decode-leg $2 --images-per-gpu 8 2> $out/decode_$2_$1.err | sed "s/^/$1 /" | tee -a $out/decode_legs.jsonl
done
timeout 1200 python -m pytest tests/test_engine_models.py -m gpu -q -x 2>&1 | tail -3
( time timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real
tail -c 1500 $out/bench_default.err
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["whole_step_frac"], d["config"]["workload"])
print(json.dumps(d.get("other_configs"), indent=1)[:1800])
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
