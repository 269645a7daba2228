#!/bin/bash
# GPU box: the final HEAD of round 6 once more (after the deferred P.V MFMAs, the host-side knob pruning and the new attention test cases): the whole GPU
# suite, smoke, the driver's bench command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${QD_OUT:-r06f4}; mkdir -p $out
timeout 1900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_sd.json 2> $out/bench_sd.err; echo "bench rc=$?"; tail -4 $out/bench_sd.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06f4/bench_sd.json") if l.startswith("{")][-1])
r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "whole", r["whole_step_frac"], "box", d["box"].get("mfma_ubench_tops"), d["box"].get("exp_ginst_s"), "frac_of_box", r.get("frac_of_box_ubench"))
print({k: v.get("ms") for k, v in r["by_class"].items()}, "replay", r.get("graph_replay_eval_ms"))
oc=d["other_configs"]
print("fp16", oc["sd_fp16_stream"].get("ms_per_step"), "script", oc["sd_as_script"].get("ms_per_step"), "cifar", oc["cifar"].get("value"), "ldm", oc["ldm"].get("value"), "extra", d["config"].get("extra_batch"))
PY
