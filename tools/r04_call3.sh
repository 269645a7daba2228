#!/bin/bash
# Round 4, GPU call 3: the prepared-context GPU test with its traceback; d = 80 on the lean kernel (parity + A/B).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c3; mkdir -p $out
timeout 600 python -m pytest tests/test_engine_models.py -m gpu -q -x -k "prepared_context" 2>&1 | grep -v "^Loading\|^Initializing" | tail -40
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention or attn" 2>&1 | tail -5
for lean in 2 1 2 1; do echo "== QD_ATTN_LEAN=$lean"; QD_ATTN_LEAN=$lean timeout 200 python tools/bench_attn.py 5 "sd " 2>&1 | tail -4; done | tee $out/bench_attn_d80.txt
for lean in 2 1 2 1; do
  echo "== SD bench QD_ATTN_LEAN=$lean"
  QD_ATTN_LEAN=$lean timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-denominators --no-extras 2> $out/bench_sd_lean$lean.err | tee $out/bench_sd_lean$lean.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['igemm_ms_per_eval'])"
done
