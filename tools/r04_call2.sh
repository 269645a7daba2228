#!/bin/bash
# Round 4, GPU call 2: prepared context (K / V^T of the run's conditioning once per sampling run): parity + A/B; the 32-bit
# attention epilogue; steady-state breakdown of the graph-replayed evaluation with the context prepared.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c2; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py -m gpu -q -x -k "attention or attn or prepared_context or graph or plms" 2>&1 | tail -5
BENCH_ATTN_FLAT=0 timeout 200 python tools/bench_attn.py 5 2>&1 | tail -5 | tee $out/bench_attn.txt
for pin in 0 1 0 1; do
  echo "== SD bench QDIFF_CTX_PIN=$pin"
  QDIFF_CTX_PIN=$pin timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-denominators --no-extras 2> $out/bench_sd_pin$pin.err | tee $out/bench_sd_pin$pin.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['igemm_ms_per_eval'], d['config']['context_prepare_ms'], d['roofline']['launches_per_eval'])"
done
timeout 600 rocprofv3 --kernel-trace -d $out -o evb -- python tools/eval_breakdown.py run sd 8 3 graph pin > $out/evb.log 2>&1
db=$(find $out -name 'evb_results.db' | head -1)
python tools/eval_breakdown.py join $db 3 > $out/sd_eval_breakdown_graph.txt; head -45 $out/sd_eval_breakdown_graph.txt | cut -c1-170
python tools/eval_breakdown.py timeline $db 3 $out/sd_eval_timeline.tsv
find $out -name '*.db' -delete
