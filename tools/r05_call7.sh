#!/bin/bash
# Round 5, GPU call 7: the LDM AttentionBlock's qkv as three operand projections — parity tests, A/B of the LDM line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c7
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py tests/test_block_parity.py tests/test_first_stage_hip.py -m gpu -x -q \
  -k "projection_heads_epilogue or ldm_attention_qkv or (blocks_teacher_forced and (ldm or churches)) or (quantised_unet_matches_reference and (ldm or churches)) or hip_graph_replay_equals_eager or autocast or adopted" \
  > $O/pytest_targeted.log 2>&1; echo "pytest rc=$?" >> $O/pytest_targeted.log; tail -5 $O/pytest_targeted.log
B="python bench.py --model ldm --images-per-gpu 64 --extra-batch 10 --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
for rep in 1 2; do
  for v in 1 0; do
    echo "== ldm QDIFF_QKV_HEADS=$v rep$rep" >> $O/ab.log
    QDIFF_QKV_HEADS=$v $B >> $O/ab.log 2>> $O/ab.err
  done
done
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c7/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), d["config"].get("extra_batch"))
PY
cat $O/ab_summary.txt
timeout 300 rocprofv3 --kernel-trace -d $O -o evb_ldm -- python tools/eval_breakdown.py run ldm 64 3 graph > $O/evb_ldm.log 2>&1
db=$(find $O -name "evb_ldm_results.db" | head -1)
python tools/eval_breakdown.py join $db 3 > $O/ldm_eval_breakdown_graph.txt; head -24 $O/ldm_eval_breakdown_graph.txt | cut -c1-150
find $O -name '*.db' -delete
