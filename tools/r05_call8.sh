#!/bin/bash
# Round 5, GPU call 8: head-layout epilogues behind int8 weights (the DDIM AttnBlock) — parity, CIFAR A/B, split-K target on CIFAR / LDM, LDM fp16 stream.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_c8
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine_models.py tests/test_block_parity.py -m gpu -x -q \
  -k "projection_heads_epilogue or ldm_attention_qkv or (blocks_teacher_forced and cifar) or (quantised_unet_matches_reference and cifar) or (hip_graph_replay_equals_eager) or (foreign_model_classes and cifar) or (packed_checkpoint and cifar)" \
  > $O/pytest_targeted.log 2>&1; echo "pytest rc=$?" >> $O/pytest_targeted.log; tail -5 $O/pytest_targeted.log
C="python bench.py --model cifar --images-per-gpu 64 --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
L="python bench.py --model ldm --images-per-gpu 64 --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5"
one() { name=$1; shift; echo "== $name" >> $O/ab.log; ( "$@" ) >> $O/ab.log 2>> $O/ab.err; }
for rep in 1 2; do
  one "cifar heads=1 rep$rep" env QDIFF_QKV_HEADS=1 $C
  one "cifar heads=0 rep$rep" env QDIFF_QKV_HEADS=0 $C
done
for t in 256 512 1024; do one "cifar splitk-target $t" env QD_SPLITK_TARGET=$t $C; done
for t in 256 512; do one "ldm splitk-target $t" env QD_SPLITK_TARGET=$t $L; done
one "ldm fp16 stream" $L --stream fp16
python - <<'PY' > $O/ab_summary.txt
import json
name=None
for ln in open("gpurun_out/r05_c8/ab.log"):
    if ln.startswith("=="): name=ln.strip(); continue
    if ln.startswith("{"):
        d=json.loads(ln); r=d.get("roofline",{})
        print(name, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "igemm_ms", r.get("igemm_ms_per_eval"), "frac", r.get("frac"), "launches", r.get("launches_per_eval"))
PY
cat $O/ab_summary.txt
timeout 300 rocprofv3 --kernel-trace -d $O -o evb_cifar -- python tools/eval_breakdown.py run cifar 64 3 graph > $O/evb_cifar.log 2>&1
db=$(find $O -name "evb_cifar_results.db" | head -1)
python tools/eval_breakdown.py join $db 3 > $O/cifar_eval_breakdown_graph.txt; head -12 $O/cifar_eval_breakdown_graph.txt | cut -c1-150
find $O -name '*.db' -delete
