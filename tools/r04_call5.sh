#!/bin/bash
# Round 4, GPU call 5: fp16 operand mode of the first-stage decoder (kernel tests, decoder vs the reference golden, timing of
# fp16 / bf16 / library legs), transformer sub-layer parity at full SD shapes, prepared-context tests after the memset fix.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c5; mkdir -p $out
timeout 600 python -m pytest tests/test_first_stage.py tests/test_first_stage_hip.py -m gpu -q -s 2>&1 | grep -E "decoder|passed|failed|Error" | tail -12
for leg in fp16_autocast hip hip_bf16; do
  timeout 300 python bench.py --decode-leg $leg --images-per-gpu 8 2> $out/decode_$leg.err | tee -a $out/decode_legs.jsonl
done
timeout 900 python -m pytest tests/test_engine_models.py -m gpu -q -x -k "prepared_context or graph or plms" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_block_parity.py -m gpu -q -s -k "sd_tiny or sd_full" > $out/block_parity_sd.txt 2>&1; tail -3 $out/block_parity_sd.txt; grep "^\[sd" $out/block_parity_sd.txt | tail -40
