#!/bin/bash
# Round 4, GPU call 1: (a) legacy-shape MFMA issue rate (is K = 48 as 32 + 16 cheaper than K = 64?), (b) attention with the
# key-term table: parity tests + micro-benchmark A/B against the constant-operand MFMAs, (c) first-stage decode, every engine
# in its own process, 8 and 16 latents (round 3's last call faulted somewhere in there), (d) SD step A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c1; mkdir -p $out
timeout 120 tools/probes/bin/ubench_issue mfma > $out/ubench_mfma.txt 2>&1; tail -19 $out/ubench_mfma.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention or attn" 2>&1 | tail -5
for flat in 0 1; do for kt in 0 1 0 1; do
  echo "== FLAT=$flat KTAB=$kt"; BENCH_ATTN_FLAT=$flat QD_ATTN_KTAB=$kt timeout 200 python tools/bench_attn.py 5 "sd " 2>&1 | tail -4
done; done | tee $out/bench_attn_ab.txt
for n in 8 16; do for leg in fp32 bf16_autocast hip; do
  echo "== decode leg $leg, $n latents"
  timeout 400 python bench.py --decode-leg $leg --images-per-gpu $n 2> $out/decode_${leg}_$n.err | tee -a $out/decode_legs.jsonl
  echo "rc=$?"; tail -2 $out/decode_${leg}_$n.err
done; done
timeout 400 python -m pytest tests/test_first_stage.py tests/test_first_stage_hip.py -m gpu -q 2>&1 | tail -4
for kt in 0 1 0 1; do
  echo "== SD bench KTAB=$kt"
  QD_ATTN_KTAB=$kt timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-denominators --no-extras 2> $out/bench_sd_kt$kt.err | tee $out/bench_sd_kt$kt.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['igemm_ms_per_eval'])"
done
