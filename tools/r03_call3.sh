#!/bin/bash
# Round 3, GPU call 3: the software-pipelined lean attention kernel — bit-equality with the unpipelined one, oracle parity,
# micro-benchmark A/B and whole-evaluation A/B on one box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c3; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -12 $out/pytest_attn.log
for e in 0 1 0 1; do echo "== QD_ATTN_PIPE=$e"; QD_ATTN_PIPE=$e timeout 200 python tools/bench_attn.py 5 2>&1 | tail -5; done | tee $out/bench_attn_ab.txt
timeout 600 python -m pytest tests/test_block_parity.py tests/test_engine_models.py -m gpu -q -x -k "sd_tiny or sd_full or ldm_updown_tiny or churches" > $out/pytest_models.log 2>&1; echo "pytest models rc=$?"; tail -5 $out/pytest_models.log
tools/r02_ab.sh "QD_ATTN_PIPE=0" "QD_ATTN_PIPE=1" "QD_ATTN_PIPE=0" "QD_ATTN_PIPE=1" 2>&1 | tee $out/sd_ab.txt
