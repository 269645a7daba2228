#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c9; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -12 $out/pytest_attn.log
for e in 0 2 0 2; do echo "== QD_ATTN_PIPE=$e"; QD_ATTN_PIPE=$e timeout 200 python tools/bench_attn.py 5 2>&1 | tail -5; done | tee $out/bench_attn_ab.txt
