#!/bin/bash
# GPU box: attention key-sum table (QD_ATTN_KZ) parity + A/B.  Output: gpurun_out/c3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/c3; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention or wide or projection_heads" > $out/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $out/pytest_k.log
timeout 600 python -m pytest tests/test_block_parity.py -m gpu -q -k "sd" > $out/pytest_b.log 2>&1; echo "pytest block parity sd rc=$?"; tail -3 $out/pytest_b.log
for e in "QD_ATTN_KZ=0" "QD_ATTN_KZ=1"; do echo "== attn $e"; env $e timeout 200 python tools/bench_attn.py 5 sd 2>&1 | tail -4; done | tee $out/attn_ab.txt
tools/r02_ab.sh "QD_ATTN_KZ=0" "QD_ATTN_KZ=1" 2>&1 | tee $out/sd_ab.txt
