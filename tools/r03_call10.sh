#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c10; mkdir -p $out
timeout 120 tools/probes/bin/ubench_issue s2 > $out/ubench_s2.txt 2>&1; cat $out/ubench_s2.txt
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -4 $out/pytest_attn.log; grep -E "AssertionError: \(" $out/pytest_attn.log | head
