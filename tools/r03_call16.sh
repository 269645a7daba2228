#!/bin/bash
# Round 3, GPU call 16: the full GPU suite (new: calibration vs the reference's fixtures on the GPU, foreign-class adoption)
# with the teacher-forced per-block report printed (-s) for profiles/r03_block_parity_report.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c16; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -s -x > $out/pytest_gpu_s.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu_s.log
grep -n "passed\|failed" $out/pytest_gpu_s.log | tail -3
