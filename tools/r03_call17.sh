#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c17; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "fp16 or conv or linear or splitk" > $out/pytest_k.log 2>&1; echo "kernel tests rc=$?"; tail -4 $out/pytest_k.log
timeout 900 python -m pytest tests/test_engine_models.py -m gpu -q -s -k "fp16_activation or foreign" > $out/pytest_m.log 2>&1; echo "model tests rc=$?"; grep -E "fp16 stream|passed|failed|Error" $out/pytest_m.log | tail -14
timeout 900 python -m pytest tests/test_calibration.py -m gpu -q -s > $out/pytest_c.log 2>&1; echo "calibration rc=$?"; grep -E "AdaRound|passed|failed|^E " $out/pytest_c.log | tail -12
tools/r02_ab.sh "QDIFF_STREAM=fp32" "QDIFF_STREAM=fp16" "QDIFF_STREAM=fp32" "QDIFF_STREAM=fp16" 2>&1 | tee $out/sd_ab.txt
