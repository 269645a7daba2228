#!/bin/bash
# GPU box: pointwise prologue parity + A/B.  Output: gpurun_out/c4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/c4; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "conv or linear or projection or geglu or concatenation or temb" > $out/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $out/pytest_k.log
SH="16,320,64,320,1,1;16,640,32,640,1,1;16,1280,64,320,1,1;16,320,64,2560,1,1;16,960,64,320,1,1"
for e in "QD_POINTWISE=0" "QD_POINTWISE=1"; do
  echo "== igemm $e"; env $e IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -6
done | tee $out/igemm_ab.txt
tools/r02_ab.sh "QD_POINTWISE=0" "QD_POINTWISE=1" "QD_POINTWISE=0" "QD_POINTWISE=1" 2>&1 | tee $out/sd_ab.txt
