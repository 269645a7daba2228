#!/bin/bash
# GPU box: correctness of the planned-concatenation / raw-quant / fat-tile changes + A/B timings.  Output: gpurun_out/c1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/c1; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
# the fat tiles through the kernel parity tests (thresholds lowered so that the test shapes take them)
QD_FAT_TILE=2 QD_FAT_MINBLK=1 QD_FAT_MINK=64 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "conv or concatenation or groupnorm_stat" > $out/pytest_fat.log 2>&1; echo "pytest fat(256x320) rc=$?"; tail -3 $out/pytest_fat.log
QD_FAT_TILE=2 QD_FAT_MINBLK=5 QD_FAT_MINK=64 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "wide or concatenation" > $out/pytest_fat2.log 2>&1; echo "pytest fat(128x320) rc=$?"; tail -3 $out/pytest_fat2.log
SH="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,320,64,320,1,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1920,32,640,3,1"
for e in "QD_FAT_TILE=0" "QD_FAT_TILE=1" "QD_FAT_TILE=2"; do
  echo "== igemm $e"; env $e IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -8
done > $out/igemm_ab.txt 2>&1
cat $out/igemm_ab.txt
tools/r02_ab.sh "QDIFF_CAT_SLOTS=0 QDIFF_FUSE_SKIP_QUANT=0" "QDIFF_CAT_SLOTS=1" "QD_FAT_TILE=1" "QD_FAT_TILE=2" 2>&1 | tee $out/sd_ab.txt
