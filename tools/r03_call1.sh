#!/bin/bash
# Round 3, GPU call 1: first hardware execution of the three knob-gated kernels of round 2 (3x3 halo slab, multi-row
# GroupNorm apply, vectorised split-K finalise): parity, then A/B inside ONE box. What wins becomes default, what loses is deleted.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c1; mkdir -p $out
QDIFF_HALO=1 timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "halo" > $out/pytest_halo.log 2>&1; echo "halo parity rc=$?"; tail -15 $out/pytest_halo.log
SH="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1280,16,1280,3,1;16,2560,16,1280,3,1;16,640,64,640,3,1"
for e in "QDIFF_HALO=0" "QDIFF_HALO=1" "QDIFF_HALO=0" "QDIFF_HALO=1"; do
  echo "== igemm $e"; env $e IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -9
done | tee $out/igemm_halo_ab.txt
for u in 2 4; do
  QD_GN_ROWS=$u timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "groupnorm or concatenation" > $out/pytest_gnrows$u.log 2>&1; echo "gn rows=$u parity rc=$?"; tail -2 $out/pytest_gnrows$u.log
done
QD_FIN_VEC=1 timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "splitk" > $out/pytest_finvec.log 2>&1; echo "fin vec parity rc=$?"; tail -2 $out/pytest_finvec.log
tools/r02_ab.sh "QDIFF_HALO=0" "QDIFF_HALO=1" "QD_GN_ROWS=2" "QD_GN_ROWS=4" "QD_FIN_VEC=1" "QDIFF_HALO=0" "QDIFF_HALO=1" 2>&1 | tee $out/sd_ab.txt
cp gpurun_out/ab/run*.json gpurun_out/ab/run*.err $out/ 2>/dev/null
