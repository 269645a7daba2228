#!/bin/bash
# First GPU call of the next round: parity of everything this round could not verify on a GPU (the experimental 3x3 halo
# kernel and its up-sampling fold, GPU calibration), then the A/B numbers that decide whether the halo kernel becomes default.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_first; mkdir -p $out
QDIFF_HALO=1 timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "halo" > $out/pytest_halo.log 2>&1; echo "halo parity rc=$?"; tail -4 $out/pytest_halo.log
# everything marked `late` (collected after the parity tests of the round's last GPU run): GPU calibration, the modulated
# GroupNorm (qd_groupnorm_mod_silu_quant), the LSUN-Churches attention shapes, ldm_updown_tiny / churches_full whole-UNet and
# teacher-forced blocks — WITHOUT -x, so that one failure does not hide the others
QDIFF_RUN_LATE=1 timeout 900 python -m pytest tests -m "gpu and late" -q > $out/pytest_late.log 2>&1; echo "late tests rc=$?"; tail -8 $out/pytest_late.log
timeout 300 python bench.py --model churches --steps 10 --warmup 2 > $out/bench_churches.json 2> $out/bench_churches.err; echo "churches bench rc=$?"; tail -c 1500 $out/bench_churches.json
timeout 120 python tools/bench_fakequant.py 2>&1 | tail -2 | tee $out/fakequant.txt
SH="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1280,16,1280,3,1"
for e in "QDIFF_HALO=0" "QDIFF_HALO=1"; do
  echo "== igemm $e"; env $e IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -7
done | tee $out/igemm_halo_ab.txt
tools/r02_ab.sh "QDIFF_HALO=0" "QDIFF_HALO=1" "QDIFF_HALO=0" "QDIFF_HALO=1" 2>&1 | tee $out/sd_halo_ab.txt
# experimental multi-row GroupNorm apply pass (csrc/norm_quant.hip gn_apply_rows_kernel): parity through the existing tests, then A/B
for u in 2 4; do
  QD_GN_ROWS=$u timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "groupnorm or concatenation" > $out/pytest_gnrows$u.log 2>&1; echo "gn rows=$u parity rc=$?"; tail -2 $out/pytest_gnrows$u.log
done
tools/r02_ab.sh "QD_GN_ROWS=0" "QD_GN_ROWS=2" "QD_GN_ROWS=4" "QD_GN_ROWS=0" 2>&1 | tee $out/sd_gnrows_ab.txt
# experimental vectorised split-K second pass (csrc/igemm_dma.hip splitk_finalize4_kernel): parity (bit-identical to unsplit), then A/B
QD_FIN_VEC=1 timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "splitk" > $out/pytest_finvec.log 2>&1; echo "fin vec parity rc=$?"; tail -2 $out/pytest_finvec.log
tools/r02_ab.sh "QD_FIN_VEC=0" "QD_FIN_VEC=1" "QD_FIN_VEC=0" "QD_FIN_VEC=1" 2>&1 | tee $out/sd_finvec_ab.txt
