#!/bin/bash
# lean attention / packed epilogues / skip branch: correctness first, then A/B on this box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p14; mkdir -p $out
L=$GRAFT_REPO_ROOT/q-diffusion_amd/lib
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_engine_models.py tests/test_block_parity.py -m gpu -q -k "sd_tiny or sd_full or graph or plms or ldm_tiny" > $out/pytest_models.log 2>&1; echo "models rc=$?"; tail -3 $out/pytest_models.log
echo "--- attention micro (new lean)"; python tools/bench_attn.py 5 2>&1 | tail -5
echo "--- attention micro (QD_ATTN_LEAN=0)"; QD_ATTN_LEAN=0 python tools/bench_attn.py 5 "d40" 2>&1 | tail -2
echo "--- attention micro (lean, 3 blocks/CU)"; QDIFF_HIP_LIB=$L/libqdiff_hip_aocc3.so python tools/bench_attn.py 5 "d40" 2>&1 | tail -2
tools/r02_ab.sh "" "QDIFF_HIP_LIB=$L/libqdiff_hip_prev.so" "QDIFF_SKIP_BRANCH=0" "QD_ATTN_LEAN=0" "QDIFF_HIP_LIB=$L/libqdiff_hip_aocc3.so" ""
