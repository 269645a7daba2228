#!/bin/bash
# K-step pipelined across the barrier: correctness, micro-benchmark and whole-step A/B against the previous library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p16; mkdir -p $out
L=$GRAFT_REPO_ROOT/q-diffusion_amd/lib
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
timeout 600 python -m pytest tests/test_engine_models.py -m gpu -q -k "sd_tiny or cifar_tiny or graph" > $out/pytest_models.log 2>&1; echo "models rc=$?"; tail -2 $out/pytest_models.log
export IGEMM_ONLY='c3 320->320 @64|c3 640->640 @32|c3 1280->1280 @16|geglu|c1 320->320'
for v in "" prev bd2 bd4; do
  echo "--- igemm micro [$v]"; if [ -n "$v" ]; then export QDIFF_HIP_LIB=$L/libqdiff_hip_$v.so; else unset QDIFF_HIP_LIB; fi
  python tools/bench_igemm.py 4 10 2>&1 | grep -v amdgpu.ids
done
unset QDIFF_HIP_LIB
tools/r02_ab.sh "" "QDIFF_HIP_LIB=$L/libqdiff_hip_prev.so" "QDIFF_HIP_LIB=$L/libqdiff_hip_bd2.so" "QDIFF_HIP_LIB=$L/libqdiff_hip_bd4.so" ""
