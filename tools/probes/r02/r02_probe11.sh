#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p11; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
echo "== new"; python tools/bench_attn.py 5 2>&1 | grep -v amdgpu
echo "== prev"; QDIFF_HIP_LIB=$GRAFT_REPO_ROOT/q-diffusion_amd/lib/libqdiff_hip_prev.so python tools/bench_attn.py 5 2>&1 | grep -v amdgpu
