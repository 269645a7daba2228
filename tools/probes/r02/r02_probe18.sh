#!/bin/bash
# ablation of the contraction K-step (measurement-only builds: results are wrong by construction)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p18; mkdir -p $out
L=$GRAFT_REPO_ROOT/q-diffusion_amd/lib
export IGEMM_SHAPES="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,320,64,160,3,1;16,320,64,320,1,1;16,640,32,640,3,1;16,1280,16,1280,3,1"
for v in "" abl_NOMFMA abl_NOUNPACK abl_NODMA abl_NOMFMA_NODMA; do
  echo "--- [$v]"; if [ -n "$v" ]; then export QDIFF_HIP_LIB=$L/libqdiff_hip_$v.so; else unset QDIFF_HIP_LIB; fi
  python tools/bench_igemm.py 4 10 2>&1 | grep -v amdgpu.ids | tee -a $out/ablation.txt
done
