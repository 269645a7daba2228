#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c7; mkdir -p $out
L=q-diffusion_amd/lib
for m in 0 1 2 4 6 8 16 17 22 31; do
  v=$([ $m = 0 ] && echo "" || echo "$L/libqdiff_hip_ablA$m.so")
  echo "== QD_ABL_ATTN=$m  $(QDIFF_HIP_LIB=$v QD_ATTN_PIPE=1 timeout 200 python tools/bench_attn.py 5 'sd self 64x64' 2>&1 | tail -1)"
done | tee $out/attn_pipe_ablation.txt
