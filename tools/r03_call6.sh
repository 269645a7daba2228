#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c6; mkdir -p $out
L=q-diffusion_amd/lib
for v in "" "$L/libqdiff_hip_abl_attn1.so" "$L/libqdiff_hip_abl_attn2.so"; do
  for pipe in 0 1; do
    echo "== lib=${v:-product} QD_ATTN_PIPE=$pipe"
    QDIFF_HIP_LIB=$v QD_ATTN_PIPE=$pipe timeout 200 python tools/bench_attn.py 5 "sd self 64x64" 2>&1 | tail -1
  done
done | tee $out/attn_load_ablation.txt
