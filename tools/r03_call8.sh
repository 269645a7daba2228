#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c8; mkdir -p $out
L=q-diffusion_amd/lib
for m in "" pipeS ablA16 pipeS16 ablA22 pipeS22; do
  v=$([ -z "$m" ] && echo "" || echo "$L/libqdiff_hip_$m.so")
  echo "== lib=${m:-product}  $(QDIFF_HIP_LIB=$v QD_ATTN_PIPE=1 timeout 200 python tools/bench_attn.py 5 'sd self 64x64' 2>&1 | tail -1)"
done | tee $out/attn_pipe_scalar.txt
