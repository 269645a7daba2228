#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c20; mkdir -p $out
L=q-diffusion_amd/lib
SH="16,320,64,320,1,1;16,640,32,640,1,1;16,1280,16,1280,1,1;16,320,64,320,3,1;16,320,64,2560,1,1"
for v in "" noepi nokloop; do
  lib=$([ -z "$v" ] && echo "" || echo "$L/libqdiff_hip_$v.so")
  echo "== igemm lib=${v:-product}"; QDIFF_HIP_LIB=$lib IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 30 2>&1 | tail -6
done | tee $out/igemm_phase_ablation.txt
for mt in 1 2; do echo "== QD_TILE_MT=$mt"; QD_TILE_MT=$mt IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 30 2>&1 | tail -6; done | tee -a $out/igemm_phase_ablation.txt
