#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c15; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -3 $out/pytest_attn.log
for e in 0 2; do echo "== QD_ATTN_PIPE=$e"; BENCH_ATTN_FLAT=1 QD_ATTN_PIPE=$e timeout 200 python tools/bench_attn.py 5 2>&1 | tail -5; done | tee $out/bench_attn_final.txt
L=q-diffusion_amd/lib
SH="16,320,64,320,3,1;16,960,64,320,3,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1280,16,1280,3,1;16,640,64,640,3,1;16,320,64,2560,1,1"
for v in "" noasum nounpack noasum_nounpack; do
  lib=$([ -z "$v" ] && echo "" || echo "$L/libqdiff_hip_$v.so")
  echo "== igemm lib=${v:-product}"; QDIFF_HIP_LIB=$lib IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -8
done | tee $out/igemm_ablation.txt
tools/r02_ab.sh "QD_ATTN_PIPE=0" "QD_ATTN_PIPE=2" "QD_ATTN_PIPE=0" "QD_ATTN_PIPE=2" 2>&1 | tee $out/sd_ab.txt
