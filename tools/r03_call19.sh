#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c19; mkdir -p $out
SH="16,320,64,320,1,1;16,640,32,640,1,1;16,1280,16,1280,1,1;16,320,64,320,3,1;16,640,32,640,3,1"
for o in fp32 fp16 fp32 fp16; do echo "== IGEMM_OUT=$o"; IGEMM_OUT=$o IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 30 2>&1 | tail -6; done | tee $out/igemm_out_dtype.txt
QDIFF_STREAM=fp16 timeout 300 python tools/layer_times.py 8 2>&1 | head -24 | tee $out/layer_times_fp16.txt
