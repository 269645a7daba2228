#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c21; mkdir -p $out
SH="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1920,32,640,3,1;16,640,64,640,3,1"
for sgr in 0 1 2 3 4 0; do echo "== QD_STAGGER=$sgr"; QD_STAGGER=$sgr IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -8; done | tee $out/igemm_stagger.txt
