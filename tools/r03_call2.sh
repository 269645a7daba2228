#!/bin/bash
# Round 3, GPU call 2: instruction-issue micro-benchmark (what bounds the attention / contraction loops), per-shape launch table
# of one SD evaluation, and the full GPU suite after the clean-up (halo / FIN_VEC deleted, GN_ROWS=2 default, nearest-2x folded
# into the gather kernel, `late` tests un-gated).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c2; mkdir -p $out
timeout 120 tools/probes/bin/ubench_issue > $out/ubench_issue.txt 2>&1; echo "ubench rc=$?"; cat $out/ubench_issue.txt
timeout 600 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_gpu.log
timeout 300 python tools/layer_times.py 8 > $out/layer_times.txt 2>&1; echo "layer_times rc=$?"; cat $out/layer_times.txt
tools/r02_ab.sh "QDIFF_UPSAMPLE_FOLD=0" "QDIFF_UPSAMPLE_FOLD=1" 2>&1 | tee $out/sd_ab.txt
