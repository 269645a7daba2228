#!/bin/bash
# GPU box: 128x320 tile policy A/B, CIFAR / LDM lines with the planned concatenations.  Output: gpurun_out/c2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/c2; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "wide or concatenation or groupnorm_stat or conv_fp32" > $out/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $out/pytest_k.log
SH="16,320,64,320,3,1;16,640,64,320,3,1;16,960,64,320,3,1;16,640,32,640,3,1;16,1280,32,640,3,1;16,1920,32,640,3,1"
for e in "QD_WIDE_TILE=0" "QD_WIDE_TILE=1" "QD_WIDE_TILE=2"; do
  echo "== igemm $e"; env $e IGEMM_SHAPES="$SH" timeout 200 python tools/bench_igemm.py 4 20 2>&1 | tail -7
done > $out/igemm_ab.txt 2>&1
cat $out/igemm_ab.txt
tools/r02_ab.sh "QD_WIDE_TILE=0" "QD_WIDE_TILE=1" "QD_WIDE_TILE=2" 2>&1 | tee $out/sd_ab.txt
for m in cifar ldm; do
  for e in "QDIFF_CAT_SLOTS=0 QDIFF_FUSE_SKIP_QUANT=0" "QDIFF_CAT_SLOTS=1"; do
    env $e timeout 300 python bench.py --model $m --images-per-gpu 64 --no-cpu-baseline --no-denominators > $out/bench_$m.json 2> $out/bench_$m.err
    echo "[$m $e] $(python -c "import json;d=json.loads(open('$out/bench_$m.json').read().strip().splitlines()[-1]);print(d['value'],d['unit'],d['ms_per_step'])" 2>&1 | tail -1)"
  done
done 2>&1 | tee $out/other_ab.txt
