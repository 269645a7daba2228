#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p4; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o eb -- python tools/eval_breakdown.py run sd 8 3 > $out/eb.log 2>&1
python tools/eval_breakdown.py join $out/eb_results.db 3 > $out/eval_breakdown.txt 2>&1
cat $out/eval_breakdown.txt | cut -c1-170
for mt in 1 2 0; do
  QD_TILE_MT=$mt rocprofv3 --kernel-trace -d $out -o lp$mt -- python tools/layer_prof.py run $out/layers$mt.json 8 > $out/lp$mt.log 2>&1
  python tools/layer_prof.py join $out/layers$mt.json $out/lp${mt}_results.db > $out/layer_table_mt$mt.txt 2>&1
  head -1 $out/layer_table_mt$mt.txt
done
find $out -name '*.db' -delete
