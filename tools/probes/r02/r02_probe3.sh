#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p3; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -15 $out/pytest.log
grep -E "^\[(cifar|ldm|sd)_(full)" $out/pytest.log | head -60
for mt in 1 2; do
  QD_TILE_MT=$mt rocprofv3 --kernel-trace -d $out -o lp$mt -- python tools/layer_prof.py run $out/layers$mt.json 8 > $out/lp$mt.log 2>&1
  python tools/layer_prof.py join $out/layers$mt.json $out/lp${mt}_results.db > $out/layer_table_mt$mt.txt 2>&1
  head -1 $out/layer_table_mt$mt.txt
done
find $out -name '*.db' -delete
