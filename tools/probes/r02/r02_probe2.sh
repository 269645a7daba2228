#!/bin/bash
# round-2 probe: GPU test suite + igemm micro-benchmark + per-layer table with the rewritten contraction kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p2; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -5 $out/pytest.log
grep -E "^\[(cifar|ldm|sd)_" $out/pytest.log | head -80
python tools/bench_igemm.py 4 10 > $out/bench_igemm.txt 2>&1; cat $out/bench_igemm.txt
rocprofv3 --kernel-trace -d $out -o lp -- python tools/layer_prof.py run $out/layers.json 8 > $out/lp.log 2>&1
python tools/layer_prof.py join $out/layers.json $out/lp_results.db > $out/layer_table.txt 2>&1
python tools/rocpd_stats.py $out/lp_results.db > $out/kernel_stats.txt 2>&1
head -50 $out/layer_table.txt
find $out -name '*.db' -delete
