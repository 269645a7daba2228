#!/bin/bash
# GPU box: round-3 final records under gpurun_out/r03f (copied to profiles/r03f_* by hand afterwards)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03f; mkdir -p $out
# 1. the driver's command, with denominators, CPU baseline and the first-stage decode timing
timeout 900 python bench.py > $out/bench_sd.json 2> $out/bench_sd.err; tail -c 700 $out/bench_sd.json; echo
# 2. per-kernel breakdown of the graph-replayed evaluation + eager kernel stats of the bench command
rocprofv3 --kernel-trace -d $out -o eb -- python tools/eval_breakdown.py run sd 8 3 graph > $out/eb.log 2>&1
python tools/eval_breakdown.py join $out/eb_results.db 3 > $out/sd_eval_breakdown_graph.txt 2>&1; head -8 $out/sd_eval_breakdown_graph.txt
python tools/eval_breakdown.py timeline $out/eb_results.db 3 $out/sd_eval_timeline.tsv
rocprofv3 --kernel-trace -d $out -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-denominators --no-graph > $out/kt.log 2>&1
python tools/rocpd_stats.py $out/kt_results.db --md > $out/sd_b16_kernel_stats.md 2>&1
# 3. HBM traffic of steady-state evaluations (separate PMC passes, eager)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python tools/eval_breakdown.py run sd 8 2 > $out/pmc_$c.log 2>&1
done
QD_COMMIT=${QD_COMMIT:-unknown} python tools/pmc_eval_traffic.py $out 2 $out/sd_igemm_hbm_traffic.json sd > $out/traffic.log 2>&1; tail -c 400 $out/traffic.log; echo
# 4. the other two BASELINE configurations (C2 CIFAR W8A8, C3 LDM-4 W4A8)
for m in cifar ldm; do
  timeout 600 python bench.py --model $m --images-per-gpu 64 --no-cpu-baseline > $out/bench_$m.json 2> $out/bench_$m.err; tail -c 300 $out/bench_$m.json; echo
done
find $out -name '*.db' -delete; find $out -name '*.csv' -size +1M -delete
# 5. the GPU test suite and the smoke entry
timeout 1500 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
