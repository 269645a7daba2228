#!/bin/bash
# Round 5, profiles of the final HEAD (split-K target 256, operand epilogues behind the LDM / DDIM attention blocks):
# steady-state breakdowns of the three configurations, kernel-trace stats of the bench command, HBM traffic of the fp32 stream.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05f3; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace -d $out -o evb -- python tools/eval_breakdown.py run sd 8 3 graph pin > $out/evb.log 2>&1
db=$(find $out -name 'evb_results.db' | head -1)
python tools/eval_breakdown.py join $db 3 > $out/sd_eval_breakdown_graph.txt; head -8 $out/sd_eval_breakdown_graph.txt | cut -c1-150
python tools/eval_breakdown.py timeline $db 3 $out/sd_eval_timeline.tsv
for m in cifar ldm; do
  timeout 300 rocprofv3 --kernel-trace -d $out -o evb_$m -- python tools/eval_breakdown.py run $m 64 3 graph > $out/evb_$m.log 2>&1
  python tools/eval_breakdown.py join $(find $out -name "evb_${m}_results.db" | head -1) 3 > $out/${m}_eval_breakdown_graph.txt; head -6 $out/${m}_eval_breakdown_graph.txt | cut -c1-150
done
timeout 400 rocprofv3 --kernel-trace -d $out -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-denominators --no-extras > $out/kt.log 2>&1
python tools/rocpd_stats.py $(find $out -name 'kt_results.db' | head -1) --md > $out/sd_bench_kernel_stats.md 2>&1; head -8 $out/sd_bench_kernel_stats.md | cut -c1-150
tail -1 $out/kt.log | cut -c1-300
find $out -name '*.db' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_hbm_fp32 -o pmc_$c -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc_fp32_$c.log 2>&1
done
QD_COMMIT=$QD_COMMIT python tools/pmc_eval_traffic.py $out/pmc_hbm_fp32 2 $out/sd_igemm_hbm_traffic_fp32.json | cut -c1-400
find $out -name '*.csv' -size +2M -delete; find $out -name '*.db' -delete
