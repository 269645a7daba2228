#!/bin/bash
# GPU box: round-6 closing run — the whole GPU suite, smoke, the driver's bench command (headline + the child lines), steady-state
# breakdown + timeline of the graph-replayed evaluation (SD, CIFAR, LDM-4), kernel-trace stats of the bench command, HBM traffic
# PMC passes (fp32 stream), per-kernel MFMA-busy table, attention PMC passes.  QD_OUT names the output directory.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${QD_OUT:-r06f}; mkdir -p $out
timeout 1900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_sd.json 2> $out/bench_sd.err; echo "bench rc=$?"; tail -c 1200 $out/bench_sd.json; echo
# steady-state graph-replayed evaluations
for m in sd cifar ldm; do
  n=8; [ $m = cifar ] && n=64; [ $m = ldm ] && n=64
  timeout 600 rocprofv3 --kernel-trace -d $out -o evb_$m -- python tools/eval_breakdown.py run $m $n 3 graph pin > $out/evb_$m.log 2>&1
  db=$(find $out -name "evb_${m}_results.db" | head -1)
  python tools/eval_breakdown.py join $db 3 > $out/${m}_eval_breakdown_graph.txt; head -8 $out/${m}_eval_breakdown_graph.txt | cut -c1-150
  [ $m = sd ] && python tools/eval_breakdown.py timeline $db 3 $out/sd_eval_timeline.tsv
done
# kernel-trace stats of the bench command itself
timeout 600 rocprofv3 --kernel-trace -d $out -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-denominators --no-extras > $out/kt.log 2>&1
python tools/rocpd_stats.py $(find $out -name 'kt_results.db' | head -1) --md > $out/sd_bench_kernel_stats.md 2>&1; head -8 $out/sd_bench_kernel_stats.md | cut -c1-150
find $out -name '*.db' -delete
# HBM traffic (separate PMC passes, no other tracing domains)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_hbm_fp32 -o pmc_$c -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc_fp32_$c.log 2>&1
done
QD_COMMIT=$QD_COMMIT python tools/pmc_eval_traffic.py $out/pmc_hbm_fp32 2 $out/sd_igemm_hbm_traffic.json | cut -c1-400
# per-kernel MFMA-busy / VALU-per-MFMA table of the evaluation
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o mfma -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc.log 2>&1
python tools/pmc_table.py $out/pmc igemm attn_ splitk gn_apply ln_quant > $out/pmc_eval_mfma_busy.txt 2>&1
grep -E "^igemm|^attn|MFMA busy" $out/pmc_eval_mfma_busy.txt | head -30
# attention counters
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES"
for set in A B; do
  ctr=$([ $set = A ] && echo "$A" || echo "$B")
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_attn -o pmc_attn_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $out/pmc_attn_$set.log 2>&1
done
python tools/pmc_table.py $out/pmc_attn attn > $out/pmc_attn_table.txt 2>&1; head -5 $out/pmc_attn_table.txt
find $out -name '*.csv' -size +1M -delete; find $out -name '*.db' -delete
