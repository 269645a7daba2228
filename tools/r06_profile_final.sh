#!/bin/bash
# Round 6, final HEAD: the profile records once more on the code that ships (after the deferred P.V MFMAs): per-kernel breakdown and
# timeline of the graph-replayed SD evaluation (kernel trace), the per-kernel MFMA-busy table (one PMC pass), attention counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06e; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace -d $out -o evb -- python tools/eval_breakdown.py run sd 8 3 graph pin > $out/evb.log 2>&1
db=$(find $out -name 'evb_results.db' | head -1)
python tools/eval_breakdown.py join $db 3 > $out/sd_eval_breakdown_graph.txt; head -12 $out/sd_eval_breakdown_graph.txt | cut -c1-150
python tools/eval_breakdown.py timeline $db 3 $out/sd_eval_timeline.tsv
find $out -name '*.db' -delete
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --no-cpu-baseline --no-denominators --no-extras --steps 20 --warmup 5 > $out/kt.log 2>&1
f=$(find $out/kt -name '*kernel_stats*' | head -1); [ -n "$f" ] && head -25 "$f" > $out/sd_bench_kernel_stats.txt
find $out/kt -type f -size +1M -delete
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o mfma -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc.log 2>&1
python tools/pmc_table.py $out/pmc igemm attn_ splitk gn_apply ln_quant > $out/pmc_eval_mfma_busy.txt 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES"
for set in A B; do
  ctr=$([ $set = A ] && echo "$A" || echo "$B")
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_attn -o pmc_attn_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $out/pmc_attn_$set.log 2>&1
done
python tools/pmc_table.py $out/pmc_attn attn > $out/pmc_attn_table.txt 2>&1; grep -A30 "attn_pv_kernel<2, true, 2, true>" $out/pmc_attn_table.txt | head -32
find $out -name '*.csv' -size +1M -delete; find $out -name '*.db' -delete
