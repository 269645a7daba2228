#!/bin/bash
# Round 6: steady-state profile of the SD evaluation at HEAD — per-kernel breakdown + timeline of the graph-replayed evaluation
# (kernel trace), the per-kernel MFMA-busy / VALU-per-MFMA table (one PMC pass over two eager evaluations, context prepared).
# QD_OUT names the output directory under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${QD_OUT:-r06_prof}; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace -d $out -o evb -- python tools/eval_breakdown.py run sd 8 3 graph pin > $out/evb.log 2>&1
db=$(find $out -name 'evb_results.db' | head -1)
python tools/eval_breakdown.py join $db 3 > $out/sd_eval_breakdown_graph.txt; head -45 $out/sd_eval_breakdown_graph.txt | cut -c1-150
python tools/eval_breakdown.py timeline $db 3 $out/sd_eval_timeline.tsv
find $out -name '*.db' -delete
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o mfma -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc.log 2>&1
python tools/pmc_table.py $out/pmc igemm attn_ splitk gn_apply ln_quant > $out/pmc_eval_mfma_busy.txt 2>&1
grep -E "^igemm|^attn|MFMA busy|VALU \(non" $out/pmc_eval_mfma_busy.txt | head -90
find $out -name '*.csv' -size +1M -delete; find $out -name '*.db' -delete
