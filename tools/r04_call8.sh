#!/bin/bash
# Round 4, GPU call 8: MFMA-pipe occupancy per kernel over two steady-state SD evaluations (eager launches, context prepared):
# one PMC pass, counters only (+ the kernel trace rocprofv3 needs for names)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_c8; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o mfma -- python tools/eval_breakdown.py run sd 8 2 pin > $out/pmc.log 2>&1
python tools/pmc_table.py $out/pmc igemm attn_ splitk gn_apply ln_quant > $out/pmc_eval_mfma_busy.txt 2>&1
grep -E "^igemm|^attn|MFMA busy" $out/pmc_eval_mfma_busy.txt | head -80
find $out -name '*.csv' -size +1M -delete; find $out -name '*.db' -delete
