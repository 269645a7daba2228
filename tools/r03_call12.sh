#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c12; mkdir -p $out
for flat in 0 1; do for e in 0 1 2; do echo "== FLAT=$flat QD_ATTN_PIPE=$e  $(BENCH_ATTN_FLAT=$flat QD_ATTN_PIPE=$e timeout 200 python tools/bench_attn.py 5 'sd self 64x64' 2>&1 | tail -1)"; done; done | tee $out/bench_attn_flat.txt
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES"
for set in A B; do
  ctr=$([ $set = A ] && echo "$A" || echo "$B")
  BENCH_ATTN_FLAT=1 QD_ATTN_PIPE=2 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc_lds_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $out/pmc_lds_$set.log 2>&1
done
python tools/pmc_table.py $out attn > $out/pmc_attn_table.txt 2>&1; cat $out/pmc_attn_table.txt
find $out -name '*.db' -delete
