#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p5; mkdir -p $out
python tools/bench_attn.py 5 > $out/bench_attn.txt 2>&1; cat $out/bench_attn.txt
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc$i -o p -- python tools/bench_attn.py 2 "self 64x64" > $out/pmc$i.log 2>&1
done
python tools/pmc_table.py $out attn_kernel > $out/pmc_table.txt 2>&1; cat $out/pmc_table.txt
find $out -name '*.csv' -size +2M -delete
