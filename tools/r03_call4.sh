#!/bin/bash
# Round 3, GPU call 4: where do the waves of the attention kernels wait?  PMC passes (no trace domains mixed in) over
# tools/bench_attn.py "sd self 64x64" for the unpipelined and the pipelined kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c4; mkdir -p $out
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES TCC_HIT_sum TCC_MISS_sum"
for pipe in 0 1; do
  for set in A B; do
    ctr=$([ $set = A ] && echo "$A" || echo "$B")
    QD_ATTN_PIPE=$pipe timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc_p${pipe}_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $out/pmc_p${pipe}_$set.log 2>&1
    echo "pipe=$pipe set=$set rc=$?"
  done
done
python tools/pmc_table.py $out attn > $out/pmc_attn_table.txt 2>&1; cat $out/pmc_attn_table.txt
find $out -name '*.db' -delete
