#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_c5; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; echo "pytest attention rc=$?"; tail -3 $out/pytest_attn.log
for e in "QD_ATTN_PIPE=0 QD_ATTN_XCD=0" "QD_ATTN_PIPE=0 QD_ATTN_XCD=1" "QD_ATTN_PIPE=1 QD_ATTN_XCD=0" "QD_ATTN_PIPE=1 QD_ATTN_XCD=1"; do echo "== $e"; env $e timeout 200 python tools/bench_attn.py 5 2>&1 | tail -5; done | tee $out/bench_attn_ab.txt
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES TCC_HIT_sum TCC_MISS_sum"
for pipe in 0 1; do
  for set in A B; do
    ctr=$([ $set = A ] && echo "$A" || echo "$B")
    QD_ATTN_PIPE=$pipe timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc_p${pipe}_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $out/pmc_p${pipe}_$set.log 2>&1
  done
done
python tools/pmc_table.py $out attn > $out/pmc_attn_table.txt 2>&1; cat $out/pmc_attn_table.txt
find $out -name '*.db' -delete
