#!/bin/bash
# Round 6, GPU call 2: where the three-launch attention spends its time — blocks-per-CU sweep (LDS padding variants), per-kernel
# durations (kernel trace) and the SQ counters of the default build.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c2
mkdir -p $O
L=q-diffusion_amd/lib
for rep in 1 2; do
  for v in "" _pad4 _pad3 _pad56; do
    echo "== lib$v flat rep=$rep" >> $O/occ_sweep.txt
    QDIFF_HIP_LIB=$PWD/$L/libqdiff_hip$v.so BENCH_ATTN_FLAT=1 timeout 300 python tools/bench_attn.py 10 "sd self 64x64" >> $O/occ_sweep.txt 2>> $O/err.txt
  done
done
cat $O/occ_sweep.txt
for v in "" _pad56; do
BENCH_ATTN_FLAT=1 QDIFF_HIP_LIB=$PWD/$L/libqdiff_hip$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$v -o t -- python tools/bench_attn.py 5 "sd self 64x64" > $O/trace$v.log 2>&1
f=$(find $O/trace$v -name "*kernel_stats.csv" | head -1); echo "== kernel stats lib$v"; head -8 "$f" | cut -c1-220
done
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES"
for set in A B; do
  ctr=$([ $set = A ] && echo "$A" || echo "$B")
  BENCH_ATTN_FLAT=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_attn -o pmc_attn_$set -- python tools/bench_attn.py 3 "sd self 64x64" > $O/pmc_attn_$set.log 2>&1
done
python tools/pmc_table.py $O/pmc_attn attn > $O/pmc_attn_table.txt 2>&1; cat $O/pmc_attn_table.txt | cut -c1-120
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
