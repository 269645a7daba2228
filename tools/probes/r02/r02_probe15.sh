#!/bin/bash
# kernel tests, per-dispatch timeline of one graph-replayed evaluation, PMC passes (attention d=40, 3x3 convolutions)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p15; mkdir -p $out
L=$GRAFT_REPO_ROOT/q-diffusion_amd/lib
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q > $out/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $out/pytest_kernels.log
rocprofv3 --kernel-trace -d $out -o eb -- python tools/eval_breakdown.py run sd 8 3 graph > $out/eb.log 2>&1
python tools/eval_breakdown.py join $out/eb_results.db 3 > $out/eval_breakdown_graph.txt 2>&1
python tools/eval_breakdown.py timeline $out/eb_results.db 3 $out/timeline.tsv
head -12 $out/eval_breakdown_graph.txt | cut -c1-150
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc_attn$i -o p -- python tools/bench_attn.py 3 "self 64x64" > $out/pmc_attn$i.log 2>&1
  IGEMM_ONLY='c3 320->320 @64|c3 640->640 @32|c3 1280->1280 @16|geglu|c1 320->320' rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc_igemm$i -o p -- python tools/bench_igemm.py 4 3 > $out/pmc_igemm$i.log 2>&1
done
python tools/pmc_table.py $out attn > $out/pmc_attn_table.txt 2>&1
python tools/pmc_table.py $out igemm > $out/pmc_igemm_table.txt 2>&1
IGEMM_ONLY='c3 320->320 @64|c3 640->640 @32|c3 1280->1280 @16|geglu|c1 320->320' python tools/bench_igemm.py 4 10 > $out/bench_igemm.txt 2>&1
find $out -name '*.db' -delete; find $out -name '*.csv' -size +2M -delete
tools/r02_ab.sh "" "QDIFF_HIP_LIB=$L/libqdiff_hip_prev.so" ""
