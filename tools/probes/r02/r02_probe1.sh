#!/bin/bash
# round-2 baseline probe (GPU box): per-layer table of one SD evaluation + PMC passes on the dominant shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p1; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o lp -- python tools/layer_prof.py run $out/layers.json 8 > $out/lp.log 2>&1
python tools/layer_prof.py join $out/layers.json $out/lp_results.db > $out/layer_table.txt 2>&1
python tools/rocpd_stats.py $out/lp_results.db > $out/kernel_stats.txt 2>&1
export IGEMM_ONLY='c3 320->320 @64|c3 640->640 @32|c3 1280->1280 @16|geglu|c1 320->320'
python tools/bench_igemm.py 4 10 > $out/bench_igemm.txt 2>&1
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc$i -o p -- python tools/bench_igemm.py 4 3 > $out/pmc$i.log 2>&1
done
python tools/pmc_table.py $out igemm > $out/pmc_table.txt 2>&1
timeout 400 python tools/gpu_denoms.py sd 8 > $out/denoms.txt 2>&1
find $out -name '*.db' -delete; find $out -name '*.csv' -size +2M -delete
cat $out/layer_table.txt | head -60; cat $out/bench_igemm.txt; tail -3 $out/denoms.txt
