#!/bin/bash
# Round 6, GPU call 29: score MFMAs of the next tile inside the statistics chain of the current one (attn_stats_kernel): attention
# tests on the variant, A/B against the product library.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c29
mkdir -p $O
L=$PWD/q-diffusion_amd/lib
QDIFF_HIP_LIB=$L/libqdiff_hip_statspipe.so timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_random_shapes_gpu.py -m gpu -q -k "attention" > $O/pytest_variant.log 2>&1; echo "pytest rc=$?" >> $O/pytest_variant.log
tail -5 $O/pytest_variant.log
for rep in 1 2 3; do
  for v in product statspipe statspipe12; do
    lib=$L/libqdiff_hip_$v.so; [ $v = product ] && lib=$L/libqdiff_hip.so
    for flat in 0 1; do
      echo "== $v flat=$flat rep=$rep" >> $O/attn_ab.txt; BENCH_ATTN_FLAT=$flat QDIFF_HIP_LIB=$lib timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
    done
  done
done
paste - - < $O/attn_ab.txt | sed -E 's/sd self 64x64 d40 +BH= 128 T= 4096 S= 4096 d=  40//' | awk '{print $2, $3, $4, $5, $6}'
