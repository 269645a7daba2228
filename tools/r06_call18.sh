#!/bin/bash
# Round 6, GPU call 18: spacing of the deferred P.V MFMAs (vector instructions between two of them: 0 = all six MFMAs in one cluster
# at the head of the tile, 8, 15, 22), bench_attn A/B in one call.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_c18
mkdir -p $O
L=$PWD/q-diffusion_amd/lib
for rep in 1 2 3; do
  for v in product pvdefer0 pvdefer8 pvdefer pvdefer22; do
    lib=$L/libqdiff_hip_$v.so; [ $v = product ] && lib=$L/libqdiff_hip.so
    echo "== $v rep=$rep" >> $O/attn_ab.txt; QDIFF_HIP_LIB=$lib timeout 300 python tools/bench_attn.py 10 "self 64x64" 2>/dev/null >> $O/attn_ab.txt
  done
done
cat $O/attn_ab.txt | paste - - | awk '{print $2, $3, $13, $14}'
