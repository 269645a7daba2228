#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export QDIFF_HIP_LIB=$GRAFT_REPO_ROOT/q-diffusion_amd/lib/libqdiff_hip_dbg.so
export IGEMM_ONLY='c3 320->320 @64|c3 640->640 @32|c3 1280->1280 @16|geglu|c1 320->320|ff out|c3 960'
for dbg in 0 1 2 3; do
  echo "== QD_DBG=$dbg (bit0: no epilogue, bit1: one K-step only)"
  QD_DBG=$dbg python tools/bench_igemm.py 4 20 2>&1 | grep -v amdgpu.ids
done
