#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/q-diffusion_amd/lib
tools/r02_ab.sh "" "QD_GEGLU_MT=2" "QDIFF_HIP_LIB=$L/libqdiff_hip_mocc3.so" "QDIFF_HIP_LIB=$L/libqdiff_hip_mocc4.so" "QD_TILE_MT=1" ""
