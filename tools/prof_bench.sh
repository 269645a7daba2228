#!/bin/bash
# usage (GPU box): tools/prof_bench.sh <tag> [bench args...]  -> rocprofv3 kernel trace of bench.py + top-kernel table
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace -d gpurun_out/$tag -o sd -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph "$@" > gpurun_out/$tag/bench.log 2>&1
python tools/rocpd_stats.py gpurun_out/$tag/sd_results.db | head -${TOPN:-24} | cut -c1-140
