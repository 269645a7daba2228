#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03f; mkdir -p $out
timeout 900 python bench.py > $out/bench_sd.json 2> $out/bench_sd.err; echo "bench rc=$?"; tail -c 1200 $out/bench_sd.json; echo
