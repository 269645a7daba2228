#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03n1; mkdir -p $out
timeout 600 python -m pytest tests/test_first_stage_hip.py -m gpu -q -s -x 2>&1 | tail -40 > $out/tests.txt; cat $out/tests.txt | tail -25
timeout 300 python tools/bench_decoder.py 4 5 > $out/decoder_shapes.txt 2>&1; cat $out/decoder_shapes.txt
