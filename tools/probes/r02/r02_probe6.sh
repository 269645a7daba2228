#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p6; mkdir -p $out
python tools/bench_attn.py 5 > $out/bench_attn.txt 2>&1; cat $out/bench_attn.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "attention" > $out/pytest_attn.log 2>&1; tail -3 $out/pytest_attn.log
timeout 900 python -m pytest tests/test_block_parity.py -m gpu -q -s -k "sd_full or sd_tiny" > $out/pytest_blk.log 2>&1; tail -3 $out/pytest_blk.log
grep -E "^\[sd_" $out/pytest_blk.log | cut -c1-260 | head -40
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-denominators > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-400
