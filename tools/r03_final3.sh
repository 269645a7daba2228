#!/bin/bash
# GPU box: round-3 closing run — the whole GPU suite, smoke, the driver's bench command, decoder kernel stats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03g; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 900 python bench.py > $out/bench_sd.json 2> $out/bench_sd.err; echo "bench rc=$?"; tail -c 900 $out/bench_sd.json; echo
rocprofv3 --kernel-trace -d $out -o dec -- python tools/decode_once.py 4 3 > $out/dec.log 2>&1; tail -1 $out/dec.log
python tools/rocpd_stats.py $out/dec_results.db --md > $out/decoder_kernel_stats.md 2>&1; head -14 $out/decoder_kernel_stats.md
find $out -name '*.db' -delete
