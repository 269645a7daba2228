#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/p10; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o eb -- python tools/eval_breakdown.py run sd 8 3 graph > $out/eb.log 2>&1
python tools/eval_breakdown.py join $out/eb_results.db 3 > $out/eval_breakdown_graph.txt 2>&1
cat $out/eval_breakdown_graph.txt | cut -c1-150 | head -45
find $out -name '*.db' -delete
