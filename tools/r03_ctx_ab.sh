#!/bin/bash
# Round 3: where the context branch forks (QDIFF_CTX_FORK = late | start | attn): whole-evaluation A/B on one box + timeline.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03ctx; mkdir -p $out
tools/r02_ab.sh "QDIFF_CTX_FORK=late" "QDIFF_CTX_FORK=attn" "QDIFF_CTX_FORK=start" "QDIFF_CTX_FORK=late" "QDIFF_CTX_FORK=attn" "QDIFF_CTX_FORK=start" 2>&1 | tee $out/sd_ab2.txt
QDIFF_CTX_FORK=attn rocprofv3 --kernel-trace -d $out -o eb -- python tools/eval_breakdown.py run sd 8 3 graph > $out/eb.log 2>&1
python tools/eval_breakdown.py timeline $out/eb_results.db 3 $out/sd_eval_timeline_ctx_attn.tsv
find $out -name '*.db' -delete
