#!/bin/bash
# usage: tools/pmc_run.sh "<counters>" <outname> -- <command...>   (run on the GPU box; one PMC pass per call)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ctrs="$1"; name="$2"; shift 3
mkdir -p gpurun_out/pmc
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o "$name" -- "$@" > gpurun_out/pmc/$name.log 2>&1 || true
f=$(find gpurun_out/pmc -name "${name}_counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "igemm" not in k and "attn" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v / cnt[(k, c)]:16.1f}  (avg over {cnt[(k,c)]} dispatches)")
PY
