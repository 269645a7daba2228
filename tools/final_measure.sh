#!/bin/bash
# usage (GPU box): tools/final_measure.sh <tag>  -> bench line, rocprofv3 kernel-trace stats of the same command, HBM PMC passes
tag="$1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace -d $out -o kt -- $CMD > $out/kt.log 2>&1
python tools/rocpd_stats.py $out/kt_results.db --md > $out/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $out/pmc_$c.log 2>&1
done
python - $out <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/**/pmc_{c}_counter_collection.csv", recursive=True)
    if not f:
        continue
    tot = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = "igemm" if ("igemm" in r["Kernel_Name"] or "splitk_finalize" in r["Kernel_Name"]) else ("attn" if "attn_kernel" in r["Kernel_Name"] else "other")
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    res[c] = {k: dict(sum_kb=tot[k], dispatches=n[k]) for k in tot}
json.dump(res, open(f"{out}/pmc_hbm.json", "w"), indent=1)
print(json.dumps(res))
PY
head -40 $out/kernel_stats.md | cut -c1-150
# keep the summaries only: the raw databases exceed what gpurun copies back
find $out -name '*.db' -delete; find $out -name '*.csv' -delete
