#!/bin/bash
# PMC passes over the dominant kernel only; run on the GPU box. usage: pmc_iter.sh "<set1>" "<set2>" ...
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_iter; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/run_iter_only.py 10 > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_iter"
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "iterate3" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k, (v, n) in sorted(agg.items()):
        line = "%-45s per-launch %16.1f  (n=%d)" % (k, v / max(n, 1), n)
        print(line); fo.write(line + "\n")
PY
