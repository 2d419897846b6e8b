#!/bin/bash
# PMC passes over a short bench.py run; per-kernel per-grid averages. usage: pmc_bench.sh "<set1>" "<set2>" ...
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/pmc_bench_raw; rm -rf $RAW; mkdir -p $RAW; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --streams 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-legs > $RAW/p$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python - <<'PY'
import csv, glob, collections, os, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bench"
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("/tmp/pmc_bench_raw/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
        key = (name[:40], r["Grid_Size"], r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k, (v, n) in sorted(agg.items()):
        line = "%-40s grid %-9s %-28s per-launch %14.1f (n=%d)" % (k[0], k[1], k[2], v / max(n, 1), n)
        print(line); fo.write(line + "\n")
PY
