#!/bin/bash
# PMC passes (one counter set per pass) over (a) a short bench.py run -- per-kernel per-grid averages -- and (b) tools/run_batch_calls.py --
# exactly CALLS timed-workload steps, for the bytes of a whole pair.  usage: pmc_bench.sh "<set1>" "<set2>" ...      (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/pmc_bench_raw; rm -rf $RAW; mkdir -p $RAW; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench; rm -rf $OUT; mkdir -p $OUT
CALLS=2; BATCH=8
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extra-legs --no-batch16 > $RAW/p$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $RAW/c$i -o p -- python $GRAFT_REPO_ROOT/tools/run_batch_calls.py --batch $BATCH --calls $CALLS > $RAW/c$i.log 2>&1 || echo "calls pass $i ($set) failed/timeout"
done
python - $CALLS $BATCH <<'PY'
import csv, glob, collections, os, re, sys
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bench"
def name_of(r):
    return re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))[:44]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("/tmp/pmc_bench_raw/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        key = (name_of(r), r["Grid_Size"], r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k, (v, n) in sorted(agg.items()):
        line = "%-44s grid %-9s %-28s per-launch %14.1f (n=%d)" % (k[0], k[1], k[2], v / max(n, 1), n)
        print(line); fo.write(line + "\n")
tot = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("/tmp/pmc_bench_raw/c*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        key = (name_of(r), r["Counter_Name"])
        tot[key][0] += float(r["Counter_Value"]); tot[key][1] += 1
pairs = int(sys.argv[1]) * int(sys.argv[2])
with open(out + "/summary_calls.txt", "w") as fo:
    fo.write("# %d batched calls of %d pairs (tools/run_batch_calls.py): per kernel the SUM of the counter over the run and its launches\n" % (int(sys.argv[1]), int(sys.argv[2])))
    fo.write("pairs %d\n" % pairs)
    for k, (v, n) in sorted(tot.items()):
        fo.write("%-44s %-28s total %16.1f (launches=%d)\n" % (k[0], k[1], v, n))
PY
