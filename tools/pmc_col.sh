#!/bin/bash
# PMC passes over one batched Farneback call (tools/ab_iter.py --batch N); per-kernel per-grid averages of the counters.
# usage: pmc_col.sh [--size WxH] [--batch N] [--opts "k=v,..."] -- "<set1>" "<set2>" ...    summary -> gpurun_out/pmc_col/
SIZE=1920x1080; BATCH=8; OPTS=""
while [ "$1" != "--" ] && [ -n "$1" ]; do
  case "$1" in --size) SIZE=$2; shift 2;; --batch) BATCH=$2; shift 2;; --opts) OPTS=$2; shift 2;; *) break;; esac
done
[ "$1" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/pmc_col; rm -rf $RAW; mkdir -p $RAW
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_col; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o p -- python $GRAFT_REPO_ROOT/tools/ab_iter.py --size $SIZE --batch $BATCH --calls 1 "$OPTS" > $RAW/p$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python - "$RAW" "$OUT/summary_${SIZE}_b${BATCH}.txt" <<'PY'
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
        key = (name[:44], r["Grid_Size"], r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
with open(sys.argv[2], "w") as fo:
    for k, (v, n) in sorted(agg.items()):
        fo.write("%-44s grid %-9s %-32s per-launch %14.1f (n=%d)\n" % (k[0], k[1], k[2], v / max(n, 1), n))
PY
