#!/bin/bash
# per-kernel time of ONE timed-workload step (2 x B gray LUTs + one batched call of B pairs): rocprofv3 kernel trace of tools/run_batch_calls.py,
# the first call (warm-up, allocations) dropped.  usage (GPU box): tools/call_breakdown.sh [batch] [calls] [opts]   -> gpurun_out/call_breakdown.txt
cd /tmp && export TMPDIR=/tmp
B=${1:-8}; K=${2:-6}; OPTS=${3:-}
RAW=/tmp/call_bd; rm -rf $RAW; mkdir -p $RAW $GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python $GRAFT_REPO_ROOT/tools/run_batch_calls.py --batch $B --calls $K $OPTS > $RAW/log 2>&1 || { echo "trace failed"; tail -5 $RAW/log; exit 1; }
python3 - $B $K <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/call_breakdown.txt
import csv, glob, collections, re, sys
B, K = int(sys.argv[1]), int(sys.argv[2])
rows = []
for f in glob.glob("/tmp/call_bd/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))[:52]
# a step begins with its first gray LUT after a non-LUT kernel
steps, cur, prev_lut = [], [], False
for r in rows:
    n = nm(r)
    if "at::native" in n or "rocclr" in n:
        continue
    lut = n.startswith("gray_lut")
    if lut and not prev_lut and cur:
        steps.append(cur); cur = []
    cur.append(r); prev_lut = lut
if cur: steps.append(cur)
steps = steps[1:]  # warm-up
agg = collections.defaultdict(lambda: [0.0, 0])
span = 0.0
for s in steps:
    span += (int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3
    for r in s:
        k = (nm(r), "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
        agg[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; agg[k][1] += 1
n = len(steps)
busy = sum(v[0] for v in agg.values())
print("# %d steps of %d pairs: GPU span per step %.1f us (= %.1f pairs/s), kernel time per step %.1f us" % (n, B, span / n, B * n / span * 1e6, busy / n))
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-52s grid %-16s launches/step %5.1f  avg %8.1f us  per step %8.1f us  %5.1f %%" % (k[0], k[1], c / n, t / c, t / n, 100 * t / busy))
PY
