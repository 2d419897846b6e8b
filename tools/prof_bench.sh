#!/bin/bash
# rocprofv3 kernel stats of a short bench run; usage: prof_bench.sh <pairs> [steps]   (run on the GPU box)
P=${1:-3}; S=${2:-60}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_p$P; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --pairs $P --steps $S --warmup 10 --no-cpu-baseline > $OUT/bench.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
    print("%-34s calls %6s  avg %8.1f us  total %9.1f us  %5s%%" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
grep -o '"value": [0-9.]*' $OUT/bench.log
