#!/bin/bash
# rocprofv3 kernel stats of a short bench run; usage: prof_bench.sh <batch> <streams> [steps]   (run on the GPU box)
B=${1:-4}; S=${2:-2}; N=${3:-40}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_b${B}s${S}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --batch $B --streams $S --steps $N --warmup 5 --repeats 2 --no-cpu-baseline --no-extra-legs > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $OUT/b_kernel_trace.csv > $OUT/by_grid.txt
rm -f $OUT/b_kernel_trace.csv
head -45 $OUT/by_grid.txt
grep -o '"value": [0-9.]*' $OUT/bench.log
