#!/bin/bash
# rocprofv3 kernel stats of a short bench run; usage: prof_bench.sh <batch> <streams> [steps]   (run on the GPU box)
# raw traces stay in /tmp; the per-kernel / per-grid table and the stats csv go to gpurun_out/prof_b<B>s<S>/
B=${1:-4}; S=${2:-3}; N=${3:-40}
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/prof_raw; rm -rf $RAW; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_b${B}s${S}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o b -- python $GRAFT_REPO_ROOT/bench.py --batch $B --streams $S --steps $N --warmup 5 --repeats 2 --no-cpu-baseline --no-extra-legs > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $RAW/b_kernel_trace.csv > $OUT/by_grid.txt
cp $RAW/b_kernel_stats.csv $OUT/
head -45 $OUT/by_grid.txt
grep -o '"value": [0-9.]*' $OUT/bench.log
