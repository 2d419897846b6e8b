cd $GRAFT_REPO_ROOT
export BENCH_CTX_OPTIONS=farneback.opencv_rounding=1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $R/bench.py --pairs 1 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/prof.log 2>&1
python $R/tools/trace_by_grid.py $OUT/prof/b_kernel_trace.csv > $OUT/strict_by_grid.txt
rm -f $OUT/prof/b_kernel_trace.csv
cat $OUT/strict_by_grid.txt
