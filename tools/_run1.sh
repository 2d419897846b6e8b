set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python tools/strict_check.py > gpurun_out/r2a/strict_check.txt 2>&1
tail -40 gpurun_out/r2a/strict_check.txt
for o in "" "farneback.opencv_rounding=1,farneback.strict_rows=2" "farneback.opencv_rounding=1,farneback.strict_rows=4" "farneback.opencv_rounding=1,farneback.strict_rows=8" "farneback.opencv_rounding=2"; do
  echo "== stage opts: $o"; BENCH_CTX_OPTIONS="$o" python tools/bench_stage.py 2>&1 | head -3
done 2>&1 | tee gpurun_out/r2a/stage.txt
for o in "" "farneback.opencv_rounding=1" "farneback.opencv_rounding=1,farneback.strict_rows=4"; do
  for p in 1 3; do
  echo "== bench opts: $o pairs $p"; BENCH_CTX_OPTIONS="$o" python bench.py --steps 100 --warmup 10 --pairs $p --no-cpu-baseline
  done
done 2>&1 | tee gpurun_out/r2a/bench.txt
