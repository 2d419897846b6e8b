#!/bin/bash
# Run on the GPU box: regenerates everything under profiles/ for the current build (round tag $1, default r04).
# Raw traces stay in /tmp; summaries go to gpurun_out/refresh/<tag>_* -- copy those into profiles/ afterwards.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/refresh; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_PMC" ]; then
  # PMC passes first: bench.py reads the traffic json they produce
  bash $R/tools/pmc_bench.sh "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" > $OUT/pmc.log 2>&1
  cp $R/gpurun_out/pmc_bench/summary.txt $OUT/${TAG}_pmc_bench_summary.txt
  cp $R/gpurun_out/pmc_bench/summary_calls.txt $OUT/${TAG}_pmc_bench_calls_summary.txt
  python $R/tools/pmc_traffic_json.py $OUT/${TAG}_pmc_bench_summary.txt $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_bench_calls_summary.txt
  cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
fi
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# kernel trace of the timed workload (defaults: batched calls on several streams) and of one single-pair call at a time
for cfg in "default" "single" "b16"; do
  if [ $cfg = default ]; then ARGS=""; elif [ $cfg = b16 ]; then ARGS="--batch 16"; else ARGS="--batch 1 --streams 1"; fi
  RAW=/tmp/prof_$cfg; rm -rf $RAW
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o b -- python $R/bench.py $ARGS --steps 40 --warmup 10 --repeats 2 --no-cpu-baseline --no-extra-legs --no-batch16 > $OUT/prof_$cfg.log 2>&1
  cp $RAW/b_kernel_stats.csv $OUT/${TAG}_bench_${cfg}_kernel_stats.csv
  [ $cfg = default ] && cp $RAW/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv  # the timed workload under the plain name as well
  python $R/tools/trace_by_grid.py $RAW/b_kernel_trace.csv > $OUT/${TAG}_bench_${cfg}_by_grid.txt
  [ $cfg = default ] && python $R/tools/step_timeline.py $RAW/b_kernel_trace.csv --window 0.15,0.45 > $OUT/${TAG}_step_timeline.txt
done
head -c 700 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_bench_default_by_grid.txt
