#!/bin/bash
# Run on the GPU box: regenerates everything under profiles/ for the current build (round tag $1, default r02).
# Copy gpurun_out/refresh/<tag>_* into profiles/ afterwards.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/refresh; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC passes first: bench.py reads the traffic json it produces
bash $R/tools/pmc_bench.sh "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" > $OUT/pmc.log 2>&1
cp $R/gpurun_out/pmc_bench/summary.txt $OUT/${TAG}_pmc_bench_summary.txt
python $R/tools/pmc_traffic_json.py $OUT/${TAG}_pmc_bench_summary.txt $OUT/${TAG}_pmc_traffic.json
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
for P in 3 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_p$P -o b -- python $R/bench.py --pairs $P --steps 50 --warmup 10 --repeats 2 --no-cpu-baseline --no-extra-legs > $OUT/prof_p$P.log 2>&1
done
cp $OUT/prof_p3/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp $OUT/prof_p1/b_kernel_stats.csv $OUT/${TAG}_bench_pairs1_kernel_stats.csv
python $R/tools/trace_by_grid.py $OUT/prof_p1/b_kernel_trace.csv > $OUT/${TAG}_bench_pairs1_by_grid.txt
rm -f $OUT/prof_p*/b_kernel_trace.csv
cat $OUT/${TAG}_bench.json | head -c 600
