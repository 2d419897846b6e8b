cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh
python bench.py > gpurun_out/refresh/r02_bench.json 2> gpurun_out/refresh/bench.err; head -c 200 gpurun_out/refresh/r02_bench.json
