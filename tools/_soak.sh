cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak; rm -f gpurun_out/soak/*
for seed in 0 1 2 3; do python tests/perf/fuzz_inpaint.py $seed 2>&1 | grep -v amdgpu.ids | tail -1; done
for i in 1 2 3 4 5 6; do
python -X faulthandler -m pytest tests -m gpu -q -x > gpurun_out/soak/run_$i.log 2>&1
rc=$?
echo "run $i rc $rc: $(grep -E 'passed|failed' gpurun_out/soak/run_$i.log | tail -1)"
if [ $rc -eq 0 ]; then rm gpurun_out/soak/run_$i.log; fi
done
