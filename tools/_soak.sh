cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
for i in 1 2 3 4 5 6 7 8 9 10; do
python -X faulthandler -m pytest tests -m gpu -q -x > gpurun_out/soak/run_$i.log 2>&1
rc=$?
echo "run $i rc $rc: $(grep -E 'passed|failed' gpurun_out/soak/run_$i.log | tail -1)"
if [ $rc -eq 0 ]; then rm gpurun_out/soak/run_$i.log; fi
done
