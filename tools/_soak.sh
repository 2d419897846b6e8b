cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak; rm -f gpurun_out/soak/*
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for i in $(seq 1 60); do
timeout 300 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "thread apply all bt 25" --args python -m pytest tests/test_robustness_gpu.py -m gpu -q -x -k "concurrent_renders" > gpurun_out/soak/gdb_$i.log 2>&1
if grep -q "SIGSEGV\|SIGABRT\|SIGBUS" gpurun_out/soak/gdb_$i.log; then echo "crash in run $i"; break; else rm gpurun_out/soak/gdb_$i.log; fi
done
echo finished $i
