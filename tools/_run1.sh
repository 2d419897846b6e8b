cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r2d/gpu_tests.txt
