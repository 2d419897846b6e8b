cd $GRAFT_REPO_ROOT
python -m pytest tests/test_inpaint_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
