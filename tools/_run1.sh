cd $GRAFT_REPO_ROOT
python -m pytest tests/test_farneback_gpu.py -m gpu -q -x -s -k "wide_dynamic" 2>&1 | grep -v amdgpu.ids | tail -5
