cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cv2_crosscheck.py tests/test_inpaint_gpu.py tests/test_segment_gpu.py tests/test_ofx_boundary.py -m gpu -q -s -rs 2>&1 | grep -v amdgpu.ids | tail -25
