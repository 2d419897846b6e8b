cd $GRAFT_REPO_ROOT
python -m pytest tests/test_robustness_gpu.py -m gpu -q -x -k registered 2>&1 | grep -v amdgpu.ids | grep -E "^E|Error|assert" | head -20
OFXCV_TRACE_HOST=2 python tools/bench_host_path.py 2>&1 | grep -v amdgpu.ids | tail -3
