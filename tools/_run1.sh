cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
