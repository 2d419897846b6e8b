cd $GRAFT_REPO_ROOT
for seed in 0 1 2; do python tests/perf/fuzz_sizes.py $seed 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|mismatching"; done
