cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids > gpurun_out/gpu_tests_$i.log
tail -3 gpurun_out/gpu_tests_$i.log | head -1
grep -n -m3 "Fatal\|Segmentation\|Abort\|error" gpurun_out/gpu_tests_$i.log
done
