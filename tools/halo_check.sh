cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q -k "folded_carry_variants or batch_shared or batch_launch_groups or first_matrix" 2>&1 | tail -5
timeout 300 python tools/ab_iter.py "" "farneback.fold_carries=4" "farneback.fold_carries=4,farneback.halo_deep=0" "farneback.fold_carries=4,farneback.halo_deep=3"  "farneback.fold_carries=4,farneback.halo_min8=100" 2>&1 | grep pairs
timeout 300 python tools/ab_iter.py --batch 4 "" "farneback.fold_carries=4" "farneback.fold_carries=4,farneback.halo_deep=0" "farneback.fold_carries=4,farneback.halo_deep=3" "farneback.fold_carries=4,farneback.halo_min8=300" 2>&1 | grep pairs
timeout 300 python tools/ab_iter.py --size 3840x2160 "" "farneback.fold_carries=4" "farneback.fold_carries=4,farneback.halo_deep=3" 2>&1 | grep pairs
timeout 300 python tools/ab_iter.py --size 3840x2160 --batch 4 "" "farneback.fold_carries=4" 2>&1 | grep pairs
cd /tmp && export TMPDIR=/tmp
for o in "farneback.fold_carries=4"; do
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python /root/repo/tools/trace_call.py "$o" 2>&1 | grep pairs
python /root/repo/tools/trace_by_grid.py /tmp/tr/t_kernel_trace.csv | head -24
done
