cd /root/repo
timeout 600 python -m pytest tests/test_farneback_gpu.py -x -q -k "faithful_oracle_everywhere or tiny or persistent or variants" 2>&1 | tail -2
timeout 120 python tools/ab_iter.py "" 2>&1 | grep pairs
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python /root/repo/tools/trace_call.py "" 2>&1 | grep pairs
python /root/repo/tools/trace_by_grid.py /tmp/tr/t_kernel_trace.csv | head -6
