cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q 2>&1 | tail -5
timeout 120 python tools/ab_iter.py "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --batch 4 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" 2>&1 | grep pairs
timeout 300 python bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['value_one_pair_in_flight'], d['value_one_batch_in_flight'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python /root/repo/tools/trace_call.py "" 2>&1 | grep pairs
python /root/repo/tools/trace_by_grid.py /tmp/tr/t_kernel_trace.csv | head -8
