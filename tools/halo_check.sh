cd /root/repo
timeout 300 python -m pytest tests/test_farneback_gpu.py -x -q -k "gray_lut" 2>&1 | tail -2
for o in "lut.four=0" ""; do BENCH_CTX_OPTIONS=$o timeout 300 python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o]', round(d['value'],1), round(d['value_one_pair_in_flight'],1))"; done
cd /tmp && export TMPDIR=/tmp
for o in "lut.four=0" ""; do rm -rf /tmp/tr; BENCH_CTX_OPTIONS=$o timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python /root/repo/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1; python /root/repo/tools/trace_by_grid.py /tmp/tr/t_kernel_trace.csv | grep gray_lut; done
