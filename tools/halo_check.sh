cd /root/repo
timeout 600 python -m pytest tests/test_farneback_gpu.py -x -q 2>&1 | tail -2
timeout 120 python tools/ab_iter.py "" "farneback.halo_min5=100000" 2>&1 | grep pairs
for o in "farneback.halo_min5=100000" ""; do BENCH_CTX_OPTIONS=$o timeout 300 python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o]', round(d['value'],1), round(d['value_one_pair_in_flight'],1))"; done
