cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py tests/test_robustness_gpu.py -x -q 2>&1 | tail -3
for o in "" "farneback.l0_token=0"; do for bs in "8 1" "8 2" "4 3" "8 3" "4 2"; do set -- $bs; BENCH_CTX_OPTIONS=$o timeout 300 python bench.py --batch $1 --streams $2 --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o] batch $1 streams $2', round(d['value'],1), round(d['value_one_pair_in_flight'],1))"; done; done
