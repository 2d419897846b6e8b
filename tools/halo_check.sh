cd /root/repo
timeout 600 python -m pytest tests/test_farneback_gpu.py -x -q -k "folded_carry_variants" 2>&1 | tail -3
timeout 120 python tools/ab_iter.py "" "farneback.halo_min16=1" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --batch 4 "" "farneback.halo_min16=1" "farneback.halo_min16=300" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" "farneback.halo_min16=1" "farneback.halo_min16=600" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 --batch 4 "" "farneback.halo_min16=600" 2>&1 | grep pairs
for o in "farneback.halo_min16=300" ""; do BENCH_CTX_OPTIONS=$o timeout 300 python bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['value_one_pair_in_flight'], d['value_one_batch_in_flight'])"; done
