cd /root/repo
timeout 900 python -m pytest tests/test_farneback_gpu.py -x -q -k "folded_carry_variants or faithful_oracle_everywhere or tiny or first_matrix" 2>&1 | tail -3
timeout 120 python tools/ab_iter.py "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --batch 8 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" 2>&1 | grep pairs
