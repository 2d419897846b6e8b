cd /root/repo
timeout 300 python -m pytest tests/test_farneback_gpu.py -x -q -k "faithful_oracle_everywhere" 2>&1 | tail -2
timeout 120 python tools/ab_iter.py "" "farneback.halo_geom=5" "farneback.halo_geom=5,farneback.halo_strip=42" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" "farneback.halo_geom=5" 2>&1 | grep pairs
