cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python tools/ab_iter.py "" "farneback.halo_geom=2" "farneback.halo_strip=66" "farneback.halo_strip=69" "farneback.halo_strip=72" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" "farneback.halo_geom=2" "farneback.halo_strip=66" "farneback.halo_strip=70" 2>&1 | grep pairs
