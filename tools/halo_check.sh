cd /root/repo
timeout 120 python tools/ab_iter.py --size 3840x2160 "" "farneback.halo_nt=1" "farneback.halo_nt=4" "farneback.halo_nt=5" "" "farneback.halo_nt=5" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 --batch 4 "" "farneback.halo_nt=5" "farneback.halo_nt=1" 2>&1 | grep pairs
