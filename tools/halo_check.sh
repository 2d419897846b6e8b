cd /root/repo
timeout 120 python tools/ab_iter.py --size 3840x552 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x276 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x1104 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 3840x2160 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 1920x276 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 1920x552 "" 2>&1 | grep pairs
timeout 120 python tools/ab_iter.py --size 1920x1080 "" 2>&1 | grep pairs
