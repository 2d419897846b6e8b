cd /root/repo
for e in "" "OFXCV_F7_RMW=1" "" "OFXCV_F7_RMW=1"; do env $e timeout 300 python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', round(d['value'],1), round(d['value_one_pair_in_flight'],1))"; done
