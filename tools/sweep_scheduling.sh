#!/bin/bash
# pairs/s of the headline workload for {stream priority 0/1/2} x {pairs per batched call x calls in flight} (run on the GPU box)
# OFXCV_STREAM_PRIORITY: 0 all streams alike; 1 preparation stream high; 2 preparation + coarse pyramid levels on high-priority streams
cd $GRAFT_REPO_ROOT
echo "priority batch streams pairs/s one-batch-in-flight"
for prio in 0 1 2; do
for cfg in "1 3" "1 4" "1 6" "2 2" "2 3" "2 4" "3 2" "3 3" "4 2" "4 3" "8 1" "8 2"; do
  set -- $cfg
  echo -n "$prio $1 $2 "
  OFXCV_STREAM_PRIORITY=$prio timeout 300 python bench.py --batch $1 --streams $2 --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-legs 2>&1 | python -c "
import sys,json
t=sys.stdin.read()
try:
    l=json.loads(t.strip().splitlines()[-1]); print(round(l['value'],1), round(l['value_one_batch_in_flight'],1))
except Exception as e:
    print('FAILED', e, t[-800:])
"
done
done
