#!/bin/bash
# A/B of library builds by bench.py figures (value = batched calls of 8, one = single-pair calls, b16): tools/ab_bench.sh "<variant> ..." [rounds]
cd $GRAFT_REPO_ROOT
V="$1"; N=${2:-3}
for i in $(seq $N); do
  for v in "-" $V; do
    [ "$v" = "-" ] && v=""
    L=$GRAFT_REPO_ROOT/openfx-opencv_amd/lib/libofxcv_hip$v.so
    OFXCV_LIB_PATH=$L timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --repeats 3 --steps 100 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('lib${v:-_default}', 'value',round(d['value'],1),'launch',round(d['roofline']['avg_launch_us'],1),'one',round(d['value_one_pair_in_flight'],1),'b16',round(d['value_batches_of_16'],1))"
  done
done
