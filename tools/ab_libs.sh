#!/bin/bash
# A/B of library builds on the GPU box: the dominant launch's time (tools/col_time.py) for every lib/libofxcv_hip<variant>.so named, alternating, N rounds
# usage (on the box): tools/ab_libs.sh "<variant> <variant> ..." [rounds] [col_time args]      ("" = the default build)
cd $GRAFT_REPO_ROOT
V="$1"; N=${2:-3}; shift 2
for i in $(seq $N); do
  for v in "-" $V; do
    [ "$v" = "-" ] && v=""
    L=$GRAFT_REPO_ROOT/openfx-opencv_amd/lib/libofxcv_hip$v.so
    echo -n "lib${v:-_default} "; OFXCV_LIB_PATH=$L timeout 300 python tools/col_time.py --reps 1 "$@" 2>&1 | grep "dominant"
  done
done
