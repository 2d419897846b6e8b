#!/bin/bash
# Hunts the round-5 soak that did not return: 8 render threads over N logical devices on ONE physical GPU with the per-device lock indexed by the
# LOGICAL device (OFXCV_LOCK_BY_LOGICAL=1: the round-5 behaviour) and graph replay (the round-5 default) -- i.e. concurrent hipGraphLaunch, capture,
# allocation across logical devices.  Every run has 75 s; a run that is still alive then gets its native stacks dumped (rocgdb) and is killed.
# usage (on the GPU box): tools/soak_hang_hunt.sh [runs] [devices] [extra soak args]
cd $GRAFT_REPO_ROOT
RUNS=${1:-6}; DEV=${2:-4}; shift 2
ulimit -c 0
for i in $(seq $RUNS); do
  OFXCV_LOCK_PER_DEVICE=1 OFXCV_LOCK_BY_LOGICAL=1 python tools/soak_named_frames.py --threads 8 --seconds 20 --devices $DEV --opts farneback.graph=1,host.coalesce=0 "$@" > /tmp/soak_$i.log 2>&1 &
  PID=$!
  for s in $(seq 75); do sleep 1; kill -0 $PID 2>/dev/null || break; done
  if kill -0 $PID 2>/dev/null; then
    echo "run $i: STILL ALIVE after 75 s -- native stacks:"
    timeout 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "set pagination off" -ex "thread apply all bt 14" 2>&1 | grep -v "^\[New\|^Reading\|warning:" | head -250
    kill -9 $PID
  else
    wait $PID; echo "run $i: exit $? : $(grep -v amdgpu.ids /tmp/soak_$i.log | tail -2 | cut -c1-200)"
  fi
done
