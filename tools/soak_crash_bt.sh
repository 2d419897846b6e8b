#!/bin/bash
# The round-5 configuration (per-LOGICAL-device lock over one physical GPU, graph replay: concurrent capture / hipGraphLaunch on one device) under rocgdb until it
# crashes: prints the backtrace of the faulting thread.  usage (GPU box): tools/soak_crash_bt.sh [max runs]
cd $GRAFT_REPO_ROOT
ulimit -c 0
for i in $(seq ${1:-12}); do
  OFXCV_LOCK_PER_DEVICE=1 OFXCV_LOCK_BY_LOGICAL=1 timeout 150 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 16" \
     --args python tools/soak_named_frames.py --threads 8 --seconds 15 --devices 4 --opts farneback.graph=1,host.coalesce=0 > /tmp/gdb_$i.log 2>&1
  if grep -q "SIGSEGV\|SIGABRT\|SIGBUS" /tmp/gdb_$i.log; then
    echo "run $i: crashed"; grep -A24 "SIGSEGV\|SIGABRT\|SIGBUS" /tmp/gdb_$i.log | grep -v "^\[New\|^\[Thread\|Switching" | head -40; exit 0
  else echo "run $i: $(grep -a "wrong frames" /tmp/gdb_$i.log | cut -c1-60)"; fi
done
