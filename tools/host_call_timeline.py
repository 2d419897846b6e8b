#!/usr/bin/env python3
"""One steady-state host-image call (ofxcv_vectorgen_flows_host[_keyed]) as the GPU saw it: copies and kernels of a window of the rocprofv3
traces (--kernel-trace --memory-copy-trace, csv), start / end relative to the first H2D copy of a call in the middle of the run.
usage: host_call_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [calls]"""
import csv, glob, sys, re
d = sys.argv[1]; ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n[:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nb = next((v for k, v in r.items() if k and ("byte" in k.lower() or "size" in k.lower()) and str(v).isdigit()), "0")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", r.get("Kind", "?")), nb)))
ev.sort()
big = [i for i, e in enumerate(ev) if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2].upper() and int(e[2].split()[-2]) > 1000000]
if not big:
    big = [i for i, e in enumerate(ev) if e[2].startswith("K gray")]
i0 = big[len(big) // 2]
t0 = ev[i0][0]
stop = big[len(big) // 2 + ncalls * 2] if len(big) > len(big) // 2 + ncalls * 2 else len(ev) - 1
last = None
for s, e, n in ev[i0:stop]:
    # runs of the same small kernel are folded
    key = n
    if last and last[0] == key and n.startswith("K iterate3h"):
        last[2] = (e - t0) / 1e3; last[3] += 1
        continue
    if last: print("%9.1f .. %9.1f us  %s%s" % (last[1], last[2], last[0], "  x%d" % last[3] if last[3] > 1 else ""))
    last = [key, (s - t0) / 1e3, (e - t0) / 1e3, 1]
if last: print("%9.1f .. %9.1f us  %s%s" % (last[1], last[2], last[0], "  x%d" % last[3] if last[3] > 1 else ""))
