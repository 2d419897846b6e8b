#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (+ memory-copy trace) csv of a multi-threaded host run: fraction of the wall time at least one kernel was
running, the idle gaps (histogram), and per-kernel averages of the big kernels.  usage: gpu_busy.py <dir> [skip_fraction]"""
import csv, glob, sys, re, collections
d = sys.argv[1]; skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:40], r.get("Grid_Size_Z", "")))
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
ev = [e for e in ev if e[0] >= t_lo]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, n, z in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = ev[-1][1] - ev[0][0]
print("window %.1f ms: some kernel running %.1f %%; idle gaps: %d, total %.1f ms; gaps > 50 us: %d totalling %.1f ms; > 200 us: %d totalling %.1f ms" % (
    wall / 1e6, 100 * busy / wall, len(gaps), sum(gaps) / 1e6, sum(g > 5e4 for g in gaps), sum(g for g in gaps if g > 5e4) / 1e6,
    sum(g > 2e5 for g in gaps), sum(g for g in gaps if g > 2e5) / 1e6))
agg = collections.defaultdict(lambda: [0.0, 0])
for s, e, n, z in ev:
    agg[(n, z)][0] += (e - s) / 1e3; agg[(n, z)][1] += 1
tot = sum(v[0] for v in agg.values())
print("kernel time summed %.1f ms = %.2f x wall" % (tot / 1e3, tot * 1e3 / wall))
for (n, z), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
    print("  %-40s z=%-3s calls %5d avg %8.1f us total %8.1f ms" % (n, z, c, t / c, t / 1e3))
