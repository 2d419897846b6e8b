#!/usr/bin/env python3
"""Where the wall time of a steady-state bench step goes, from a rocprofv3 kernel trace csv (start / end stamps of every dispatch):
how long no kernel runs at all, how long only launches that cannot fill the chip run, the span and the exclusive time of every
phase of the level walk, and the gaps between consecutive dependent launches of the iteration chain.
usage: step_timeline.py <kernel_trace.csv> [--window 0.3,0.8]   (fraction of the trace's time span that is analysed: the timed region)"""
import csv, sys, re, collections, argparse
ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--window", default="0.3,0.8")
ap.add_argument("--pairs", type=int, default=8)
args = ap.parse_args()
rows = []
for r in csv.DictReader(open(args.csv)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
    gx, gy, gz, wx = (int(r.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X"))
    wy, wz = (int(r.get(k, 1) or 1) for k in ("Workgroup_Size_Y", "Workgroup_Size_Z"))
    wgs = (gx // max(1, wx)) * (gy // max(1, wy)) * (gz // max(1, wz))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, (gx, gy, gz, wx), wgs, r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = (float(v) for v in args.window.split(","))
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
rows = [r for r in rows if r[0] >= a and r[1] <= b]
span = (b - a) / 1e3


def phase(n, g):
    gx, gy, gz, wx = g
    if n.startswith("iterate_col"):
        return "L0 col" if gx // wx >= 30 else "L1 col"
    if n.startswith("iterate3h"):
        tiles = gx // wx
        return {31: "L0 strips", 30: "L0 strips", 16: "L1 strips", 15: "L1 strips", 8: "L2 strips", 4: "L3 strips"}.get(tiles, "L? strips %d" % tiles)
    if n.startswith("polyexp"):
        return "polyexp"
    if n.startswith("pyr_"):
        return "pyramid"
    if n.startswith("gray_lut"):
        return "gray"
    return "other"


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv) / 1e3


busy = union([(r[0], r[1]) for r in rows])
print("window %.0f us, %d dispatches; some kernel running %.0f us (%.1f %%), none %.0f us (%.1f %%)" %
      (span, len(rows), length(busy), 100 * length(busy) / span, span - length(busy), 100 * (1 - length(busy) / span)))
full = union([(r[0], r[1]) for r in rows if r[4] >= 256])
print("a launch of >= 256 workgroups running: %.0f us (%.1f %%)" % (length(full), 100 * length(full) / span))
by = collections.defaultdict(list)
for r in rows:
    by[phase(r[2], r[3])].append((r[0], r[1]))
# exclusive time of a phase: its union minus the union of every other phase
def minus(A, B):
    out = []
    j = 0
    for s, e in A:
        cur = s
        for bs, be in B:
            if be <= cur or bs >= e:
                continue
            if bs > cur:
                out.append([cur, bs])
            cur = max(cur, be)
            if cur >= e:
                break
        if cur < e:
            out.append([cur, e])
    return out
steps = max(1, len(by.get("gray", [])))  # one gray-LUT launch per step of the bench
print("%d steps in the window: %.0f us of kernel time (union) per step" % (steps, length(busy) / steps))
print("%-12s %8s %10s %10s %10s   (us per step of %d pairs)" % ("phase", "launches", "sum", "union", "exclusive", args.pairs))
for p, iv in sorted(by.items(), key=lambda kv: -length(union(kv[1]))):
    u = union(iv)
    others = union([x for q, v in by.items() if q != p for x in v])
    print("%-12s %8.1f %10.0f %10.0f %10.0f" % (p, len(iv) / steps, sum(e - s for s, e in iv) / 1e3 / steps, length(u) / steps, length(minus(u, others)) / steps))
# gaps inside the iteration chain: consecutive iterate* dispatches, end -> next start, when nothing of the chain runs in between
chain = [r for r in rows if r[2].startswith("iterate")]
gaps = collections.defaultdict(list)
for p, q in zip(chain, chain[1:]):
    g = (q[0] - p[1]) / 1e3
    if 0 <= g < 100:
        gaps[(phase(p[2], p[3]), phase(q[2], q[3]))].append(g)
print("gaps between consecutive launches of the iteration chain (end -> next start):")
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    tot += sum(v)
    print("  %-10s -> %-10s n %5d  median %5.1f us  p90 %5.1f  sum %8.0f us" % (k[0], k[1], len(v), v[len(v) // 2], v[int(len(v) * 0.9)], sum(v)))
print("  all: %.0f us = %.1f %% of the window" % (tot, 100 * tot / span))
