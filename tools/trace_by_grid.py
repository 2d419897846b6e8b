#!/usr/bin/env python3
"""Per (kernel, grid size) average duration from a rocprofv3 kernel trace csv."""
import csv, sys, re, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
    # threads in x, workgroups in y, pairs of the batched call (grid z), threads per workgroup
    g = "%sx%sx%s/%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[(n[:34], g)][0] += d; agg[(n[:34], g)][1] += 1
tot = sum(v[0] for v in agg.values())
for (n, g), (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-34s grid %-20s calls %5d  avg %8.1f us  total %9.1f us  %5.1f%%" % (n, g, c, d / c, d, 100 * d / tot))
