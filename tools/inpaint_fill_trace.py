#!/usr/bin/env python3
"""Timeline of the fill launches of the LAST inpaint call in a rocprofv3 kernel trace (run on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ip -o t -- python tools/inpaint_ab.py ""
  python tools/inpaint_fill_trace.py /tmp/ip/t_kernel_trace.csv [launches per call, default 13]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 13
fills = [r for r in rows if "telea_fill" in r["Kernel_Name"]]
last = fills[-per_call:]
t0 = int(last[0]["Start_Timestamp"])
gk = [k for k in rows[0].keys() if "Grid" in k or "Workgroup" in k]
busy = 0.0
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    print("fill launch: start %8.1f us  duration %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, " ".join("%s=%s" % (k, r[k]) for k in gk)))
print("first fill start -> last fill end %.1f us, fill kernels busy %.1f us" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3, busy))
st = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if t0 - 2_500_000 <= int(r["Start_Timestamp"]) <= int(last[-1]["End_Timestamp"]) + 500_000:
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:48]
        st[k][0] += 1
        st[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        st[k].append((int(r["Start_Timestamp"]) - t0) / 1e3)
for k, v in sorted(st.items(), key=lambda kv: kv[1][2]):
    print("%-50s x%-3d %9.1f us   first at %9.1f us" % (k, v[0], v[1], v[2]))
