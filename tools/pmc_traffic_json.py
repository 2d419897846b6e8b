#!/usr/bin/env python3
"""gpurun_out/pmc_bench/summary.txt (tools/pmc_bench.sh) -> profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch of
the level-0 iteration kernels, read = sum of TCC_EA0_RDREQ_{32B,64B,128B} x size, write = WRREQ_64B x 64 + the rest x 32
(MI355X_MICROARCH.md, HBM / rocprofv3 section); per-pair bytes from summary_calls.txt (totals over a known number of pairs).
usage: pmc_traffic_json.py <summary.txt> <out.json> [<summary_calls.txt>]"""
import collections
import json
import re
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(.+?)\s+grid (\d+)\s+(\S+)\s+per-launch\s+([0-9.]+)", line)
    if m:
        rows[(m.group(1).strip(), int(m.group(2)))][m.group(3)] = float(m.group(4))


def traffic(c):
    rd = c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128
    w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = w64 * 64 + (c.get("TCC_EA0_WRREQ_sum", 0) - w64) * 32
    return rd, wr


out = {"method": "rocprofv3 --pmc, separate passes (tools/pmc_bench.sh over a short bench.py run of the timed workload -- batched calls of 8 pairs -- plus its "
                 "single-pair legs); per-launch averages per kernel and grid.  The iterate_col_kernel launches carry all 8 pairs of the call, the iterate3h_kernel "
                 "launches at level 0 one pair",
       "kernels": {}}
want = {"iterate_col_kernel<1, 1,": "opencv_order_col_two_iterations_level0", "iterate3h_kernel<1, 9, 8": "opencv_order_halo_iteration_level0",
        "blur_solve_update_kernel<true>": "direct_window_iteration_level0"}
for (name, grid), c in rows.items():
    for k, tag in want.items():
        if name.startswith(k) and "TCC_EA0_RDREQ_sum" in c:
            best = out["kernels"].get(tag)
            if best and best["grid_threads"] >= grid:
                continue
            rd, wr = traffic(c)
            out["kernels"][tag] = {"kernel": name, "grid_threads": grid, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                                   "traffic_bytes_per_launch": rd + wr, "counters_per_launch": c}

# HBM-side bytes of one whole 1080p frame pair in the timed workload (2 gray LUTs + its share of a batched Farneback call of 8 with F7 inside):
# totals of the counters over tools/run_batch_calls.py divided by the pairs it processed
import os
calls = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(sys.argv[1]), "summary_calls.txt")
tot = collections.defaultdict(dict)
pairs = 0
if os.path.exists(calls):
    for line in open(calls):
        if line.startswith("pairs "):
            pairs = int(line.split()[1])
        m = re.match(r"(.+?)\s+(\S+)\s+total\s+([0-9.]+) \(launches=(\d+)\)", line)
        if m:
            tot[m.group(1).strip()][m.group(2)] = float(m.group(3))
            tot[m.group(1).strip()]["launches"] = int(m.group(4))
if pairs:
    per = {k: sum(traffic(c)) / pairs for k, c in tot.items()}
    out["per_pair_traffic_bytes"] = {"opencv_order": sum(per.values()), "by_kernel": {k: v for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v > 1e5},
                                     "note": "sum over every kernel of the run of its L2 <-> fabric bytes / %d pairs (tools/run_batch_calls.py: batched calls of 8 pairs, "
                                             "2 gray LUTs per pair, F7 inside the call)" % pairs}
json.dump(out, open(sys.argv[2], "w"), indent=1)
if "per_pair_traffic_bytes" in out:
    print("per pair (timed workload): %.0f MB" % (out["per_pair_traffic_bytes"]["opencv_order"] / 1e6))
for k, v in out["kernels"].items():
    print(k, v["kernel"], "%.1f MB" % (v["traffic_bytes_per_launch"] / 1e6))
