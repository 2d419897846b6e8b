#!/usr/bin/env python3
"""gpurun_out/pmc_bench/summary.txt (tools/pmc_bench.sh) -> profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch of
the level-0 iteration kernels, read = sum of TCC_EA0_RDREQ_{32B,64B,128B} x size, write = WRREQ_64B x 64 + the rest x 32
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  usage: pmc_traffic_json.py <summary.txt> <out.json>"""
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


out = {"method": "rocprofv3 --pmc, separate passes (tools/pmc_bench.sh over bench.py --batch 1 --streams 1 --steps 2: single-pair calls, so a launch "
                 "is one pair's); per-launch averages per kernel and grid",
       "kernels": {}}
want = {"iterate3h_kernel<1, 9, 8": "opencv_order_halo_iteration_level0", "iterate3f_kernel<true,": "opencv_order_folded_iteration_level0", "iterate3s_kernel<true, 8,": "opencv_order_iteration_level0", "vsum_carry_kernel<8>": "opencv_order_carry_level0", "fold_scan_kernel": "opencv_order_fold_scan_level0",
        "iterate3x2_kernel<true>": "direct_window_fused_pair_level0"}
for (name, grid), c in rows.items():
    for k, tag in want.items():
        if name.startswith(k) and "TCC_EA0_RDREQ_sum" in c:
            best = out["kernels"].get(tag)
            if best and best["grid_threads"] >= grid:
                continue
            rd, wr = traffic(c)
            out["kernels"][tag] = {"kernel": name, "grid_threads": grid, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                                   "traffic_bytes_per_launch": rd + wr, "counters_per_launch": c}

# HBM-side bytes of one whole 1080p frame pair (gray LUT x2 + Farneback with levels 3, iterations 15 + flow -> RGBA) in each
# window mode: per-launch traffic of every kernel and grid (largest grid = pyramid level 0) x its launches per pair
def per_level(prefix):
    grids = sorted(((g, traffic(c)) for (n, g), c in rows.items() if n.startswith(prefix) and "TCC_EA0_RDREQ_sum" in c), reverse=True)
    return [sum(t) for g, t in grids]


def total(spec):
    return sum(mult * sum(per_level(prefix)) for prefix, mult in spec)


prep = [("polyexp_persistent_kernel", 1), ("pyr_", 1), ("gray_lut_kernel", 2)]  # one pyramid / polynomial-expansion launch per level carries both frames of the pair
out["per_pair_traffic_bytes"] = {
    # OpenCV-order mode (default): the overlapped-strip kernel does everything on the main stream -- per level one "first" launch (kind 2 on
    # the coarsest level, 3 below it), 14 iterating launches (kind 1) and the last one (kind 0, which also stores the RGBA pixels)
    "opencv_order": total(prep + [("iterate3h_kernel<1,", 14), ("iterate3h_kernel<0,", 1), ("iterate3h_kernel<2,", 1), ("iterate3h_kernel<3,", 1)]),
    "direct_window": total(prep + [("update_matrices_kernel", 1), ("flow_to_rgba_kernel", 1), ("iterate3x2_kernel", 7), ("iterate3_kernel<false", 1)]),
    "note": "sum over kernels of (bytes per launch from the PMC passes) x (launches per 1920x1080 pair: per level 1 first + 14 iterating + 1 last "
            "launch of iterate3h_kernel in the OpenCV-order mode, 1 first update + 7 fused pairs + 1 final in the direct-window mode + 1 flow -> RGBA; "
            "1 pyramid-image launch + 1 polynomial-expansion launch per level (both frames each), 2 gray LUTs)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("per pair: OpenCV order %.0f MB, direct window %.0f MB" % (out["per_pair_traffic_bytes"]["opencv_order"] / 1e6, out["per_pair_traffic_bytes"]["direct_window"] / 1e6))
for k, v in out["kernels"].items():
    print(k, v["kernel"], "%.1f MB" % (v["traffic_bytes_per_launch"] / 1e6))
