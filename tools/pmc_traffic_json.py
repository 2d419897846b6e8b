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
want = {"iterate3f_kernel<true,": "opencv_order_folded_iteration_level0", "iterate3s_kernel<true, 8,": "opencv_order_iteration_level0", "vsum_carry_kernel<8>": "opencv_order_carry_level0", "fold_scan_kernel": "opencv_order_fold_scan_level0",
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


shared = [("update_matrices_kernel", 1), ("polyexp_persistent_kernel", 2), ("pyr_", 2), ("gray_lut_kernel", 2), ("flow_to_rgba_kernel", 1)]
out["per_pair_traffic_bytes"] = {
    # every iteration / carry kernel of the mode as it ran (folded-carry kernels on the large levels, pre-pass form on the small ones)
    "opencv_order": total(shared + [("iterate3s_kernel<true", 14), ("iterate3s_kernel<false", 1), ("vsum_carry_kernel", 15), ("iterate3f_kernel<true", 14),
                                    ("iterate3f_kernel<false", 1), ("fold_scan_kernel", 15), ("vsum_seed_kernel", 1)]),
    "direct_window": total(shared + [("iterate3x2_kernel", 7), ("iterate3_kernel<false", 1)]),
    "note": "sum over kernels of (bytes per launch from the PMC passes) x (launches per 1920x1080 pair: 14 updating iterations + 1 final per level "
            "and 15 carry pre-passes in the OpenCV-order mode, 7 fused pairs + 1 final in the direct-window mode; 2 pyramid images + 2 polynomial "
            "expansions + 1 first update per level; 2 gray LUTs, 1 flow -> RGBA)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("per pair: OpenCV order %.0f MB, direct window %.0f MB" % (out["per_pair_traffic_bytes"]["opencv_order"] / 1e6, out["per_pair_traffic_bytes"]["direct_window"] / 1e6))
for k, v in out["kernels"].items():
    print(k, v["kernel"], "%.1f MB" % (v["traffic_bytes_per_launch"] / 1e6))
