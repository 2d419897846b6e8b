#!/usr/bin/env python3
"""Instruction mix of one kernel of csrc/farneback.hip (gfx950 device assembly): the whole kernel and its longest loop body
(for iterate_col_kernel that is one round of the workgroup = RW rows x 2 steps per wavefront).
usage: python tools/isa_stats.py [--src csrc/farneback.hip] [--asm /tmp/isa/farneback.s] [--rebuild] <substring of the mangled name> [--dump out.s]
The assembly is produced with the library's own flags (openfx-opencv_amd/Makefile: -O3 -ffp-contract=off, gfx950)."""
import argparse, collections, os, re, subprocess, sys

ap = argparse.ArgumentParser()
ap.add_argument("--src", default="csrc/fb_column.hip")
ap.add_argument("--asm", default="/tmp/isa/fb_column.s")
ap.add_argument("--rebuild", action="store_true")
ap.add_argument("--dump")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("name", nargs="?", default="iterate_col_kernelILi1ELi1ELi4ELi8ELi1ELb1ELb0E")
args = ap.parse_args()
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openfx-opencv_amd")
if args.rebuild or not os.path.exists(args.asm):
    os.makedirs(os.path.dirname(args.asm), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I../include",
                           "--cuda-device-only", "-S", args.src, "-o", args.asm], cwd=pkg, stderr=subprocess.DEVNULL)
lines = open(args.asm).read().split("\n")
start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(args.name), l)), None)
if start is None:
    sys.exit("no kernel matching %s" % args.name)
end = start
while "s_endpgm" not in lines[end]:
    end += 1
body = lines[start:end + 1]
if args.dump:
    open(args.dump, "w").write("\n".join(body))


def mix(seg):
    c = collections.Counter()
    for l in seg:
        m = re.match(r"^\s+([a-z][a-z_0-9]+)", l)
        if m:
            c[m.group(1)] += 1
    return c


def report(tag, seg):
    c = mix(seg)
    vec = sum(v for k, v in c.items() if k.startswith("v_"))
    f64 = sum(v for k, v in c.items() if k.startswith("v_") and ("f64" in k))
    vmem = sum(v for k, v in c.items() if k.startswith("buffer_") or k.startswith("global_") or k.startswith("flat_") or k.startswith("scratch_"))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    sal = sum(v for k, v in c.items() if k.startswith("s_"))
    print("%s: %d instructions; vector %d (f64 %d, v_mov_b32 %d, v_cndmask_b32 %d, dpp moves %d, v_pk_* %d), memory %d, LDS %d, scalar %d (s_nop %d, s_waitcnt %d)" %
          (tag, sum(c.values()), vec, f64, c["v_mov_b32_e32"] + c["v_mov_b32_e64"], c["v_cndmask_b32_e32"] + c["v_cndmask_b32_e64"],
           c["v_mov_b32_dpp"], sum(v for k, v in c.items() if k.startswith("v_pk_")), vmem, lds, sal, c["s_nop"], c["s_waitcnt"]))
    for k, v in c.most_common(args.top):
        print("    %-28s %5d" % (k, v))


report("kernel " + lines[start][:60], body)
# longest backward branch = the round loop
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[1] - best[0]:
            best = (labels[m.group(1)], i)
if best:
    report("longest loop (lines %d..%d)" % best, body[best[0]:best[1] + 1])
for l in lines[end:end + 120]:
    m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
    if m and args.name in l:
        print("   ", m.group(1), m.group(2))
