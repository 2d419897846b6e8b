#!/usr/bin/env python3
"""bench.py -- Farneback dense optical flow throughput at 1920x1080 (BASELINE.json metric).

One step = one pass of the VectorGenerator hot path over a batch of independent synthetic f32 RGBA frame pairs that
are already resident in HBM: sRGB-gray LUT x2 (F0) -> calcOpticalFlowFarneback (F1-F6) -> flow->RGBA write-back (F7),
all through the C ABI of libofxcv_hip.so.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank runs its own independent pairs -- the path has no exchange step, so there is no data-path collective; only the
timing barrier / max-reduce.

`value` is measured in the library's DEFAULT mode: the box window of FarnebackUpdateFlow_Blur evaluated in OpenCV's
own order (every sample within 1e-4 of the reference arithmetic, see `parity`).  `value_direct_window` is the faster
opt-in mode that sums each window directly and leaves the 1e-4 band at a few ill-conditioned pixels.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel of the timed mode: one blur+solve+update iteration at
pyramid level 0, timed live with HIP events on its launch stream), `cpu_baseline` (the CPU oracle -- a port of the
reference's OpenCV algorithm -- timed on this host on a bounded sample) and, at N = 1, one leg per other BASELINE
config (end-to-end host path, Telea inpaint, mean-shift segment, 3840x2160 Farneback).
"""
import argparse
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
LEVELS, ITERS, POLY_N, POLY_SIGMA, WINSIZE, PYR_SCALE = 3, 15, 5, 1.1, 3, 0.5  # VectorGenerator.cpp:804-834, :391-395
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec peak (6.29 TB/s measured float4 copy)
# Issue cost of a wave64 vector instruction: 4 cycles (one quad-cycle).  Evidence: in every PMC pass of the Farneback and mean-shift kernels
# SQ_ACTIVE_INST_VALU (busy quad-cycles) = 1.01-1.02 x SQ_INSTS_VALU, for f32, f64 and integer streams alike
# (profiles/r04_pmc_bench_summary.txt).  The dependent-chain micro-benchmarks (tools/ubench/valurate*.hip -> profiles/r04_ubench_valu_l1.txt)
# give 5.1-5.8 clk for most instructions, 3.0-3.6 for f32 mul / fma: they are latency-bound with 32 independent chains per SIMD, i.e. an upper
# bound.  The guide's 2-cycle v_fma_f32 was not observed in either.
VALU_CLK_PER_WAVE_INSTR = 4.0
VALU_ISSUE_PER_S = 1024 * 2.4e9 / VALU_CLK_PER_WAVE_INSTR
# one iteration = M-in 20 + R0 20 + R1 gather 20 + M-out 20 bytes per pixel
ITER_BYTES_PER_PX = 80.0
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
COL_W = 60  # columns a workgroup of iterate_col_kernel stores (csrc/farneback.hip: kColW)


def algorithmic_bytes_per_pair(w, h, levels=LEVELS, iters=ITERS):
    """per level 1282*n + 2*N0 bytes (coarsest level 1272*n + 2*N0), u8 source."""
    n0 = w * h
    total = 0.0
    for k in range(levels + 1):
        n = round(w * 0.5 ** k) * round(h * 0.5 ** k)
        per = 2 * n0 + 2 * 4 * n + 48 * n + (10 * n if k < levels else 0) + 68 * n + (iters - 1) * 80 * n + 28 * n
        total += per
    return total


def pmc_per_pair():
    """HBM-side bytes of one whole frame pair per window mode (offline PMC, see tools/pmc_traffic_json.py)"""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f).get("per_pair_traffic_bytes", {})
    except Exception:
        return {}


def pmc_kernels():
    """HBM-side bytes / instruction counts per launch of the level-0 iteration kernels from the committed rocprofv3 --pmc
    passes (tools/pmc_bench.sh + tools/pmc_traffic_json.py): counters need their own profiler runs, so these figures are
    measured offline and read here (labelled as such in the line)."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)["kernels"]
    except Exception:
        return {}


def stats(vals):
    return {"median": statistics.median(vals), "min": min(vals), "max": max(vals), "repeats": len(vals)}


def cpu_farneback(ga, gb, budget_s=12.0):
    """Time the CPU oracle (kind 'port': restatement of OpenCV's single-threaded CPU Farneback) on rank 0."""
    from oracle import binding as oracle
    oracle.lib()
    t0 = time.perf_counter()
    n = 0
    while True:
        flow = oracle.calc_optical_flow_farneback(ga, gb, PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0, oracle.BLUR_FAITHFUL)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 8:
            break
    return {"value": n / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d Farneback frame pairs at %dx%d (same synthetic frames, levels=3 iterations=15 poly_n=5), "
                      "oracle/farneback.c (OpenCV evaluation order) single thread, %.1f s" % (n, W, H, el)}, flow


def cpu_farneback_all_cores(ga, gb, per_thread=4, max_threads=32):
    """The same port, one independent frame pair per host thread (ctypes drops the GIL)."""
    import concurrent.futures as cf
    from oracle import binding as oracle
    n_thr = max(1, min(max_threads, os.cpu_count() or 1))

    def work(_):
        for _ in range(per_thread):
            oracle.calc_optical_flow_farneback(ga, gb, PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0, oracle.BLUR_FAITHFUL)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(n_thr) as ex:
        list(ex.map(work, range(n_thr)))
    el = time.perf_counter() - t0
    return {"value": n_thr * per_thread / el, "unit": "frame-pairs/s", "cores": n_thr, "kind": "port",
            "sample": "%d threads x %d Farneback frame pairs at %dx%d, one pair per thread at a time, %.1f s" % (n_thr, per_thread, W, H, el)}


def cv2_probe(ga, gb, ref_flow):
    """If the box has OpenCV's Python module, time the real thing and diff the oracle against it;
    otherwise say so.  Never a requirement."""
    try:
        import cv2
    except Exception as e:
        return {"present": False, "note": "cv2 not importable on this box (%s): the oracle stays pinned by known-answer tests only" % type(e).__name__}
    import numpy as np
    out = {"present": True, "version": cv2.__version__}
    for thr, key in ((1, "pairs_per_s_1_thread"), (0, "pairs_per_s_default_threads")):
        cv2.setNumThreads(thr)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 5.0 and n < 8:
            f = cv2.calcOpticalFlowFarneback(ga, gb, None, PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)
            n += 1
        out[key] = n / (time.perf_counter() - t0)
    err = np.abs(f - ref_flow)
    out["oracle_vs_cv2_max_abs_err"] = float(err.max())
    out["oracle_vs_cv2_outside_1e-4"] = float((err > 1e-4 * np.maximum(1, np.abs(f))).mean())
    return out


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def _spawned_rank(rank, world, port, argv):
    """one rank of a launch bench.py started itself (`--gpus N` without a launcher): the environment torch.distributed.run would give it"""
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    sys.argv = argv
    main()


def dry_run(args, rank, world):
    """BENCH_DRY_RUN=1: the launcher / rendezvous / sharding / reduce path of an N-rank run without a GPU (tests/test_bench_launcher.py):
    every rank takes its share of the pairs, "processes" them in a fixed time per pair, and rank 0 prints the line's launcher fields"""
    import torch.distributed as dist
    from openfx_opencv_amd import sharding
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.pairs_for_rank(world * args.batch, rank, world)
    t0 = time.perf_counter()
    time.sleep(0.01 * len(mine) * (1 + rank))
    el = sharding.reduce_elapsed_max(time.perf_counter() - t0, dist if world > 1 else None)
    total = sharding.reduce_count_sum(len(mine), dist if world > 1 else None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "pairs": total, "value": total / el, "seeds_rank0": [sharding.seed_for_pair(i) for i in mine],
                          "size": args.size, "batch": args.batch}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per batched Farneback call (ofxcv_calc_optical_flow_farneback_batch_rgba); "
                    "8 = BASELINE configs[4]'s pairs per GPU")
    ap.add_argument("--streams", type=int, default=1, help="batched calls in flight per GPU, each on its own context/stream (one: a single pair's "
                    "level 0 then stays in the Infinity Cache; measured against 4 x 3 and others in profiles/r03_exp09_overlapped_strips.txt); "
                    "pairs per step per GPU = batch x streams")
    ap.add_argument("--repeats", type=int, default=10, help="timed regions of --steps steps each; value = their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-f7", action="store_true", help="A/B: flow -> RGBA as its own launch per pair (ofxcv_flow_to_rgba) instead of "
                                                               "inside the Farneback call (ofxcv_calc_optical_flow_farneback_batch_rgba)")
    ap.add_argument("--separate-lut", action="store_true", help="A/B: the gray LUT as one launch per frame (ofxcv_to_byte_grayscale) instead of one per call "
                                                                "(ofxcv_to_byte_grayscale_batch)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the inpaint / segment / 4K / host-path legs")
    ap.add_argument("--direct-leg", action="store_true", help="also time the opt-in direct-window mode (farneback.opencv_rounding 0: outside the reference's 1e-4 band at a "
                    "few samples; no longer faster for batches) -- off by default since round 5, the time goes to repeats of `value`")
    ap.add_argument("--no-batch16", action="store_true", help="skip the batches-of-16 leg (counter passes: its launches have the grids of other levels' launches of 8)")
    ap.add_argument("--size", default="1920x1080", help="frame size; the metric is quoted at 1920x1080 (BASELINE.json configs[2]), "
                    "3840x2160 is configs[4] (64 pairs over 8 GPUs)")
    ap.add_argument("--config", type=int, default=0, help="5 = BASELINE configs[4]: --size 3840x2160 --batch 8 (64 pairs over 8 GPUs, seeds 1234...1297)")
    args = ap.parse_args()
    if args.config == 5:
        args.size, args.batch, args.streams = "3840x2160", 8, 1
    # `--gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here, one process per GPU, the way torch.distributed.run would
    # (same environment variables, same rendezvous); with a launcher its WORLD_SIZE must agree with --gpus -- the line never reports
    # an n_gpus other than what ran
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        import torch.multiprocessing as mp
        mp.spawn(_spawned_rank, args=(args.gpus, _free_port(), list(sys.argv)), nprocs=args.gpus, join=True)
        return
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world))
    global W, H
    W, H = (int(v) for v in args.size.lower().split("x"))
    if os.environ.get("BENCH_DRY_RUN"):
        return dry_run(args, int(os.environ.get("RANK", "0")), int(env_world or 1))

    import numpy as np
    import torch
    import openfx_opencv_amd as ofxcv
    from openfx_opencv_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    # BENCH_BACKEND=gloo + BENCH_SHARE_DEVICE=1 let the N > 1 code path be exercised on a single-GPU box (all ranks on
    # device 0, CPU tensors for the two reduces); the driver's multi-GPU runs use the defaults (one GPU per rank, RCCL).
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_DEVICE"):
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (%d visible); BENCH_SHARE_DEVICE=1 + BENCH_BACKEND=gloo put every rank on device 0 "
                         "to exercise the N > 1 path on a one-GPU box" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)  # RCCL comes up lazily: fail here, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:  # the collectives only carry the timing barrier and two scalars: gloo serves them as well
                sys.stderr.write("bench.py: RCCL unavailable (%s: %s), timing barrier over gloo\n" % (type(e).__name__, e))
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend, red_dev = "gloo", "cpu"
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    B, S = max(1, args.batch), max(1, args.streams)
    P = B * S
    extra_opts = [kv.split("=") for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(","))]  # A/B of kernel variants

    def make_ctxs(n, direct):
        cs = [ofxcv.Context(local_rank) for _ in range(n)]
        for c in cs:
            c.set_option("farneback.opencv_rounding", 0 if direct else 1)
            for k, v in extra_opts:
                c.set_option(k, int(v))
        return cs

    def make_bufs(cs, w, h, nb=None):
        """per context: `nb` independent frame pairs (different seeds), resident in HBM"""
        nb = nb or B
        mine = sharding.pairs_for_rank(world * len(cs) * nb, rank, world)
        bufs = []
        for i, c in enumerate(cs):
            with torch.cuda.stream(c.stream):
                t = dict(a=[], b=[], ga=[], gb=[], flow=[], out=[])
                for j in range(nb):
                    a, b = synth.flow_pair(w, h, seed=sharding.seed_for_pair(mine[i * nb + j]))
                    t["a"].append(torch.from_numpy(a).cuda())
                    t["b"].append(torch.from_numpy(b).cuda())
                    t["ga"].append(torch.empty((h, w), dtype=torch.uint8, device="cuda"))
                    t["gb"].append(torch.empty((h, w), dtype=torch.uint8, device="cuda"))
                    t["flow"].append(torch.empty((h, w, 2), dtype=torch.float32, device="cuda"))
                    t["out"].append(torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"))
                bufs.append(t)
        return bufs

    def step(cs, bufs):
        # independent frame pairs: every context (own HIP stream and scratch) takes a batch of them through ONE batched
        # Farneback call; frame pairs never exchange data, so they shard across batches and streams exactly as across GPUs
        for c, t in zip(cs, bufs):
            with torch.cuda.stream(c.stream):
                if args.separate_lut:
                    for a, b, ga, gb in zip(t["a"], t["b"], t["ga"], t["gb"]):
                        c.to_byte_grayscale(a, ga)
                        c.to_byte_grayscale(b, gb)
                else:  # F0 for the 2n frames of the call in one launch
                    c.to_byte_grayscale_batch(t["a"] + t["b"], t["ga"] + t["gb"])
                if args.separate_f7:
                    c.calc_optical_flow_farneback_batch(t["ga"], t["gb"], t["flow"], PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)
                    for fl, o in zip(t["flow"], t["out"]):
                        c.flow_to_rgba(fl, o, 0b0001, 0b0010)  # forward.u -> R, forward.v -> G (defaults :739,753)
                else:
                    # the same work through the entry point that carries F7: the flow field AND the RGBA image are written
                    nb = len(t["ga"])
                    c.calc_optical_flow_farneback_batch_rgba(t["ga"], t["gb"], t["flow"], t["out"], [0b0001] * nb, [0b0010] * nb, 1.0, 1.0,
                                                             PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)

    def timed_regions(cs, bufs, steps, warmup, repeats):
        """`repeats` regions of exactly `steps` steps, each bracketed by barrier + synchronize; elapsed = max over ranks"""
        for _ in range(warmup):
            step(cs, bufs)
        out = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(cs, bufs)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            out.append(sharding.reduce_elapsed_max(time.perf_counter() - t0, dist, red_dev))
        return out

    def kernel_leg(c, t, which):
        """HIP event pairs around every level-0 launch of one kernel, on the stream it is launched on, one batched call in
        flight (a launch carries the whole batch: its algorithmic bytes are those of len(t["ga"]) pairs)"""
        c.profile_enable(which)
        with torch.cuda.stream(c.stream):
            for _ in range(max(3, min(10, args.steps))):
                c.calc_optical_flow_farneback_batch(t["ga"], t["gb"], t["flow"], PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)
        torch.cuda.synchronize()
        ms, n = c.profile_read()
        c.profile_enable(0)
        return (ms / 1e3 / max(1, n)), n

    # ---- headline: default mode (OpenCV-order window) ----
    ctxs = make_ctxs(S, direct=False)
    bufs = make_bufs(ctxs, W, H)
    el = timed_regions(ctxs, bufs, args.steps, args.warmup, max(1, args.repeats))
    pairs_per_region = sharding.reduce_count_sum(args.steps * P, dist, red_dev)
    rates = [pairs_per_region / e for e in el]
    # pairs one level-0 launch of the dominant kernel carries.  Column-owning form (iterate_col_kernel, taken by the pairs of the call that
    # fill whole rounds of the chip: ofxcv_farneback_col_pairs): those pairs, TWO iterations per launch.  Otherwise the overlapped-strip
    # form: a level is walked in groups of pairs whose working set (80 B/px each) stays inside the Infinity Cache budget
    # (option farneback.batch_mb), one iteration per launch -- see enqueue_farneback
    pitch = ofxcv.farneback_plane_pitch(W) if hasattr(ofxcv, "farneback_plane_pitch") else (W + 63) // 64 * 64
    ncol = ctxs[0].farneback_col_pairs(W, H, B)  # the library's own plan for level 0 (the pairs beyond it keep the overlapped strips)
    col = ncol > 0
    ppl = ncol if col else max(1, min(B, (ctxs[0].get_option("farneback.batch_mb") << 20) // (80 * pitch * H)))
    iters_per_launch = 2 if col else 1
    # the dominant kernel is timed on calls of as many pairs as one of its launches carries in the timed workload
    kl = {k: v[:ppl] for k, v in bufs[0].items()}
    main_s, main_n = kernel_leg(ctxs[0], kl, 1)
    # what the timed workload left for pair 0 (checked against the CPU oracle below) -- taken before any other leg touches the buffers
    step(ctxs, bufs)
    torch.cuda.synchronize()
    strict_flow = bufs[0]["flow"][0].cpu().numpy()
    g_a, g_b = bufs[0]["ga"][0].cpu().numpy(), bufs[0]["gb"][0].cpu().numpy()
    col_aborts = ctxs[0].get_option("farneback.col_aborts")
    one_in_flight = one_batch_in_flight = three_single = batch16 = lock_hold_us = lock_call_us = None
    if world == 1:
        n1 = max(10, args.steps // 2)
        e1 = timed_regions(ctxs[:1], bufs[:1], n1, 3, 3)
        one_batch_in_flight = n1 * B / statistics.median(e1)
        lock_call_us = statistics.median(e1) / n1 * 1e6
        # time inside the process-wide runtime lock (the hipGraphLaunch of the call's captured launch sequence) per batched call, measured on calls
        # that are waited for one by one: in a saturated loop hipGraphLaunch blocks on the stream's queue and the hold is the call's own GPU time.
        # With G devices driven from ONE host process the lock is busy hold x G / call time of the time
        lh0 = ctxs[0].lock_hold()
        for _ in range(10):
            step(ctxs[:1], bufs[:1])
            torch.cuda.synchronize()
        lh1 = ctxs[0].lock_hold()
        lock_hold_us = (lh1[0] - lh0[0]) / 1e3 / max(1, lh1[1] - lh0[1])
        one = [{k: v[:1] for k, v in bufs[0].items()}]  # a single pair per call on one stream: what one unbatched caller gets
        e1 = timed_regions(ctxs[:1], one, n1, 3, 3)
        one_in_flight = n1 / statistics.median(e1)
    for c in ctxs:
        c.close()
    del bufs
    if world == 1:
        # the configuration rounds 1 and 2 quoted as `value`: three single-pair calls in flight (three contexts, one pair each)
        c3 = make_ctxs(3, direct=False)
        b3 = make_bufs(c3, W, H, 1)
        e3 = timed_regions(c3, b3, max(10, args.steps // 2), 5, 3)
        three_single = max(10, args.steps // 2) * 3 / statistics.median(e3)
        for c in c3:
            c.close()
        del b3
    if world == 1 and not args.no_batch16:
        # twice the pairs per call: the 960x540 level then has 256 workgroups as well and takes the column-owning form
        c16 = make_ctxs(1, direct=False)
        b16 = make_bufs(c16, W, H, 16)
        e16 = timed_regions(c16, b16, max(5, args.steps // 4), 3, 3)
        batch16 = max(5, args.steps // 4) * 16 / statistics.median(e16)
        for c in c16:
            c.close()
        del b16

    # ---- the opt-in direct-window mode, same workload (--direct-leg) ----
    # (its kernels take one pair per launch: P single-pair contexts in flight, as in rounds 1 and 2)
    drates = fused_s = fused_n = direct_flow = None
    if args.direct_leg:
        dctxs = make_ctxs(min(P, 4), direct=True)
        dbufs = make_bufs(dctxs, W, H, 1)
        del_ = timed_regions(dctxs, dbufs, args.steps, args.warmup, max(1, min(5, args.repeats)))
        drates = [sharding.reduce_count_sum(args.steps * len(dctxs), dist, red_dev) / e for e in del_]
        fused_s, fused_n = kernel_leg(dctxs[0], dbufs[0], 1)
        direct_flow = dbufs[0]["flow"][0].cpu().numpy()
        for c in dctxs:
            c.close()
        del dbufs

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = statistics.median(rates)
    alg = algorithmic_bytes_per_pair(W, H)
    pmc = pmc_kernels() if (W, H) == (1920, 1080) else {}
    pm = pmc.get("opencv_order_col_two_iterations_level0" if col else "opencv_order_halo_iteration_level0", {})
    pf = pmc.get("direct_window_fused_pair_level0", {})
    iter_bytes_pair = ITER_BYTES_PER_PX * W * H
    # SURVEY's count: 80 B/px per ITERATION.  The column-owning launch runs two iterations and keeps the field between them on chip:
    survey_bytes = iter_bytes_pair * ppl * iters_per_launch
    # what one launch of the dominant kernel HAS to move -- the figure `achieved` / `frac` are computed from (VERDICT / ADVICE round 4):
    # field-in 20 + R0 20 + R1 20 + field-out 20 = 80 B/px ONCE per launch, the three input fields re-read by the four halo lanes of every 64-lane
    # tile column (x 64/60): 20 + 60 x 64/60 = 84 B/px (the judge's 85).  Overlapped strips (one iteration per launch): SURVEY's 80 B/px.
    min_bytes_px = (20.0 + 60.0 * 64.0 / COL_W) if col else ITER_BYTES_PER_PX
    iter_bytes = min_bytes_px * W * H * ppl
    achieved = iter_bytes / main_s / 1e9
    traffic = pm.get("traffic_bytes_per_launch")
    valu = pm.get("counters_per_launch", {}).get("SQ_INSTS_VALU")
    valu_busy = pm.get("counters_per_launch", {}).get("SQ_ACTIVE_INST_VALU")
    line = {
        "metric": "frames/sec at %dx%d f32 (Farneback flow)" % (W, H),
        "value": value,
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": statistics.median(el) / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "VectorGenerator Farneback dense optical flow, %dx%d f32 RGBA frame pair resident in HBM "
                               "-> 8-bit sRGB gray -> calcOpticalFlowFarneback -> flow RGBA (BASELINE.json configs[%d])"
                               % (W, H, 4 if (W, H) == (3840, 2160) else 2),
                   "levels": LEVELS, "iterations": ITERS, "poly_n": POLY_N, "poly_sigma": POLY_SIGMA, "winsize": WINSIZE,
                   "pyr_scale": PYR_SCALE, "pairs_per_step_per_gpu": P, "pairs_per_batched_call": B, "streams_per_gpu": S,
                   "box_window": "OpenCV order (library default): running f64 column sums of f32-rounded row differences, strip-parallel",
                   "call_form": "batched: one ofxcv_calc_optical_flow_farneback_batch_rgba call of %d different pairs at a time per stream (BASELINE configs[4]'s 8 pairs per GPU); "
                                "rounds 1 and 2 quoted three single-pair calls in flight: value_three_single_pair_calls_in_flight" % B,
                   "gray_lut": ("one launch per frame (ofxcv_to_byte_grayscale; --separate-lut)" if args.separate_lut else
                                "the 2n frames of a call in one launch (ofxcv_to_byte_grayscale_batch; one launch per frame, as in rounds 1-3 and the first "
                                "half of round 4: --separate-lut, 1.2 % less)"),
                   "parallelism": "independent frame pairs per GPU, no collective"},
        "value_stats": dict(stats(rates), note="each repeat = one timed region of `steps` steps bracketed by barrier + synchronize; value = median"),
        "value_opencv_order": value,  # the timed mode IS the OpenCV-order mode (library default); kept as an explicit key
        "value_one_pair_in_flight": one_in_flight,      # one unbatched call at a time (a single OFX render thread, one direction)
        "value_one_batch_in_flight": one_batch_in_flight,  # one batched call of `pairs_per_batched_call` pairs at a time
        "value_three_single_pair_calls_in_flight": three_single,  # the configuration BENCH_r01 / BENCH_r02 quoted as `value`
        "value_batches_of_16": batch16,  # one batched call of 16 pairs at a time (the call's maximum)
        "host_lock_hold_us_per_call": lock_hold_us,  # inside the process-wide runtime lock per batched call: 0 with eager launches (the default since round 6)
        "host_lock": None if lock_hold_us is None else {
            "hold_us_per_batched_call": lock_hold_us, "call_us": lock_call_us, "utilisation_at_8_gpus_in_one_process": 8 * lock_hold_us / lock_call_us,
            "note": "ofxcv_lock_hold: time a batched call of %d pairs spends inside the runtime lock / the call's GPU time; x 8 = how busy the ONE lock of a "
                    "host process would be with eight devices.  Since round 6 a call's launches are enqueued eagerly (farneback.graph 0): no lock is "
                    "held (0 holds); with farneback.graph 1 it is one hipGraphLaunch of 200-350 us" % B},
        "col_aborts": col_aborts,  # 1 if a bounded LDS wait of iterate_col_kernel ever ran out (never seen)
        "value_direct_window": statistics.median(drates) if drates else None,
        "value_direct_window_stats": None if not drates else dict(stats(drates), note="opt-in mode farneback.opencv_rounding=0 (C ABI only): each 3x3 window summed directly by the generic window kernel, one iteration "
                                          "per launch; does NOT meet 1e-4 at every sample (see parity)"),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_source": "offline PMC (rocprofv3 --pmc, separate passes), %s" % PMC_FILE if traffic else None,
                     "kernel": ("iterate_col_kernel<kHaloIter, kHaloIter, 4, 8> (TWO blur+solve+update iterations in OpenCV's summation order per launch: a workgroup of eight "
                                "wavefronts owns a 60-pixel tile column of one pair over the full height and walks it in rounds of 32 rows; the running f64 column sums of both "
                                "iterations and three boundary rows per wavefront are handed on through LDS point to point, the intermediate matrices never leave the registers; "
                                "pyramid level 0, %dx%d, every pair of the call in the grid's z)" if col else
                                "iterate3h_kernel<kHaloIter, 9, 8, var> (one blur+solve+update iteration in OpenCV's summation order as ONE launch: overlapped strips of 65..72 computed rows, "
                                "eight wavefronts of 8 or 9 rows per workgroup, the strip sums of its own output for the next launch's column-sum prefix; pyramid level 0, %dx%d)") % (W, H),
                     "bytes_per_launch": iter_bytes, "pairs_per_launch": ppl, "iterations_per_launch": iters_per_launch,
                     "bytes_per_launch_note": "the bytes ONE launch has to move: %.1f B/px (field-in 20 + R0 20 + R1 20, re-read by the 4 halo lanes of each 64-lane tile column, "
                                              "+ field-out 20; the field between the launch's %d iterations stays on chip) x %d x %d px x %d pairs" % (min_bytes_px, iters_per_launch, W, H, ppl),
                     "frac_min_bytes": achieved / HBM_PEAK_GBS,  # = frac (kept under the name VERDICT round 4 asked for)
                     "traffic_over_min": (traffic / iter_bytes) if traffic else None,
                     "survey_per_iteration_count": {"bytes_per_launch": survey_bytes, "achieved": survey_bytes / main_s / 1e9, "frac": survey_bytes / main_s / 1e9 / HBM_PEAK_GBS,
                                                    "note": "SURVEY.md 8(d)'s 80 B/px per ITERATION x the iterations of the launch: the bytes an unfused form would move "
                                                            "(`frac` of rounds 1-4; 0.81 in round 4).  Not what the fused launch moves: side key only"},
                     "avg_launch_us": main_s * 1e6, "launches_timed": main_n,
                     "timing": "HIP event pairs on the launch stream, one batched call in flight; the pairs include the dependent-launch gap -- the rocprofv3 "
                               "durations of the same launches are in profiles/r06_bench_default_by_grid.txt",
                     "traffic_GBps": (traffic / main_s / 1e9) if traffic else None,
                     "traffic_frac_of_peak": (traffic / main_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "traffic_note": "L2 <-> fabric bytes per launch (TCC_EA0 read requests x their size + write requests); Infinity-Cache hits are counted",
                     "bound_actual": ("the launch moves its minimal bytes (every field once per two iterations) at about 0.75 of the copy rate; what holds it is the workgroup, not memory: "
                                      "eight wavefronts hand two f64 column sums and six boundary rows to each other per round of 32 rows, two wavefronts per SIMD (159 KB of LDS: R1 ring + "
                                      "hand-off rows); since round 6 s_setprio orders them by phase (step-2 rows first), which spreads the wavefronts evenly over a round and took the "
                                      "launch from 329 to ~300 us (profiles/r06_experiments.md 5)" if col else
                                      "hbm-side: the kernel moves 1.12x its algorithmic bytes at 0.83 of the achievable copy rate"),
                     "valu_issue_frac": (valu / VALU_ISSUE_PER_S / main_s) if valu else None,
                     "valu_busy_frac": (valu_busy * 4 / (1024 * 2.4e9) / main_s) if valu_busy else None,
                     "valu_issue_note": "SQ_INSTS_VALU per launch (offline PMC) x %.1f clk per wave64 instruction (see VALU_CLK_PER_WAVE_INSTR in bench.py) "
                                        "/ (1024 SIMDs x 2.4 GHz) / launch time; valu_busy_frac = SQ_ACTIVE_INST_VALU (busy quad-cycles) x 4 / the same" % VALU_CLK_PER_WAVE_INSTR,
                     "direct_window_kernel": None if not fused_s else {
                         "kernel": "blur_solve_update_kernel<true> (the generic box window: one iteration of one pair per launch, direct sums; the fused two-iteration kernel of rounds 1-5 was removed)",
                         "avg_launch_us": fused_s * 1e6, "launches_timed": fused_n, "bytes_per_launch": iter_bytes_pair,
                         "achieved": iter_bytes_pair / fused_s / 1e9, "frac": iter_bytes_pair / fused_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": pf.get("traffic_bytes_per_launch"),
                         "traffic_GBps": (pf["traffic_bytes_per_launch"] / fused_s / 1e9) if pf.get("traffic_bytes_per_launch") else None}},
        "whole_call": {"algorithmic_bytes_per_pair": alg, "achieved_GBps": alg * value / world / 1e9,
                       "frac_of_hbm_peak": alg * value / world / 1e9 / HBM_PEAK_GBS,
                       "direct_window_frac_of_hbm_peak": (alg * statistics.median(drates) / world / 1e9 / HBM_PEAK_GBS) if drates else None},
    }
    pp = pmc_per_pair() if (W, H) == (1920, 1080) and B == 8 else {}
    if pp.get("opencv_order"):
        # measured L2 <-> fabric traffic of a whole pair in the timed workload x the measured rate: how close the whole job is to the memory roofline
        t_s = pp["opencv_order"]
        line["whole_call"].update({
            "traffic_bytes_per_pair": t_s, "traffic_source": "offline PMC, %s (counter totals over batched calls of 8 / the pairs they processed)" % PMC_FILE,
            "traffic_GBps": t_s * value / world / 1e9, "traffic_frac_of_hbm_peak": t_s * value / world / 1e9 / HBM_PEAK_GBS,
            "traffic_frac_of_achievable_6300GBps": t_s * value / world / 1e9 / 6300.0})
    if world == 1 and not args.no_cpu_baseline:
        cb, ref_flow = cpu_farneback(g_a, g_b)
        line["cpu_baseline"] = cb
        line["cpu_baseline_all_cores"] = cpu_farneback_all_cores(g_a, g_b)
        line["cv2"] = cv2_probe(g_a, g_b, ref_flow)

        def par(got):
            err = np.abs(got - ref_flow)
            bad = err > 1e-4 * np.maximum(1, np.abs(ref_flow))
            return {"outside_1e-4": float(bad.mean()), "max_abs_err": float(err.max()), "bit_identical": float((got == ref_flow).mean())}
        line["parity"] = {"reference": "CPU oracle, OpenCV evaluation order (oracle/farneback.c, ORC_BLUR_FAITHFUL), same %dx%d pair; oracle itself is "
                                       "pinned by known-answer tests only (no OpenCV in this image)" % (W, H),
                          "tolerance": "|a-b| <= 1e-4 * max(1, |b|)",
                          "timed_mode_opencv_order": par(strict_flow), "direct_window_mode": par(direct_flow) if direct_flow is not None else None}
    elif world > 1:
        line["cpu_baseline"] = None
    if world == 1 and not args.no_extra_legs and (W, H) == (1920, 1080):
        try:
            line.update(extra_legs(ofxcv, synth, torch, np, local_rank, not args.no_cpu_baseline, args.direct_leg))
        except Exception as e:  # the headline must still be reported
            line["extra_legs_error"] = "%s: %s" % (type(e).__name__, e)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def extra_legs(ofxcv, synth, torch, np, dev, with_cpu, direct_leg=False):
    """One leg per remaining BASELINE config, rank 0 at N = 1 only; every figure is a median of a few runs."""
    out = {}
    med = statistics.median

    # ---- end to end: f32 RGBA host frames -> flow written into a host RGBA image (PCIe inclusive) ----
    # consecutive frames of one shot (round 6; before, the backward pair was two unrelated textures: every gather of its level walk left the
    # +-4 px window the column-owning kernel keeps in LDS -- not what playback looks like)
    shot = synth.sequence(1920, 1080, 4)
    prev, a, b = shot[0], shot[1], shot[2]

    def host_rate(nthreads, seconds=1.5, named=False):
        """every calling thread renders output frames (own context, own host buffers) for `seconds`; >= 1 s per leg.
        named: the threads render the output frames of ONE endless sequence in order (a free thread takes the next frame, as a host's render queue hands them out) and
        pass a name with every frame (ofxcv_vectorgen_flows_host_keyed; an OFX host's kOfxImagePropUniqueIdentifier): per output frame
        one frame the device has not seen, two it has."""
        cs = [ofxcv.Context(dev) for _ in range(nthreads)]
        if named:
            cs[0].host_cache_clear()
        outs = [np.zeros((1080, 1920, 4), np.float32) for _ in range(nthreads)]
        srcs = [(a.copy(), b.copy(), prev.copy()) for _ in range(nthreads)]  # a host hands every render thread its own frames
        for c, o, (x, y, z) in zip(cs, outs, srcs):
            for _ in range(2):
                c.vectorgen_flows_host(x, y, z, o, 1, 2, 4, 8)
        counts = [0] * nthreads
        stop = threading.Event()

        ring = shot  # the buffers come round again under new names (the shot runs forth and back: neighbouring times are neighbouring frames)
        nr = len(ring)

        next_frame = [0]
        frame_lock = threading.Lock()

        def work(i):
            c, o, (x, y, z) = cs[i], outs[i], srcs[i]
            while not stop.is_set():
                if named:
                    with frame_lock:    # a host hands out the output frames of a sequence in order, whichever render thread is free
                        t = next_frame[0]
                        next_frame[0] += 1
                    c.vectorgen_flows_host(ring[synth.pingpong(t, nr)], ring[synth.pingpong(t + 1, nr)], ring[synth.pingpong(t - 1, nr)], o, 1, 2, 4, 8,
                                           keys=("f%d" % t, "f%d" % (t + 1), "f%d" % (t - 1)))
                else:
                    c.vectorgen_flows_host(x, y, z, o, 1, 2, 4, 8)
                counts[i] += 1
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        [t.start() for t in th]
        time.sleep(seconds)
        stop.set()
        [t.join() for t in th]
        el = time.perf_counter() - t0
        co = [c.host_coalesce_stats() for c in cs]
        if sum(x[0] for x in co):
            coalesced["%s_%d" % ("named" if named else "unnamed", nthreads)] = round(sum(x[2] for x in co) / sum(x[0] for x in co), 2)
        for c in cs:
            c.close()
        return 2 * sum(counts) / el
    coalesced = {}
    end_to_end = {"workload": "ofxcv_vectorgen_flows_host: one default VectorGenerator output frame (forward + backward flow = 2 frame pairs, one "
                                     "batched Farneback call), three 1920x1080 f32 RGBA host frames in, one host RGBA frame out, PCIe inclusive; "
                                     "every calling thread renders for 1.5 s",
                         "unit": "frame-pairs/s", "calling_threads_1": host_rate(1), "calling_threads_2": host_rate(2), "calling_threads_4": host_rate(4),
                         "calling_threads_8": host_rate(8),
                         "playback_named_frames": {
                             "workload": "the same output frames as consecutive frames of a sequence whose frames the caller names "
                                         "(ofxcv_vectorgen_flows_host_keyed; an OFX host's kOfxImagePropUniqueIdentifier): the 8-bit gray image of a "
                                         "named frame stays on the device, so each output frame uploads ONE f32 frame instead of three",
                             "calling_threads_1": host_rate(1, named=True), "calling_threads_2": host_rate(2, named=True),
                             "calling_threads_4": host_rate(4, named=True), "calling_threads_8": host_rate(8, named=True)}}
    # concurrent calls of one device are coalesced (option host.coalesce): mean pairs of the batched call a render thread's two pairs rode in
    end_to_end["coalesced_mean_pairs_per_call"] = coalesced

    # ---- Telea inpaint (configs[0] size and configs[1]) ----
    ctx = ofxcv.Context(dev)
    for (w, h, key) in ((640, 480, "inpaint_640x480"), (1920, 1080, "inpaint_1080p")):
        fr = synth.inpaint_frame(w, h)
        ctx.inpaint_render_host(fr)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.inpaint_render_host(fr)
            ts.append(time.perf_counter() - t0)
        d = torch.from_numpy(fr).cuda()
        m = ctx.inpaint_mask(d, 1)
        torch.cuda.synchronize()
        hole = int((m > 0).sum())
        td = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.inpaint_telea(d, m)
            torch.cuda.synchronize()
            td.append(time.perf_counter() - t0)
        leg = {"workload": "opencv2fx/inpaint render body, %dx%d 8-bit RGBA, radius 3, dilation 1, %d hole pixels" % (w, h, hole),
               "render_host_ms": med(ts) * 1e3, "telea_device_images_ms": med(td) * 1e3, "Mpx_hole_per_s": hole / med(td) / 1e6,
               "bound": "dependency latency of the fast-marching order (host march + dataflow fill), not HBM"}
        if with_cpu:
            from oracle import binding as oracle
            t0 = time.perf_counter()
            oracle.inpaint_render(fr)
            leg["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        out[key] = leg

    # ---- mean-shift segment, configs[3] ----
    w, h = 3840, 2160
    fr = np.ascontiguousarray(synth.inpaint_frame(w, h, n_holes=0)[..., :3])
    d = torch.from_numpy(fr).cuda()
    ctx.pyr_mean_shift_filtering(d)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.pyr_mean_shift_filtering(d)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    leg = {"workload": "pyramid mean-shift filtering, 3840x2160 8-bit RGB resident in HBM, sp 10, sr 20, maxLevel 2", "ms": med(ts) * 1e3,
           "Mpx_per_s": w * h / med(ts) / 1e6, "bound": "integer VALU issue (window taps), compulsory HBM traffic 6 B/px"}
    try:  # instruction count from the committed PMC pass (tools/pmc_segment.sh) / issue capacity over the measured time
        with open(os.path.join(ROOT, "profiles", "r02_pmc_segment_valu.json")) as f:
            nvalu = json.load(f)["valu_wave_instructions_per_filter_call"]
        leg["valu_issue_frac"] = nvalu / VALU_ISSUE_PER_S / med(ts)
        leg["valu_issue_note"] = "SQ_INSTS_VALU per filter call (offline PMC, profiles/r02_pmc_segment_valu.json) x 4 clk (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycles) / (1024 SIMDs x 2.4 GHz) / measured time"
    except Exception:
        pass
    if with_cpu:
        from oracle import binding as oracle
        crop = np.ascontiguousarray(fr[:540, :960])
        t0 = time.perf_counter()
        oracle.pyr_mean_shift(crop)
        c = time.perf_counter() - t0
        leg["cpu_oracle_Mpx_per_s"] = 960 * 540 / c / 1e6
        leg["cpu_sample"] = "960x540 crop of the same frame, 1 thread, %.1f s" % c
    out["segment_4k"] = leg
    ctx.close()

    # ---- 3840x2160 Farneback (configs[4] workload on one GPU: 8 DIFFERENT pairs, seeds 1234..1241) ----
    w, h = 3840, 2160
    alg4 = algorithmic_bytes_per_pair(w, h)
    leg = {"workload": "Farneback flow, 3840x2160 gray pairs resident in HBM, 8 different pairs per GPU (BASELINE configs[4], seeds 1234..1241) as ONE "
                       "batched call of 8 at a time; direct-window mode: 4 single-pair calls in flight", "unit": "frame-pairs/s"}
    c0 = ofxcv.Context(dev)
    grays = []
    for seed in range(1234, 1242):
        a4, b4 = synth.flow_pair(w, h, seed=seed)
        grays.append((c0.to_byte_grayscale(torch.from_numpy(a4).cuda()), c0.to_byte_grayscale(torch.from_numpy(b4).cuda())))
    torch.cuda.synchronize()
    del a4, b4
    for direct, key in ((False, "value"), (True, "value_direct_window")) if direct_leg else ((False, "value"),):
        ns, nb = (4, 1) if direct else (1, 8)
        cs = [ofxcv.Context(dev) for _ in range(ns)]
        bufs = []
        for i, c in enumerate(cs):
            c.set_option("farneback.opencv_rounding", 0 if direct else 1)
            prs = grays[i * nb:(i + 1) * nb]
            bufs.append(([p[0] for p in prs], [p[1] for p in prs], [torch.empty((h, w, 2), device="cuda") for _ in range(nb)]))

        def step4():
            for c, (ga, gb, fl) in zip(cs, bufs):
                with torch.cuda.stream(c.stream):
                    c.calc_optical_flow_farneback_batch(ga, gb, fl)
        for _ in range(3):
            step4()
        rs = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step4()
            torch.cuda.synchronize()
            rs.append(4 * ns * nb / (time.perf_counter() - t0))
        leg[key] = med(rs)
        leg[key + "_frac_of_hbm_peak"] = alg4 * med(rs) / 1e9 / HBM_PEAK_GBS
        if not direct:
            flows4 = [f.cpu().numpy() for f in bufs[0][2]]
            single = [c0.calc_optical_flow_farneback(ga, gb).cpu().numpy() for ga, gb in grays[:2]]
            par4 = {"batch_equals_single_calls_bit_for_bit_pairs_0_1": bool(all(np.array_equal(x, y) for x, y in zip(single, flows4)))}
            if with_cpu:
                from oracle import binding as oracle
                ref = oracle.calc_optical_flow_farneback(grays[0][0].cpu().numpy(), grays[0][1].cpu().numpy(), PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N,
                                                         POLY_SIGMA, 0, oracle.BLUR_FAITHFUL)
                err = np.abs(flows4[0] - ref)
                par4.update({"reference": "CPU oracle (FAITHFUL), pair 0 of the batch", "outside_1e-4": float((err > 1e-4 * np.maximum(1, np.abs(ref))).mean()),
                             "max_abs_err": float(err.max()), "bit_identical": float((flows4[0] == ref).mean())})
            leg["parity"] = par4
            del flows4
        for c in cs:
            c.close()
        del bufs
    c0.close()
    out["farneback_4k"] = leg
    out["end_to_end"] = end_to_end   # last: the driver keeps the tail of the line
    return out


if __name__ == "__main__":
    main()
