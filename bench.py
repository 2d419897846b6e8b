#!/usr/bin/env python3
"""bench.py -- Farneback dense optical flow throughput at 1920x1080 (BASELINE.json metric).

One step = one pass of the VectorGenerator hot path over one synthetic f32 RGBA frame pair that is
already resident in HBM: sRGB-gray LUT x2 (F0) -> calcOpticalFlowFarneback (F1-F6) -> flow->RGBA
write-back (F7), all through the C ABI of libofxcv_hip.so.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank runs its own independent pairs -- the path has no
exchange step, so there is no data-path collective; only the timing barrier / max-reduce.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the fused blur+solve+update
iteration at pyramid level 0, timed live with HIP events on its own stream) and `cpu_baseline`
(the CPU oracle -- a port of the reference's OpenCV algorithm -- timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
LEVELS, ITERS, POLY_N, POLY_SIGMA, WINSIZE, PYR_SCALE = 3, 15, 5, 1.1, 3, 0.5  # VectorGenerator.cpp:804-834, :391-395
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec peak
# one fused iteration = M-in 20 + R0 20 + R1 gather 20 + M-out 20 bytes per pixel; the dominant
# kernel (iterate3x2_kernel) runs TWO iterations per launch, i.e. 160 algorithmic bytes per pixel per launch
ITER_BYTES_PER_PX = 160.0


def algorithmic_bytes_per_pair(w, h, levels=LEVELS, iters=ITERS):
    """per level 1282*n + 2*N0 bytes (coarsest level 1272*n + 2*N0), u8 source."""
    n0 = w * h
    total = 0.0
    for k in range(levels + 1):
        n = round(w * 0.5 ** k) * round(h * 0.5 ** k)
        per = 2 * n0 + 2 * 4 * n + 48 * n + (10 * n if k < levels else 0) + 68 * n + (iters - 1) * 80 * n + 28 * n
        total += per
    return total


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (tools/pmc_bench.sh,
    summary in profiles/): the counters need their own profiler runs, so the figure is measured offline and read here."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(ga, gb, budget_s=12.0):
    """Time the CPU oracle (kind 'port': restatement of OpenCV's single-threaded CPU Farneback) on rank 0."""
    from oracle import binding as oracle
    oracle.lib()
    t0 = time.perf_counter()
    n = 0
    while True:
        oracle.calc_optical_flow_farneback(ga, gb, PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0, oracle.BLUR_FAITHFUL)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 8:
            break
    return {"value": n / el, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d Farneback frame pairs at %dx%d (same synthetic frames, levels=3 iterations=15 poly_n=5), "
                      "oracle/farneback.c single thread, %.1f s" % (n, W, H, el)}


def cpu_baseline_all_cores(ga, gb, per_thread=4, max_threads=32):
    """The same port, one independent frame pair per host thread (ctypes drops the GIL): what the host CPU of this box
    delivers on the partitioned workload.  SURVEY.md 8(d) asks for the single-thread and the all-cores figure."""
    import concurrent.futures as cf
    from oracle import binding as oracle
    oracle.lib()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=3, help="independent frame pairs per step, each on its own context/stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--size", default="1920x1080", help="frame size; the metric is quoted at 1920x1080 (BASELINE.json configs[2]), "
                    "3840x2160 is configs[4] (64 pairs over 8 GPUs); the PMC traffic figure is only recorded for 1920x1080")
    args = ap.parse_args()
    global W, H
    W, H = (int(v) for v in args.size.lower().split("x"))

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
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # One step = one batch of `pairs` independent frame pairs, each on its own context (own HIP stream and
    # scratch): frame pairs never exchange data, so they shard across streams exactly as they shard across GPUs.
    P = max(1, args.pairs)
    ctxs = [ofxcv.Context(local_rank) for _ in range(P)]
    for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):  # A/B of kernel variants: name=value,...
        for c in ctxs:
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    bufs = []
    for i, c in enumerate(ctxs):
        a, b = synth.flow_pair(W, H, seed=sharding.seed_for_pair(sharding.pairs_for_rank(world * P, rank, world)[i]))
        with torch.cuda.stream(c.stream):
            bufs.append(dict(a=torch.from_numpy(a).cuda(), b=torch.from_numpy(b).cuda(),
                             ga=torch.empty((H, W), dtype=torch.uint8, device="cuda"),
                             gb=torch.empty((H, W), dtype=torch.uint8, device="cuda"),
                             flow=torch.empty((H, W, 2), dtype=torch.float32, device="cuda"),
                             out=torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")))

    def step():
        for c, t in zip(ctxs, bufs):
            with torch.cuda.stream(c.stream):
                c.to_byte_grayscale(t["a"], t["ga"])
                c.to_byte_grayscale(t["b"], t["gb"])
                c.calc_optical_flow_farneback(t["ga"], t["gb"], t["flow"], PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)
                c.flow_to_rgba(t["flow"], t["out"], 0b0001, 0b0010)  # forward.u -> R, forward.v -> G (defaults :739,753)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # roofline leg: same process, same inputs -- HIP event pairs around every launch of the dominant kernel (the fused
    # iteration at pyramid level 0), recorded on the stream the kernel is launched on.  One pair at a time here: with
    # two streams in flight an event pair would also time the wait for the other stream's kernel.
    c0, t0b = ctxs[0], bufs[0]
    c0.profile_enable(True)
    with torch.cuda.stream(c0.stream):
        for _ in range(max(3, min(10, args.steps))):
            c0.calc_optical_flow_farneback(t0b["ga"], t0b["gb"], t0b["flow"], PYR_SCALE, LEVELS, WINSIZE, ITERS, POLY_N, POLY_SIGMA, 0)
    torch.cuda.synchronize()
    kern_ms, kern_n = c0.profile_read()
    c0.profile_enable(False)
    # for reference, the same event pairs with all `pairs` streams in flight (what a kernel trace of the timed region shows)
    conc_ms, conc_n = 0.0, 0
    if P > 1:
        for c in ctxs:
            c.profile_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        for c in ctxs:
            ms, n = c.profile_read()
            conc_ms += ms
            conc_n += n
            c.profile_enable(False)
    g_a, g_b = bufs[0]["ga"], bufs[0]["gb"]

    elapsed = sharding.reduce_elapsed_max(elapsed, dist, red_dev)        # MAX over ranks
    pairs = sharding.reduce_count_sum(args.steps * P, dist, red_dev)       # units all ranks processed

    if rank == 0:
        value = pairs / elapsed
        avg_s = kern_ms / 1e3 / max(1, kern_n)
        achieved = ITER_BYTES_PER_PX * W * H / avg_s / 1e9
        alg = algorithmic_bytes_per_pair(W, H)
        line = {
            "metric": "frames/sec at %dx%d f32 (Farneback flow)" % (W, H),
            "value": value,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "VectorGenerator Farneback dense optical flow, %dx%d f32 RGBA frame pair resident in HBM "
                                   "-> 8-bit sRGB gray -> calcOpticalFlowFarneback -> flow RGBA (BASELINE.json configs[%d])"
                                   % (W, H, 4 if (W, H) == (3840, 2160) else 2),
                       "levels": LEVELS, "iterations": ITERS, "poly_n": POLY_N, "poly_sigma": POLY_SIGMA, "winsize": WINSIZE,
                       "pyr_scale": PYR_SCALE, "pairs_per_step_per_gpu": P, "streams_per_gpu": P, "parallelism": "independent frame pairs per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic() if (W, H) == (1920, 1080) else None, "kernel": "iterate3x2_kernel<true> (two fused blur+solve+update iterations per launch; <true> = the pyramid level 0 launches, %dx%d)" % (W, H),
                         "bytes_per_launch": ITER_BYTES_PER_PX * W * H, "avg_launch_us": avg_s * 1e6, "launches_timed": kern_n,
                         "timing": "HIP event pairs on the launch stream, one frame pair in flight (compare profiles/r01_bench_pairs1_kernel_stats.csv)",
                         "avg_launch_us_all_streams_in_flight": (conc_ms / conc_n * 1e3) if conc_n else None},
            "whole_call": {"algorithmic_bytes_per_pair": alg, "achieved_GBps": alg * value / world / 1e9,
                           "frac_of_hbm_peak": alg * value / world / 1e9 / HBM_PEAK_GBS},
        }
        if not args.no_cpu_baseline and world == 1:
            ga = g_a.cpu().numpy()
            gb = g_b.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(ga, gb)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(ga, gb)
        elif world > 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
