#!/usr/bin/env python3
"""What the pieces of the host path cost on this box: host memcpy rate (1 / 2 / 4 threads), pinned H2D / D2H rate,
hipHostRegister + unregister of a 33 MB frame, pageable H2D."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
W, H = 1920, 1080
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
src = [np.random.rand(H, W, 4).astype(np.float32) for _ in range(4)]
dst = [np.empty_like(s) for s in src]
for nt in (1, 2, 4):
    def work(i):
        for _ in range(20):
            np.copyto(dst[i], src[i])
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    el = time.perf_counter() - t0
    print("memcpy %d threads: %.1f GB/s total" % (nt, nt * 20 * src[0].nbytes / el / 1e9))
pin = torch.empty((H, W, 4), dtype=torch.float32).pin_memory()
dev = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for name, fn in (("pinned H2D", lambda: dev.copy_(pin, non_blocking=True)), ("pinned D2H", lambda: pin.copy_(dev, non_blocking=True)),
                 ("pageable H2D", lambda: dev.copy_(torch.from_numpy(src[0])))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print("%s: %.1f GB/s" % (name, 10 * src[0].nbytes / (time.perf_counter() - t0) / 1e9))
t = torch.from_numpy(src[1])
cudart = torch.cuda.cudart()
t0 = time.perf_counter()
for _ in range(10):
    cudart.cudaHostRegister(t.data_ptr(), t.numel() * 4, 0)
    cudart.cudaHostUnregister(t.data_ptr())
print("register + unregister 33 MB: %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
