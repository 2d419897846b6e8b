#!/usr/bin/env python3
"""cost of hipHostRegister / hipHostUnregister of one 1080p f32 RGBA frame (33 MB), fresh and repeated"""
import ctypes as C, time, numpy as np, torch
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
for trial in range(3):
    a = np.ones((1080, 1920, 4), np.float32)
    p = C.c_void_p(a.ctypes.data)
    t0 = time.perf_counter(); rc = hip.hipHostRegister(p, C.c_size_t(a.nbytes), 0); t1 = time.perf_counter()
    rc2 = hip.hipHostUnregister(p); t2 = time.perf_counter()
    t3 = time.perf_counter(); hip.hipHostRegister(p, C.c_size_t(a.nbytes), 0); t4 = time.perf_counter(); hip.hipHostUnregister(p); t5 = time.perf_counter()
    print("register %.2f ms (rc %d), unregister %.2f ms (rc %d); again: register %.2f ms, unregister %.2f ms" % ((t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, rc2, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
