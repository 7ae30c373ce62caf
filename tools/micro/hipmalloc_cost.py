#!/usr/bin/env python3
"""hipMalloc / hipFree / hipMemset cost by size (what a context's set-up pays per buffer)."""
import ctypes, time
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
for mb in (1, 8, 67, 528):
    n = mb << 20
    ts = []
    for _ in range(5):
        p = ctypes.c_void_p()
        t0 = time.perf_counter(); assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0; t1 = time.perf_counter()
        hip.hipMemset(p, 0, ctypes.c_size_t(n)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = [sorted(x)[2] * 1e3 for x in zip(*ts)]
    print(f"{mb:4d} MB: hipMalloc {a[0]:.3f} ms, memset+sync {a[1]:.3f} ms, hipFree {a[2]:.3f} ms")
