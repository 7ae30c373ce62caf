#!/usr/bin/env python3
"""Achievable HBM bandwidth for pure writes, pure reads and copies on this part (torch fill_ / sum / copy_ on 2 GiB, HIP events):
the K(X,X) assembly and the cross kernel are write streams."""
import torch
n = (2 << 30) // 8
x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, fn, nbytes in (("fill_ (write 2 GiB)", lambda: x.fill_(1.5), 2 << 30), ("zero_ (write)", lambda: x.zero_(), 2 << 30),
                         ("sum (read 2 GiB)", lambda: x.sum(), 2 << 30), ("copy_ (read 2 + write 2 GiB)", lambda: y.copy_(x), 4 << 30),
                         ("mul_ in place (read + write)", lambda: x.mul_(1.0000001), 4 << 30)):
    s = t(fn)
    print(f"{name}: {s * 1e3:.3f} ms = {nbytes / s / 1e12:.2f} TB/s")
