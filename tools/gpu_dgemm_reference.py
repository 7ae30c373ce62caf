#!/usr/bin/env python3
"""Calibration point for the fp64 MFMA roofline: what the vendor DGEMM (torch.mm -> rocBLAS / hipBLASLt) reaches on
this box for a large square product and for the shape of one k_predict launch (dense, no triangular clipping)."""
import json
import torch


def rate(m, n, k, reps=10):
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    b = torch.randn(k, n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"m": m, "n": n, "k": k, "ms": round(ms, 3), "tflops": round(2.0 * m * n * k / (ms * 1e-3) / 1e12, 1)}


if __name__ == "__main__":
    for shape in ((4096, 4096, 4096), (8192, 8192, 8192), (2048, 20480, 2048), (2048, 2048, 2048)):
        print(json.dumps(rate(*shape)), flush=True)
