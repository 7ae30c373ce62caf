#!/usr/bin/env python3
"""How much does a cross-stream dependency cost on this part?  A chain of tiny dependent kernels (a) on ONE stream, (b) alternating
between TWO streams with an event record + wait at every hand-over — the shape of a Cholesky lookahead with one hand-over per
64-column step (tools/micro: a measurement, not product code)."""
import time
import torch

x = torch.zeros(1024, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 400


def one_stream():
    with torch.cuda.stream(s1):
        for _ in range(2 * n):
            x.add_(1.0)


def two_streams():
    ev = [torch.cuda.Event() for _ in range(2 * n)]
    for i in range(n):
        with torch.cuda.stream(s1):
            if i:
                s1.wait_event(ev[2 * i - 1])
            x.add_(1.0)
            ev[2 * i].record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(ev[2 * i])
            x.add_(1.0)
            ev[2 * i + 1].record(s2)


def side_wait_only():
    """main stream runs its chain; every step it waits for an event of the side stream recorded long ago (already signalled)"""
    ev = [torch.cuda.Event() for _ in range(n)]
    with torch.cuda.stream(s2):
        for i in range(n):
            x.add_(1.0)
            ev[i].record(s2)
    s2.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        for i in range(n):
            s1.wait_event(ev[i])
            x.add_(1.0)
            x.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, fn in (("one stream, 2 kernels per iteration", one_stream), ("two streams, hand-over after every kernel", two_streams)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per iteration")
side_wait_only()
print(f"one stream + a wait on an already-signalled event per iteration: {side_wait_only():.1f} us per iteration")
