#!/usr/bin/env python3
"""tools/probe/r06_pcie.py -- what the host link of this box gives: pinned host <-> device copies of 16 / 64 / 256 MB,
H2D alone, D2H alone, and both at once on two streams.  (Run once more under HSA_ENABLE_SDMA=0: copies by shader
kernels instead of the DMA engines.)  Diagnostic; bench.py's host_to_host records print the same ceiling for their own
sub-batch sizes."""
import os
import time

import torch

dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA", "(default)"))
for mb in (16, 64, 256):
    n = mb << 20
    h_a = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_b = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(n, dtype=torch.uint8, device=dev)
    d_b = torch.empty(n, dtype=torch.uint8, device=dev)
    reps = max(4, 2048 // mb)

    def run(h2d, d2h):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_a.copy_(h_a, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_b.copy_(d_b, non_blocking=True)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        return (int(h2d) + int(d2h)) * n * reps / dt / 1e9

    run(True, True)
    print("%4d MB: H2D %.1f GB/s   D2H %.1f GB/s   both at once %.1f GB/s (sum of the two directions)" % (
        mb, run(True, False), run(False, True), run(True, True)))

# ---- the host_to_host pass's own pattern: 32 pieces, 16 MB in and 23.6 / 31.5 MB out each
for out_mb in (23.6, 31.5):
    P, NB = 32, 4
    n_in, n_out = 16 << 20, int(out_mb * (1 << 20))
    h_in = [torch.empty(n_in, dtype=torch.uint8).pin_memory() for _ in range(P)]
    h_out = [torch.empty(n_out, dtype=torch.uint8).pin_memory() for _ in range(P)]
    d_in = [torch.empty(n_in, dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_out = [torch.empty(n_out, dtype=torch.uint8, device=dev) for _ in range(NB)]
    sc = [torch.cuda.Stream(dev) for _ in range(NB)]
    ev_in = [torch.cuda.Event() for _ in range(NB)]
    ev_c = [torch.cuda.Event() for _ in range(NB)]
    ev_out = [torch.cuda.Event() for _ in range(NB)]

    def free_pass():
        for i in range(P):
            with torch.cuda.stream(s1):
                d_in[i % NB].copy_(h_in[i], non_blocking=True)
            with torch.cuda.stream(s2):
                h_out[i].copy_(d_out[i % NB], non_blocking=True)
        torch.cuda.synchronize(dev)

    def chained_pass():
        for i in range(P):
            b = i % NB
            with torch.cuda.stream(s1):
                if i >= NB:
                    s1.wait_event(ev_c[b])
                d_in[b].copy_(h_in[i], non_blocking=True)
                ev_in[b].record(s1)
            with torch.cuda.stream(sc[b]):
                sc[b].wait_event(ev_in[b])
                if i >= NB:
                    sc[b].wait_event(ev_out[b])
                ev_c[b].record(sc[b])
            with torch.cuda.stream(s2):
                s2.wait_event(ev_c[b])
                h_out[i].copy_(d_out[b], non_blocking=True)
                ev_out[b].record(s2)
        torch.cuda.synchronize(dev)

    for name, f in (("no events", free_pass), ("event edges", chained_pass)):
        f()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        dt = (time.perf_counter() - t0) / 5
        print("32 x (16 MB in, %.1f MB out), %s: %.2f ms per pass = %.1f GB/s both ways" % (
            out_mb, name, dt * 1e3, P * (n_in + n_out) / dt / 1e9))
    del h_in, h_out, d_in, d_out
