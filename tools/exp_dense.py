#!/usr/bin/env python3
"""Micro-timings of the dense INT4 GEMM at the Llama-3.1-8B decode shapes through the C ABI.
Development aid: µs per launch and achieved GB/s of weight bytes, weights rotated over several copies."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
from oracle import oracle as O


def timeit(fn, reps=30):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def lin(k, n, seed):
    qw, sc, qz = O.make_synthetic_gptq(k, n, 128, seed, symmetric=True)
    return pkg.GptqLinear.from_raw(qw, sc.astype(np.float16).astype(np.float32), qz, None, None, 4, 128, k, n)


shapes = [(4096, 28672, "gate_up"), (14336, 4096, "down"), (4096, 6144, "qkv"), (4096, 4096, "o")]
only = [a for a in sys.argv[1:] if not a.startswith("-")]
for k, n, name in shapes:
    if only and name not in only: continue
    copies = max(2, min(6, int(300e6 // (k * n // 2))))
    lins = [lin(k, n, 10 + i) for i in range(copies)]
    wbytes = k * n // 2 + (k // 128) * n * 2
    for m in (32, 16, 1):
        xin = torch.randn(m, k, device="cuda").half(); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
        us = timeit(lambda i: lins[i % copies].forward(ctx, xin, out, m))
        print(f"{name:8s} K={k:5d} N={n:5d} m={m:2d}: {us:7.2f} us  {wbytes / us / 1e3:7.1f} GB/s", flush=True)
    del lins
