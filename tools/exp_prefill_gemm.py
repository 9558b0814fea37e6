#!/usr/bin/env python3
"""Micro-timings of the dense INT4 GEMM at prefill row counts (the 64-row tile kernel) through the C ABI.
Development aid: µs per launch and achieved TFLOP/s. Usage: exp_prefill_gemm.py [shape-name ...] [-m ROWS,ROWS]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
from oracle import oracle as O


def timeit(fn, reps=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def lin(k, n, seed):
    qw, sc, qz = O.make_synthetic_gptq(k, n, 128, seed, symmetric=True)
    return pkg.GptqLinear.from_raw(qw, sc.astype(np.float16).astype(np.float32), qz, None, None, 4, 128, k, n)


shapes = [(2048, 5120, "q3-qkv"), (4096, 2048, "q3-o"), (4096, 6144, "l8-qkv"), (4096, 4096, "l8-o"),
          (4096, 28672, "l8-gate_up"), (14336, 4096, "l8-down")]
ms = [2048]
args = [a for a in sys.argv[1:] if not a.startswith("-big=")]
if "-m" in args:
    i = args.index("-m"); ms = [int(x) for x in args[i + 1].split(",")]; del args[i:i + 2]
for k, n, name in shapes:
    if args and name not in args: continue
    lins = [lin(k, n, 10 + i) for i in range(2)]
    for m in ms:
        xin = torch.randn(m, k, device="cuda").half(); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
        us = timeit(lambda i: lins[i % 2].forward(ctx, xin, out, m))
        print(f"{name:11s} K={k:5d} N={n:5d} m={m:5d}: {us:8.2f} us  {2.0 * m * k * n / us / 1e6:7.1f} TFLOP/s", flush=True)
        bigs = [int(a[5:]) for a in sys.argv[1:] if a.startswith("-big=")]       # -big=8 -big=16: w4_gemm_big_kernel beside the default
        if bigs:
            lib = pkg.backend._lib
            lins[0].forward(ctx, xin, out, m); torch.cuda.synchronize(); ref = out.float().clone()
            for mtv in bigs:
                os.environ["FERRUM_HIP_W4_BIG"] = str(mtv); lib.ferrum_hip_debug_reload_knobs()
                us = timeit(lambda i: lins[i % 2].forward(ctx, xin, out, m))
                lins[0].forward(ctx, xin, out, m); torch.cuda.synchronize()
                d = (out.float() - ref); rel = d.norm().item() / ref.norm().item()
                print(f"    big MT={mtv:2d}: {us:8.2f} us  {2.0 * m * k * n / us / 1e6:7.1f} TFLOP/s   rel L2 diff vs default {rel:.2e}  max {d.abs().max().item():.4f}", flush=True)
                os.environ.pop("FERRUM_HIP_W4_BIG"); lib.ferrum_hip_debug_reload_knobs()
    del lins
