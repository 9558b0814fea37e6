#!/usr/bin/env python3
"""Micro-timings of individual hot-path ops through the C ABI (torch.cuda events on the op's stream).
Development aid; prints mean µs per launch with inputs rotated over `sets` distinct weight copies."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()


def timeit(fn, reps=50):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


T, H, E, K = 32, 2048, 128, 8
r = torch.randn(T, H, device="cuda").half(); x = torch.randn(T, H, device="cuda").half(); w = torch.ones(H, device="cuda").half()
nd = torch.empty_like(r)
routers = [B.dense_repack_f16t(ctx, (torch.randn(E, H, device="cuda") * 0.05).half(), E, H) for _ in range(8)]
ids = torch.empty(T, K, dtype=torch.int32, device="cuda"); wt = torch.empty(T, K, dtype=torch.float32, device="cuda")
print("B add+norm only (E=0)      %.2f us" % timeit(lambda i: B.fused_add_rms_norm_route(ctx, r, x, w, 1e-6, nd, None, 0, 0, True, None, None, None, T, H)))
print("B add+norm+router+topk     %.2f us" % timeit(lambda i: B.fused_add_rms_norm_route(ctx, r, x, w, 1e-6, nd, routers[i % 8], E, K, True, ids, wt, None, T, H)))
print("fused_add_rms_norm         %.2f us" % timeit(lambda i: B.fused_add_rms_norm(ctx, r, x, w, 1e-6, nd, T, H)))
lg = torch.empty(T, E, dtype=torch.float32, device="cuda")
print("gemm_f16t router (32x128)  %.2f us" % timeit(lambda i: B.gemm_f16t(ctx, nd, routers[i % 8], lg, T, E, H)))
print("route_topk_softmax         %.2f us" % timeit(lambda i: B.route_topk_softmax(ctx, lg, ids, wt, T, E, K, True)))

# ── INT4 dense projections at the Qwen3-30B-A3B decode shapes ──
from oracle import oracle as O
def lin(k, n, seed):
    qw, sc, qz = O.make_synthetic_gptq(k, n, 128, seed, symmetric=True)
    return pkg.GptqLinear.from_raw(qw, sc.astype(np.float16).astype(np.float32), qz, None, None, 4, 128, k, n)
if "--gemm" in sys.argv:
    for (k, n, name) in ((2048, 5120, "qkv"), (4096, 2048, "o")):
        lins = [lin(k, n, 10 + i) for i in range(6)]
        for m in (32, 1):
            xin = torch.randn(m, k, device="cuda").half(); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
            for nt, W in ((0, 0), (1, 8), (1, 16), (2, 8), (4, 8)):
                if nt: os.environ["FERRUM_HIP_W4_NT"] = str(nt); os.environ["FERRUM_HIP_W4_W"] = str(W)
                else: os.environ.pop("FERRUM_HIP_W4_NT", None); os.environ.pop("FERRUM_HIP_W4_W", None)
                try:
                    us = timeit(lambda i: lins[i % 6].forward(ctx, xin, out, m))
                    print(f"{name} m={m} NT={nt} W={W}: {us:.2f} us")
                except Exception as ex:
                    print(f"{name} m={m} NT={nt} W={W}: {ex}")
    os.environ.pop("FERRUM_HIP_W4_NT", None); os.environ.pop("FERRUM_HIP_W4_W", None)
