#!/usr/bin/env python3
"""One continuous-batching iteration of the bench model as a staggered server sees it: c−1 decode tokens + one fresh
256-token prompt in the same unified forward (development aid; run under rocprofv3 --kernel-trace for tools/trace_summary.py).
Usage: mixed_iteration.py [model] [concurrency]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-30b-a3b"
c = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dict(bench.MODELS[name])
model = bench.build_model(pkg, cfg, c + 8, 512, 8192, 9271)
rng = np.random.default_rng(1)
prompts = [rng.integers(256, cfg["vocab"], size=256).astype(np.uint32) for _ in range(c - 1)]
first = bench.prefill(model, prompts, 0, 8192)
toks = model.decode_steps(list(range(c - 1)), first, 4)[-1]
pos = 256 + 4
for rep in range(6):
    p = rng.integers(256, cfg["vocab"], size=256).astype(np.uint32)
    items = [(i, np.array([toks[i]], np.uint32), pos, True) for i in range(c - 1)] + [(1000 + rep, p, 0, True)]
    time.sleep(0.02)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, _ = model.unified_forward(items, greedy=True)
    torch.cuda.synchronize()
    print(f"mixed iteration ({c - 1} decode tokens + 256-token prompt): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    toks = out[:c - 1]
    pos += 1
    model.release(1000 + rep)
