#!/usr/bin/env python3
"""Run only prefill forwards (chunk/256 prompts x 256 tokens each, default 2048 query tokens) of the bench model: for rocprofv3 --stats.
Usage: prefill_only.py [model] [chunk_tokens]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-30b-a3b"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = dict(bench.MODELS[name])
model = bench.build_model(pkg, cfg, 32, 512, chunk, 9271)
rng = np.random.default_rng(1)
for rep in range(3):
    prompts = [rng.integers(256, cfg["vocab"], size=256).astype(np.uint32) for _ in range(chunk // 256)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.prefill(model, prompts, 100 * rep, chunk)
    torch.cuda.synchronize()
    print(f"prefill {chunk} tokens: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    for s in range(100 * rep, 100 * rep + chunk // 256): model.release(s)
for rep in range(3):
    p = rng.integers(256, cfg["vocab"], size=256).astype(np.uint32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.unified_forward([(900 + rep, p, 0, True)], greedy=True)
    torch.cuda.synchronize()
    print(f"prefill 256 tokens (TTFT c=1): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    model.release(900 + rep)
