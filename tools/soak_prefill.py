#!/usr/bin/env python3
"""Determinism soak of the prefill forms (tall-tile GEMMs, 96-pair MoE blocks, 64-key and resident-K/V attention): the same ragged
batch of prompts prefilled repeatedly must give bit-identical logits — a missing barrier or a clobbered register in the
double-buffered kernels shows up as a flaky bit.  Usage: soak_prefill.py [model] [layers] [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-30b-a3b"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = dict(bench.MODELS[name])
model = bench.build_model(pkg, cfg, 40, 8200, 8192, 9271, layers=layers)
rng = np.random.default_rng(3)
cases = {"32 x 256": [256] * 32, "ragged": [1900, 700, 256, 97, 33, 512, 1024, 1300, 64, 200, 17, 5],
         "one long": [8000], "short mix": [100, 250, 31, 256, 180, 64, 256, 222, 256, 90, 256, 256, 140, 256, 256, 256]}
bad = 0
for label, lens in cases.items():
    prompts = [rng.integers(256, cfg["vocab"], size=n).astype(np.uint32) for n in lens]
    ref = None
    for r in range(reps):
        toks, lg = model.unified_forward([(i, p, 0, True) for i, p in enumerate(prompts)], greedy=True, want_logits=True)
        torch.cuda.synchronize()
        cur = (np.array(toks).copy(), np.array(lg).copy())
        for i in range(len(prompts)): model.release(i)
        if ref is None: ref = cur
        else:
            same = np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1])
            if not same:
                bad += 1
                print(f"{label}: repeat {r} differs: ids equal {np.array_equal(ref[0], cur[0])}, max |dlogit| {np.abs(ref[1] - cur[1]).max():.3e}", flush=True)
    print(f"{label}: {sum(lens)} tokens x {reps} repeats done, finite {np.isfinite(ref[1]).all()}", flush=True)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
