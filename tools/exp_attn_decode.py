#!/usr/bin/env python3
"""Decode paged attention alone at growing context: µs per launch and achieved GB/s of K+V bytes (development aid /
roofline evidence).  Usage: exp_attn_decode.py [c] [kv_len,kv_len,...]  (Qwen3-30B-A3B heads: 32 q / 4 kv × 128)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kvs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [328, 1024, 4096, 16384]
nq, nkv, hd = 32, 4, 128
for kv in kvs:
    nb = (kv + 15) // 16
    reps = 3                                            # rotate over copies of the pool so the Infinity Cache cannot serve it
    pools = [(torch.randn(c * nb * nkv * 16 * hd, device="cuda").half(), torch.randn(c * nb * nkv * 16 * hd, device="cuda").half())
             for _ in range(reps)]
    tables = torch.from_numpy(np.random.default_rng(0).permutation(c * nb).astype(np.int32).reshape(c, nb)).cuda()
    lens = torch.full((c,), kv, dtype=torch.int32, device="cuda")
    q = torch.randn(c, nq, hd, device="cuda").half(); out = torch.empty_like(q)
    def run(i):
        k, v = pools[i % reps]
        B.paged_batched_decode_attention(ctx, q, k, v, out, tables, lens, c, kv, nq, nkv, hd, 16, nb)
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byts = c * kv * nkv * hd * 2 * 2
    print(f"c={c} kv={kv:6d}: {us:8.2f} us  {byts / 1e6:8.1f} MB  {byts / us / 1e3:7.1f} GB/s  ({byts / us / 1e3 / 8000 * 100:.1f} % of 8 TB/s)", flush=True)
    del pools
