#!/usr/bin/env python3
"""Prefill paged attention alone (paged_varlen_attention through the C ABI) at growing prompt lengths: µs per launch and
achieved TFLOP/s (4·nq·hd flops per query-key pair under the causal mask).  Qwen3-30B-A3B heads: 32 q / 4 kv × 128.
Development aid / evidence for the LDS-shared K/V form; FERRUM_HIP_ATTN_NO_FLASH=1 gives the row-split form for comparison."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
nq, nkv, hd = 32, 4, 128


def run(q_lens, pos_offs, label):
    S = len(q_lens); kv_lens = [p + t for p, t in zip(pos_offs, q_lens)]
    max_blocks = (max(kv_lens) + 15) // 16
    nb = S * max_blocks
    k = torch.randn(nb * nkv * 16 * hd, device="cuda").half(); v = torch.randn(nb * nkv * 16 * hd, device="cuda").half()
    tables = torch.arange(nb, dtype=torch.int32, device="cuda").reshape(S, max_blocks)
    m_total = sum(q_lens)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)).cuda()
    po = torch.tensor(np.array(pos_offs, np.int32)).cuda()
    q = torch.randn(m_total, nq, hd, device="cuda").half(); out = torch.empty_like(q)

    def f():
        B.paged_varlen_attention(ctx, q, k, v, out, cu, po, tables, S, m_total, max(kv_lens), nq, nkv, hd, 0, 16, max_blocks, max(q_lens))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    fl = sum(4.0 * nq * hd * (t * p + t * (t + 1) / 2) for t, p in zip(q_lens, pos_offs))
    print(f"{label:28s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


run([256] * 32, [0] * 32, "32 x 256")
run([2048] * 4, [0] * 4, "4 x 2048")
run([8192], [0], "1 x 8192")
run([2048], [6144], "chunk 2048 at offset 6144")
