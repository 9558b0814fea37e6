#!/usr/bin/env python3
"""Per-workgroup timeline of the one-launch decode chain (chain.hip) inside a real decode step of a synthetic Qwen3-30B-A3B
(a few layers): wall-clock stamps (100 MHz) at role entry, after the role's wait, before its signal, at its exit.
Needs an EXPERIMENTS build (see tools/exp_timeline_pair.py).  usage: exp_timeline_chain.py [c]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
lib = pkg.load_library()
lib.ferrum_hip_debug_set_chain_timeline.argtypes = [C.c_void_p]
lib.ferrum_hip_debug_set_chain_timeline.restype = None
c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PL, LAYERS = 256, 6
cfg = bench.QWEN3_30B_A3B
model = bench.build_model(pkg, cfg, c, PL + 64, c * PL, 1234, layers=LAYERS)
rng = np.random.default_rng(9271)
prompts = [rng.integers(256, 151936, size=PL).astype(np.uint32) for _ in range(c)]
toks = bench.prefill(model, prompts, 0, c * PL)
pos = PL
for _ in range(4):
    toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
    pos += 1
tl = torch.zeros(1 << 14, dtype=torch.int64, device="cuda")
lib.ferrum_hip_debug_set_chain_timeline(C.c_void_p(tl.data_ptr()))
toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
torch.cuda.synchronize()
lib.ferrum_hip_debug_set_chain_timeline(None)
t = tl.cpu().numpy().reshape(-1, 4)
rh = (c + 15) // 16
half = c <= 16                                   # ≤ 16 rows: 32-column blocks in the GEMM roles
qkv_cols = (cfg["num_heads"] + 2 * cfg["num_kv_heads"]) * 128
wide = (not half) and (int(os.environ.get("FERRUM_HIP_CHAIN_QKV_WIDE", "-1")) > 0 or (int(os.environ.get("FERRUM_HIP_CHAIN_QKV_WIDE", "-1")) < 0 and qkv_cols // 64 * rh + c * cfg["num_kv_heads"] > 256))   # 128-column q|k|v blocks (chain.hip decode_chain_f16)
n_a, n_qkv, n_attn, n_o, n_b = c, qkv_cols // (32 if half else (128 if wide else 64)) * rh, c * cfg["num_kv_heads"], cfg["hidden"] // (32 if half else 64) * rh * 2, c * int(os.environ.get("FERRUM_HIP_ROUTE_PARTS", "4"))
tot = n_a + n_qkv + n_attn + n_o + n_b
t = t[:tot]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
q = lambda a: "min %6.2f  p50 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 50), a.max())
print(f"decode chain (last of {LAYERS} layers), c={c}: {tot} workgroups")
o = 0
for name, n in (("A    tail", n_a), ("qkv  gemm", n_qkv), ("attention", n_attn), ("o    gemm", n_o), ("B   route", n_b)):
    if n == 0: continue
    u = us[o:o + n]
    u = u[u[:, 3] > u[:, 0]] if name.startswith("B") else u
    print(f"{name} ({n:3d} wgs): entry {q(u[:, 0])} | wait done {q(u[:, 1])} | work done {q(u[:, 2])} | exit {q(u[:, 3])}")
    print(f"                    after-wait work: {q(u[:, 2] - u[:, 1])}   signal/drain: {q(u[:, 3] - u[:, 2])}")
    o += n
print(f"launch span by the stamps: {us[:, 3].max():.2f} us")
