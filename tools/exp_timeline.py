#!/usr/bin/env python3
"""Per-wave timeline of w4_gemm_ldsa_kernel (17–32-row dense decode GEMM) at a Llama-3.1-8B shape: wall-clock stamps (100 MHz) at
wave entry, first consume, loop end and after the slab stores → where a launch's time goes (ramp, steady state, tail).
Development aid; needs `make -C ferrum-infer-rs_amd/csrc EXPERIMENTS=1`.   usage: exp_timeline.py [gate_up|down|qkv|o] [S]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
from oracle import oracle as O
lib = pkg.backend._lib
lib.ferrum_hip_debug_set_timeline.argtypes = [C.c_void_p]
lib.ferrum_hip_debug_set_timeline.restype = None
shapes = {"gate_up": (4096, 28672), "down": (14336, 4096), "qkv": (4096, 6144), "o": (4096, 4096)}
name = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
if len(sys.argv) > 2:
    os.environ["FERRUM_HIP_W4_LDSA_S"] = sys.argv[2]
    lib.ferrum_hip_debug_reload_knobs()
k, n = shapes[name]
m = 32
copies = 4
lins = []
for i in range(copies):
    qw, sc, qz = O.make_synthetic_gptq(k, n, 128, 10 + i, symmetric=True)
    lins.append(pkg.GptqLinear.from_raw(qw, sc.astype(np.float16).astype(np.float32), qz, None, None, 4, 128, k, n))
x = torch.randn(m, k, device="cuda").half(); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
tl = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
for i in range(6):
    lins[i % copies].forward(ctx, x, out, m)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    lins[i % copies].forward(ctx, x, out, m)
e1.record(); torch.cuda.synchronize()
print(f"{name} {k}->{n} m={m}: {e0.elapsed_time(e1) * 1e3 / 20:.2f} us per forward (GEMM + slab reduce), untimed-stamp run")
lib.ferrum_hip_debug_set_timeline(C.c_void_p(tl.data_ptr()))
lins[2].forward(ctx, x, out, m)
torch.cuda.synchronize()
lib.ferrum_hip_debug_set_timeline(None)
t = tl.cpu().numpy().reshape(-1, 4)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = (t - t0) / 100.0           # 100 MHz wall clock
q = lambda a: "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
print(f"waves stamped: {len(t)}")
print("wave entry            (us after the first wave): " + q(us[:, 0]))
print("first consume - entry (A staged, first W group): " + q(us[:, 1] - us[:, 0]))
print("loop end - first consume (streaming)           : " + q(us[:, 2] - us[:, 1]))
ok = t[:, 3] != 0
if ok.any():
    print("stores done - loop end                         : " + q(us[ok, 3] - us[ok, 2]))
    print("wave exit             (us after the first wave): " + q(us[ok, 3]))
    print(f"kernel span by the stamps: {us[ok, 3].max():.2f} us")
else:
    print(f"(direct fp16 output: no exit stamp)  loop end at most {us[:, 2].max():.2f} us after the first wave")
