#!/usr/bin/env python3
"""Per-workgroup timeline of the block-major merged gate_up → down launch (w4_gemm_moe_bm2_kernel) inside a real decode step of
a synthetic Qwen3-30B-A3B (a few layers): wall-clock stamps (100 MHz) at entry, after routing + align, at wait end (down) /
stores issued (gate_up), at exit.  EXPERIMENTS build (see tools/exp_timeline_pair.py).  usage: exp_timeline_bm2.py [c]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
lib = pkg.load_library()
lib.ferrum_hip_debug_set_timeline.argtypes = [C.c_void_p]
lib.ferrum_hip_debug_set_timeline.restype = None
c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
PL, LAYERS = 256, 6
model = bench.build_model(pkg, bench.QWEN3_30B_A3B, c, PL + 64, c * PL, 1234, layers=LAYERS)
rng = np.random.default_rng(9271)
prompts = [rng.integers(256, 151936, size=PL).astype(np.uint32) for _ in range(c)]
toks = bench.prefill(model, prompts, 0, c * PL)
pos = PL
for _ in range(4):
    toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
    pos += 1
tl = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
lib.ferrum_hip_debug_set_timeline_mode(3)
lib.ferrum_hip_debug_set_timeline(C.c_void_p(tl.data_ptr()))
toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
torch.cuda.synchronize()
lib.ferrum_hip_debug_set_timeline(None)
P = c * 8
max_blocks = min((P + 128 * 16) // 16, P // 16 + min(P, 128))
t = tl.cpu().numpy().reshape(-1, 4)[:max_blocks * 56]
q = lambda a: "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()) if len(a) else "-"
t0 = t[t[:, 0] != 0, 0].min()
us = (t - t0) / 100.0
tile = np.arange(len(t)) % 56
live = t[:, 3] > t[:, 1]
gu, dn = live & (tile < 24), live & (tile >= 24)
dead = (t[:, 0] != 0) & ~live
print(f"block-major merged launch (last of {LAYERS} layers), c={c}: grid {max_blocks} blocks x 56 tiles; gate_up tiles with work {int(gu.sum())}, down {int(dn.sum())}, empty {int(dead.sum())}")
print("entry (all)                   : " + q(us[t[:, 0] != 0, 0]))
print("routing + align (all)         : " + q((us[:, 1] - us[:, 0])[t[:, 0] != 0]))
print("empty workgroups leave (abs)  : " + q(us[dead, 1]))
print("gate_up routed (abs)          : " + q(us[gu, 1]))
print("gate_up stores issued (abs)   : " + q(us[gu, 2]))
print("gate_up loads+mfma+reduce     : " + q(us[gu, 2] - us[gu, 1]))
print("gate_up arrival published     : " + q(us[gu, 3]))
print("down routed (abs)             : " + q(us[dn, 1]))
print("down wait done (abs)          : " + q(us[dn, 2]))
print("down exit (abs)               : " + q(us[dn, 3]))
print("down work after the wait      : " + q(us[dn, 3] - us[dn, 2]))
print(f"launch span by the stamps: {us[live, 3].max():.2f} us")
