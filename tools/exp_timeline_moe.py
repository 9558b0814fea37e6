#!/usr/bin/env python3
"""Per-wave timeline of the expert-major MoE grouped GEMM (w4_gemm_moe_em_kernel) inside a real decode step of a synthetic
Qwen3-30B-A3B (a few layers): wall-clock stamps (100 MHz) at wave entry, after the routing scan, at loop end and after the stores.
Development aid; needs `make -C ferrum-infer-rs_amd/csrc EXPERIMENTS=1`.   usage: exp_timeline_moe.py [gate_up|down] [c]"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "gate_up"
c = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
lib.ferrum_hip_debug_set_timeline_mode(2 if which == "gate_up" else 1)
lib.ferrum_hip_debug_set_timeline(C.c_void_p(tl.data_ptr()))
toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
torch.cuda.synchronize()
lib.ferrum_hip_debug_set_timeline(None)
t = tl.cpu().numpy().reshape(-1, 4)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
q = lambda a: "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
work = t[:, 3] > t[:, 1]          # (a wave without pairs leaves after stamp 1; older launches left stamps 2, 3 behind)
print(f"{which} (last of {LAYERS} layers), c={c}: waves stamped {len(t)}, with pairs {int(work.sum())}")
print("wave entry           (us after the first wave): " + q(us[:, 0]))
late = us[:, 0] > 3.0
print(f"  waves entering later than 3 us: {int(late.sum())} of {len(t)} (resident at once: {len(t) - int(late.sum())} = {(len(t) - int(late.sum())) / 256:.1f} per CU)")
print("routing scan done - entry                     : " + q(us[:, 1] - us[:, 0]))
print("empty experts leave at                        : " + (q(us[~work, 1]) if (~work).any() else "-"))
print("loop end - scan done (16 / 6 groups streamed) : " + q(us[work, 2] - us[work, 1]))
print("stores done - loop end                        : " + q(us[work, 3] - us[work, 2]))
print("wave exit            (us after the first wave): " + q(us[work, 3]))
hist, edges = np.histogram(us[work, 3], bins=12)
print("exit histogram: " + "  ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}us:{hist[i]}" for i in range(len(hist))))
print(f"kernel span by the stamps: {us[work, 3].max():.2f} us")
# where do the slow waves sit?  workgroup id = e * n64 + st (x fastest); consecutive ids go to consecutive XCDs
ids = np.nonzero(tl.cpu().numpy().reshape(-1, 4)[:, 0] != 0)[0]
n64 = 24 if which == "gate_up" else 32
xcd, st_, ex = ids % 8, ids % n64, ids // n64
dur = us[:, 3] - us[:, 0]
print("median stream time by XCD (id % 8):   " + "  ".join(f"{x}:{np.median(dur[work & (xcd == x)]):.1f}" for x in range(8)))
print("median stream time by supertile st:   " + "  ".join(f"{x}:{np.median(dur[work & (st_ == x)]):.1f}" for x in range(0, n64, 3)))
print("median stream time by expert octile:  " + "  ".join(f"{x}:{np.median(dur[work & (ex // 16 == x)]):.1f}" for x in range(8)))
slot = ids % 1024
print("median stream time by id % 4 (SIMD?): " + "  ".join(f"{x}:{np.median(dur[work & ((ids // 8) % 4 == x)]):.1f}" for x in range(4)))
