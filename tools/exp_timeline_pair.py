#!/usr/bin/env python3
"""Per-wave timeline of the merged gate_up → down launch (w4_gemm_moe_em2_kernel) inside a real decode step of a synthetic
Qwen3-30B-A3B (a few layers): wall-clock stamps (100 MHz) at wave entry, after the routing scan, at loop end (gate_up) /
wait end (down), and after the stores.  Needs an EXPERIMENTS build:
  make -C ferrum-infer-rs_amd/csrc EXPERIMENTS=1 OBJDIR=../build_exp OUT=../lib_exp/libferrum_hip.so ../lib_exp/libferrum_hip.so
  FERRUM_HIP_LIB=ferrum-infer-rs_amd/lib_exp/libferrum_hip.so python tools/exp_timeline_pair.py [c]"""
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
c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
lib.ferrum_hip_debug_set_timeline_mode(3)          # (not the two-launch kernels)
lib.ferrum_hip_debug_set_timeline(C.c_void_p(tl.data_ptr()))
toks, _ = model.unified_forward([(i, [int(toks[i])], pos, True) for i in range(c)], greedy=True)
torch.cuda.synchronize()
lib.ferrum_hip_debug_set_timeline(None)
t = tl.cpu().numpy().reshape(-1, 4)
N_GU = 24 * 128
q = lambda a: "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()) if len(a) else "-"
t0 = t[t[:, 0] != 0, 0].min()
us = (t - t0) / 100.0
gu, dn = us[:N_GU], us[N_GU:N_GU + 32 * 128]
gu_ok, dn_ok = t[:N_GU, 0] != 0, t[N_GU:N_GU + 32 * 128, 0] != 0
gu_work = gu_ok & (t[:N_GU, 3] > t[:N_GU, 1])
dn_work = dn_ok & (t[N_GU:N_GU + 32 * 128, 3] > t[N_GU:N_GU + 32 * 128, 1])
print(f"merged launch (last of {LAYERS} layers), c={c}: gate_up waves {int(gu_ok.sum())} (with pairs {int(gu_work.sum())}), down waves {int(dn_ok.sum())} (with pairs {int(dn_work.sum())})")
print("gate_up entry                 : " + q(gu[gu_ok, 0]))
print("gate_up scan done - entry     : " + q(gu[gu_ok, 1] - gu[gu_ok, 0]))
print("gate_up loop+stores issued    : " + q(gu[gu_work, 2]))
print("gate_up arrival published     : " + q(gu[gu_work, 3]))
print("down entry                    : " + q(dn[dn_ok, 0]))
print("down wait done (abs)          : " + q(dn[dn_work, 2]))
print("down wait length              : " + q(dn[dn_work, 2] - dn[dn_work, 1]))
print("down exit (abs)               : " + q(dn[dn_work, 3]))
print("down work after the wait      : " + q(dn[dn_work, 3] - dn[dn_work, 2]))
# per expert: last gate_up arrival vs first / median down wait-done
ex_gu = np.arange(N_GU) // 24
ex_dn = np.arange(32 * 128) // 32
lag = []
for e in range(128):
    a = gu[(ex_gu == e) & gu_work, 3]
    b = dn[(ex_dn == e) & dn_work, 2]
    if len(a) and len(b):
        lag.append((a.max(), np.median(b) - a.max(), b.max() - a.max()))
lag = np.array(lag)
print("per expert: last arrival (abs): " + q(lag[:, 0]))
print("per expert: median down wake-up - last arrival: " + q(lag[:, 1]))
print("per expert: slowest down wake-up - last arrival: " + q(lag[:, 2]))
hist, edges = np.histogram(dn[dn_work, 3], bins=12)
print("down exit histogram: " + "  ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}us:{hist[i]}" for i in range(len(hist))))
print(f"launch span by the stamps: {dn[dn_work, 3].max():.2f} us")
