#!/usr/bin/env python3
"""Sweep of w4_gemm_ldsk_kernel forms (FERRUM_HIP_W4_LDSK = nw·100 + kw·10 + d, FERRUM_HIP_W4_LDSA_S = slabs) against the
default dense path at decode shapes: µs per launch (events over rotating weight copies) and max |Δ| vs the default.
Development aid; needs the library built with `make -C ferrum-infer-rs_amd/csrc EXPERIMENTS=1` (the forms are not in the product build)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
from oracle import oracle as O
lib = pkg.backend._lib


def knob(**kw):
    for k, v in kw.items():
        if v is None: os.environ.pop("FERRUM_HIP_" + k, None)
        else: os.environ["FERRUM_HIP_" + k] = str(v)
    lib.ferrum_hip_debug_reload_knobs()


TRACE = "--trace" in sys.argv      # under rocprofv3 --kernel-trace: few launches per form, durations come from the trace


def timeit(fn, reps=40):
    if TRACE: reps = 30
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def lin(k, n, seed):
    qw, sc, qz = O.make_synthetic_gptq(k, n, 128, seed, symmetric=True)
    return pkg.GptqLinear.from_raw(qw, sc.astype(np.float16).astype(np.float32), qz, None, None, 4, 128, k, n)


SHAPES = {"o": (4096, 4096), "qkv": (4096, 6144), "down": (14336, 4096), "gate_up": (4096, 28672),
          "q3o": (4096, 2048), "q3qkv": (2048, 5120), "l70o": (8192, 8192), "l70down": (28672, 8192)}
FORMS2 = [1412, 1422, 1423, 1222, 1242, 1812] if "--ares" in sys.argv else [414, 422, 424, 222, 242, 244, 182]
FORMS4 = [1412, 1422, 1222] if "--ares" in sys.argv else [412, 414, 222]
names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["o", "qkv", "down", "gate_up", "q3o", "q3qkv"]
ms = [int(a[3:]) for a in sys.argv[1:] if a.startswith("-m=")] or [32]
for name in names:
    k, n = SHAPES[name]
    copies = max(2, min(6, int(300e6 // (k * n // 2))))
    lins = [lin(k, n, 10 + i) for i in range(copies)]
    wbytes = k * n // 2 + (k // 128) * n * 2
    G = k // 128
    for m in ms:
        xin = (torch.randn(m, k, device="cuda") * 0.5).half(); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
        knob(W4_LDSK=None, W4_LDSA_S=None)
        us0 = timeit(lambda i: lins[i % copies].forward(ctx, xin, out, m))
        lins[0].forward(ctx, xin, out, m); torch.cuda.synchronize(); ref = out.float().clone()
        print(f"{name:8s} K={k:5d} N={n:5d} m={m:2d} default          : {us0:7.2f} us  {wbytes / us0 / 1e3:7.1f} GB/s", flush=True)
        best = None
        pick = [a[7:] for a in sys.argv[1:] if a.startswith("-forms=")]          # -forms=1422:2,1412:4
        pairs = [tuple(int(v) for v in fs.split(":")) for fs in pick[0].split(",")] if pick else None
        for form in (sorted({f for f, _ in pairs}) if pairs else (FORMS2 if m <= 32 else FORMS4)):
            kw = (form // 10) % 10
            for S in ([s_ for f, s_ in pairs if f == form] if pairs else (1, 2, 4, 7, 8, 14, 16)):
                if G // (S * kw) < 1 or G % S: continue
                if form >= 1000 and -(-G // S) * (2 if m <= 32 else 4) * 4096 > 160 * 1024: continue
                knob(W4_LDSK=form, W4_LDSA_S=S)
                try:
                    us = timeit(lambda i: lins[i % copies].forward(ctx, xin, out, m))
                except Exception as e:
                    print("   ", form, S, "failed:", str(e)[:100]); continue
                lins[0].forward(ctx, xin, out, m); torch.cuda.synchronize()
                err = (out.float() - ref).abs().max().item()
                tag = ""
                if best is None or us < best[0]: best = (us, form, S)
                print(f"    form {form} S={S:2d}: {us:7.2f} us  {wbytes / us / 1e3:7.1f} GB/s  maxdiff {err:.4f}", flush=True)
        print(f"  -> best {best[1]} S={best[2]} {best[0]:.2f} us vs default {us0:.2f}", flush=True)
    del lins
knob(W4_LDSK=None, W4_LDSA_S=None)
