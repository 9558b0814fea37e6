"""Development: the one-launch decode chain (chain.hip) against the seven-launch layer on a few layers of a synthetic
Qwen3-30B-A3B: same prompts, same steps, logits and ids compared exactly; eager and hipGraph decode loops.
usage: dbg_chain.py [layers] [c ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
lib = pkg.load_library()
LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CS = [int(x) for x in sys.argv[2:]] or [32, 8, 1, 20]
PL = 64


def knob(**kw):
    for k, v in kw.items():
        os.environ["FERRUM_HIP_" + k] = str(v)
    lib.ferrum_hip_debug_reload_knobs()


def run(c, chain, em2, feed=None):
    knob(DECODE_CHAIN=chain, MOE_EM2=em2)
    cfg = dict(bench.QWEN3_30B_A3B)
    cfg["vocab"] = 4096
    model = bench.build_model(pkg, cfg, c, PL + 64, c * PL, 1234, layers=LAYERS)
    rng = np.random.default_rng(9271)
    prompts = [rng.integers(256, 4096, size=PL).astype(np.uint32) for _ in range(c)]
    toks = bench.prefill(model, prompts, 0, c * PL)
    out = []
    pos = PL
    fed = [np.array(toks).copy()]
    for st in range(4):                      # eager single steps with logits; `feed`: the reference run's tokens (teacher forcing)
        inp = feed[st] if feed is not None else toks
        toks, lg = model.unified_forward([(i, [int(inp[i])], pos, True) for i in range(c)], greedy=True, want_logits=True)
        out.append((np.array(toks).copy(), lg.copy()))
        fed.append(np.array(toks).copy())
        pos += 1
    inp = feed[4] if feed is not None else toks
    ids = model.decode_steps(list(range(c)), np.array(inp, np.uint32), 6)     # hipGraph loop
    out.append((np.array(ids).copy(), None))
    del model
    return out, fed


for c in CS:
    ref, fed = run(c, 0, 0)
    for chain, em2 in ((1, 0), (1, 1)):
        got, _ = run(c, chain, em2, fed)
        ok = True
        for s, ((ti, li), (tj, lj)) in enumerate(zip(ref, got)):
            same_ids = np.array_equal(ti, tj)
            if li is not None:
                d = np.abs(li - lj)
                exact = np.array_equal(li, lj)
                # an id may only differ where the reference's own top-2 margin is within twice the logit error
                flips = np.nonzero(ti != tj)[0]
                srt = np.sort(li, axis=1)
                bad = [int(r) for r in flips if srt[r, -1] - srt[r, -2] > 2 * d[r].max() + 1e-6]
                print(f"c={c} chain={chain} em2={em2} step {s}: id flips {len(flips)} (unexplained {len(bad)})  logits {'bit-identical' if exact else 'max|d| %.3e (max|l| %.2f) nan %d' % (float(np.nanmax(d)), float(np.abs(li).max()), int(np.isnan(lj).sum()))}")
                ok &= not bad and float(np.nanmax(d)) < 0.02 * float(np.abs(li).max()) and not np.isnan(lj).any()
            else:
                print(f"c={c} chain={chain} em2={em2} graph loop: ids {'==' if same_ids else '!= (%d of %d)' % (int((ti != tj).sum()), ti.size)}")
        print(f"c={c} chain={chain} em2={em2}: {'OK' if ok else 'MISMATCH'}")
