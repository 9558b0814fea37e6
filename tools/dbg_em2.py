"""Development: the merged gate_up → down launch against the two expert-major launches at a given shape; prints where the
outputs differ (which buffer, NaN or value, rows / columns) for poisoned and zeroed hand-off buffers."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.load_library()
from oracle import oracle as O
O.build()
B = pkg.HipBackend
ctx = B.new_context()


def f16r(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


SYM = os.environ.get('DBG_ASYM', '0') != '1'


def run(tokens, E, K, H, I, fill):
    rng = np.random.default_rng(tokens * E + H)
    gu = [O.make_synthetic_gptq(H, 2 * I, 128, 100 + e, symmetric=SYM) for e in range(E)]
    dn = [O.make_synthetic_gptq(I, H, 128, 200 + e, symmetric=SYM) for e in range(E)]
    gu = [(q, f16r(s / (0.28 * np.sqrt(H))), z) for q, s, z in gu]
    dn = [(q, f16r(s / (0.28 * np.sqrt(I))), z) for q, s, z in dn]
    x = f16r(rng.standard_normal((tokens, H)))
    logits = rng.standard_normal((tokens, E)).astype(np.float32)
    rid, rw = O.route_topk(logits, E, K, True)
    P = tokens * K
    ids_d = torch.from_numpy(rid.astype(np.int32).reshape(-1).copy()).cuda()
    xd = torch.from_numpy(x).cuda().half().contiguous()
    down_stack = B.load_gptq_stacked([q for q, _, _ in dn], [s for _, s, _ in dn], [z for _, _, z in dn], None, 4, 128, I, H)
    stack = B.load_gptq_stacked([q for q, _, _ in gu], [s for _, s, _ in gu], [z for _, _, z in gu], None, 4, 128, H, 2 * I, fuse_gate_up=True)
    act = torch.zeros(P, I, dtype=torch.float16, device="cuda")
    down = torch.zeros(P, H, dtype=torch.float16, device="cuda")
    stack.gemm_phase_expert_major(ctx, xd, ids_d, act, P, E, K, fused_silu_mul=True)
    down_stack.gemm_phase_expert_major(ctx, act, ids_d, down, P, E, 1)
    ctx.sync()
    cnt = np.bincount(rid.reshape(-1), minlength=E)
    print(f"== tokens={tokens} E={E} K={K} H={H} I={I} fill={fill} pairs/expert min {cnt.min()} max {cnt.max()}")
    for rep in range(3):
        act4 = torch.full((P, I), fill, dtype=torch.float16, device="cuda")
        down4 = torch.full((P, H), fill, dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        stack.gemm_phase_expert_major_pair(ctx, down_stack, xd, ids_d, act4, down4, P, E, K)
        ctx.sync()
        for name, a, b in (("act", act, act4), ("down", down, down4)):
            bad = ~((a == b) | (torch.isnan(a) & torch.isnan(b)))
            nb = int(bad.sum())
            if nb:
                rows = torch.nonzero(bad.any(1)).flatten().tolist()
                cols = torch.nonzero(bad.any(0)).flatten().tolist()
                nan = int(torch.isnan(b).sum())
                d = (a.float() - b.float())[bad]
                print(f"  rep {rep} {name}: {nb} differ, {nan} NaN in pair output, rows {rows[:12]}… ({len(rows)}), cols {cols[:8]}… ({len(cols)}), max|Δ| {float(d.abs().max()) if nan == 0 else 'nan'}")
                r0 = rows[0]
                print(f"     row {r0} expert {int(ids_d[r0])}: ref {a[r0, cols[:4]].tolist()} got {b[r0, cols[:4]].tolist()}")
            else:
                print(f"  rep {rep} {name}: identical")
        print("  timeouts", stack.pair_timeouts(ctx))


for shape in ((32, 16, 4, 512, 256), (32, 128, 8, 2048, 768), (64, 8, 2, 256, 256), (9, 8, 2, 256, 512)):
    for fill in (float("nan"), 0.0):
        run(*shape, fill)
