import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import oracle as O
pkg = ge.load_package(); B = pkg.HipBackend; ctx = B.new_context()
f16r = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
tokens, E, K, H, I = 1, 8, 2, 256, 128
rng = np.random.default_rng(0)
gu = [O.make_synthetic_gptq(H, 2 * I, 128, 100 + e, symmetric=True) for e in range(E)]
gu = [(q, f16r(s / (0.28 * np.sqrt(H))), z) for q, s, z in gu]
x = f16r(rng.standard_normal((tokens, H)))
rid = np.array([[3, 5]], np.uint32)
P = tokens * K; sorted_max = P + E * 16
ids_d = torch.from_numpy(rid.astype(np.int32).reshape(-1).copy()).cuda()
sd = torch.empty(sorted_max, dtype=torch.int32, device="cuda"); bd = torch.empty(sorted_max // 16 + 1, dtype=torch.int32, device="cuda"); td = torch.zeros(1, dtype=torch.int32, device="cuda")
B.moe_align_block_size_pair_ids(ctx, ids_d, sd, bd, td, P, E, 16, sorted_max)
stack = B.load_gptq_stacked([q for q, _, _ in gu], [s for _, s, _ in gu], [z for _, _, z in gu], None, 4, 128, H, 2 * I)
gup = torch.zeros(P, 2 * I, dtype=torch.float16, device="cuda")
xd = torch.from_numpy(x).cuda().half()
stack.gemm_phase_vllm(ctx, xd, sd, bd, td, gup, P, 16, K, sorted_max // 16)
ctx.sync()
g = gup.float().cpu().numpy()
print("sorted", sd.cpu().numpy()[:40], "blocks", bd.cpu().numpy()[:4], td.item())
for p in range(P):
    w = O.dequant_gptq(*gu[rid[0, p]], 128, H, 2 * I)
    ref = O.gemm(x, w, 1, 2 * I, H)[0]
    print(p, "nan count", np.isnan(g[p]).sum(), "max abs diff", np.nanmax(np.abs(g[p] - ref)), "ref max", np.abs(ref).max())
    bad = np.where(np.isnan(g[p]))[0]
    print("  bad cols", bad[:40])
