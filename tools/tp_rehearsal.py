#!/usr/bin/env python3
"""Multi-process tensor-parallel rehearsal on ONE GPU (development / test aid; no xGMI involved).

`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/tp_rehearsal.py`
starts N ranks that all use device 0 and rendezvous over gloo.  RCCL refuses two ranks on one device, so the ranks talk
through the hand-written one-shot peer all-reduce over hipIpc-imported buffers — the same code path a multi-GPU group
takes with `--tp-transport oneshot`, minus the fabric.  Checks, and prints one JSON line from rank 0:
  1. ferrum_hip_all_reduce_f16 on 4 … 1 Mi fp16 elements, eagerly and from a captured + replayed hipGraph, against the
     rank-ordered fp32 sum of the gathered inputs (bit-exact), 40 calls back to back (buffer parity / epoch logic);
  2. the bench's own tensor-parallel decode case (per-rank synthetic shards of a dense model, per-rank hipGraph with the
     all-reduces inside): every rank must sample the same ids, no rank may count a one-shot timeout;
  3. the same for the MoE model as one expert-parallel group (experts sharded over the ranks).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)                                   # every rank on the one GPU
    dist.init_process_group("gloo")
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    pkg.load_library()
    B = pkg.HipBackend
    out = {"world": world}

    # 1. the collective itself
    comm = pkg.Comm.bare(world, rank)
    hs = [None] * world
    dist.all_gather_object(hs, comm.oneshot_export(4 << 20))
    comm.oneshot_attach(hs)
    stream = torch.cuda.Stream()
    rng = np.random.default_rng(100 + rank)
    worst, calls = 0.0, 0
    with torch.cuda.stream(stream):
        ctx = B.new_context()
        for it in range(40):
            n = [4, 4096, 262144, 1 << 20, 52][it % 5] if it % 5 != 4 else 1000
            x = (rng.standard_normal(n) * (1 + rank)).astype(np.float16)
            gathered = [None] * world
            dist.all_gather_object(gathered, x)
            ref = np.zeros(n, np.float32)
            for r in range(world):                               # rank order, fp32 accumulate, one rounding
                ref += gathered[r].astype(np.float32)
            ref = ref.astype(np.float16)
            t = torch.from_numpy(x).cuda()
            if it % 2 == 0:
                comm.all_reduce(t, n, ctx.stream)
            else:                                                # the same call recorded in a hipGraph and replayed once
                ctx.sync()
                B.begin_graph_capture(ctx)
                comm.all_reduce(t, n, ctx.stream)
                g = B.end_graph_capture(ctx)
                B.replay_graph(ctx, g)
                ctx.sync()
                B.reset_graph(ctx, g)
            ctx.sync()
            got = t.cpu().numpy()
            assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (rank, it, n, float(np.max(np.abs(got.astype(np.float32) - ref.astype(np.float32)))))
            calls += 1
        # 1b. the all-reduce folded into its consumer (residual add + norm): one launch ≡ all_reduce → fused_add_rms_norm, bit for bit
        fused_calls = 0
        for it, (rows, dim) in enumerate([(1, 1024), (20, 4096), (32, 8192), (64, 5376), (7, 2048), (32, 4096)]):
            x = torch.from_numpy((rng.standard_normal((rows, dim)) * (1 + rank)).astype(np.float16)).cuda()
            g = np.random.default_rng(7000 + it)                  # residual and weights: the same on every rank
            res0 = torch.from_numpy(g.standard_normal((rows, dim)).astype(np.float16)).cuda()
            w = torch.from_numpy((1 + 0.1 * g.standard_normal(dim)).astype(np.float16)).cuda()
            res_a, out_a = res0.clone(), torch.empty_like(res0)
            took = comm.all_reduce_add_rms_norm(x, res_a, w, 1e-6, out_a, rows, dim, ctx.stream)
            assert took, (rows, dim)
            xr, res_b, out_b = x.clone(), res0.clone(), torch.empty_like(res0)
            comm.all_reduce(xr, rows * dim, ctx.stream)
            B.fused_add_rms_norm(ctx, res_b, xr, w, 1e-6, out_b, rows, dim)
            ctx.sync()
            assert torch.equal(res_a, res_b) and torch.equal(out_a, out_b), (rank, rows, dim)
            fused_calls += 1
            calls += 2
    st = comm.oneshot_status()
    assert st["timeouts"] == 0 and st["epoch"] == calls, st
    out["all_reduce"] = {"calls": calls, "epoch": st["epoch"], "timeouts": st["timeouts"], "bit_exact": True,
                         "fused_norm_calls": fused_calls, "fused_norm_bit_exact": True}
    comm.destroy()
    dist.barrier()

    # 2. the runner's tensor-parallel decode across processes (graph + one-shot all-reduce inside)
    res = bench.tp_decode_case(pkg, torch, dist, os.environ.get("FERRUM_REHEARSAL_MODEL", "llama31-8b"), world, rank,
                               c=int(os.environ.get("FERRUM_REHEARSAL_C", "20")), PL=48, steps=8, warm=2, chunk=96,
                               try_oneshot=False, transport="oneshot", layers=int(os.environ.get("FERRUM_REHEARSAL_LAYERS", "3")))
    out["tp_decode"] = res
    lib = pkg.load_library()
    import ctypes as C
    lib.ferrum_hip_debug_form_name.restype = C.c_char_p
    nforms = lib.ferrum_hip_debug_form_count()
    arr = (C.c_uint64 * nforms)()
    lib.ferrum_hip_debug_form_hits(arr, nforms)
    out["forms"] = {lib.ferrum_hip_debug_form_name(i).decode(): int(arr[i]) for i in range(nforms) if arr[i]}
    # 3. the MoE model as one expert-parallel group (experts sharded, partial MoE outputs all-reduced)
    if 128 % world == 0:
        out["ep_decode"] = bench.tp_decode_case(pkg, torch, dist, "qwen3-30b-a3b", world, rank, c=16, PL=48, steps=8, warm=2, chunk=96,
                                                try_oneshot=False, transport="oneshot", layers=3)
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
