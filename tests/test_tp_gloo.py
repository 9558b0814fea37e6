"""world_size-2 `gloo` test of the N>1 path on CPU: the tensor-parallel shard math (ferrum-infer-rs_amd/tp.py,
mirroring tensor_parallel.rs:148-340) + all-reduce(sum) reproduces the unsharded layer, checked with the oracle's
own ops.  Also checks the replica aggregation bench.py uses for the MoE config (max time over ranks, summed tokens)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    from oracle import oracle as O
    from tests import modelgen
    spec = ge.importlib.util.spec_from_file_location("fh_tp", os.path.join(ROOT, "ferrum-infer-rs_amd", "tp.py"))
    tp = ge.importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)

    nq, nkv, hd, H, I, T = 4, 2, 128, 256, 256, 5
    tm = modelgen.TinyModel(False, layers=1, hidden=H, nq=nq, nkv=nkv, hd=hd, inter=I, seed=42)
    L = tm.layers[0]
    rng = np.random.default_rng(7)                      # same on every rank
    x = modelgen.f16r(rng.standard_normal((T, H)))
    mp_ = tp.TransformerParallelMapping(nq, nkv, hd, H, I, world)
    qd, kvd = nq * hd, nkv * hd

    def deq(k, n, qw, sc, qz):
        return O.dequant_gptq(qw, sc, qz, 128, k, n)

    # attention block: column-parallel qkv → local heads → row-parallel o → all-reduce
    k_, n_, qw, sc, qz = L["gptq"]["qkv"]
    sqw, ssc, sqz = tp.shard_gptq_columns(qw, sc, qz, [qd, kvd, kvd], rank, world)
    n_loc = mp_.q_proj_size() + 2 * mp_.k_proj_size()
    qkv = O.gemm(x, deq(H, n_loc, sqw, ssc, sqz), T, n_loc, H)
    q, k, v = O.split_qkv(qkv, mp_.q_proj_size(), mp_.k_proj_size())
    attn = O.cpu_attention(q.reshape(T, mp_.heads_per_rank, hd).transpose(1, 0, 2),
                           k.reshape(T, mp_.kv_heads_per_rank, hd).transpose(1, 0, 2),
                           v.reshape(T, mp_.kv_heads_per_rank, hd).transpose(1, 0, 2), T, T, True, 0,
                           mp_.heads_per_rank, mp_.kv_heads_per_rank, hd).transpose(1, 0, 2).reshape(T, -1)
    k_, n_, qw, sc, qz = L["gptq"]["o"]
    oqw, osc, oqz = tp.shard_gptq_rows(qw, sc, qz, qd, 128, rank, world)
    o_part = O.gemm(attn, deq(mp_.o_proj_in_size(), H, oqw, osc, oqz), T, H, mp_.o_proj_in_size())
    t = torch.from_numpy(o_part.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)            # tp_decode.rs:363
    o_tp = t.numpy()
    # MLP block
    k_, n_, qw, sc, qz = L["gptq"]["gate_up"]
    gqw, gsc, gqz = tp.shard_gptq_columns(qw, sc, qz, [I, I], rank, world)
    Il = mp_.intermediate_per_rank
    gu = O.gemm(x, deq(H, 2 * Il, gqw, gsc, gqz), T, 2 * Il, H)
    act = O.fused_silu_mul_split(gu, Il)
    k_, n_, qw, sc, qz = L["gptq"]["down"]
    dqw, dsc, dqz = tp.shard_gptq_rows(qw, sc, qz, I, 128, rank, world)
    d_part = O.gemm(act, deq(Il, H, dqw, dsc, dqz), T, H, Il)
    t2 = torch.from_numpy(d_part.copy())
    dist.all_reduce(t2, op=dist.ReduceOp.SUM)           # tp_decode.rs:366
    # replica aggregation used by bench.py: MAX of per-rank time, SUM of tokens
    tm_ = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.savez(os.path.join(out_dir, "tp.npz"), o_tp=o_tp, d_tp=t2.numpy(), x=x, tmax=tm_.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_shards_plus_allreduce_equal_unsharded(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from tests import modelgen
    O = oracle
    r = np.load(tmp_path / "tp.npz")
    nq, nkv, hd, H, I, T = 4, 2, 128, 256, 256, 5
    tm = modelgen.TinyModel(False, layers=1, hidden=H, nq=nq, nkv=nkv, hd=hd, inter=I, seed=42)
    L = tm.layers[0]
    x = r["x"]
    qd, kvd = nq * hd, nkv * hd
    _, _, qw, sc, qz = L["gptq"]["qkv"]
    qkv = O.gemm(x, O.dequant_gptq(qw, sc, qz, 128, H, qd + 2 * kvd), T, qd + 2 * kvd, H)
    q, k, v = O.split_qkv(qkv, qd, kvd)
    attn = O.cpu_attention(q.reshape(T, nq, hd).transpose(1, 0, 2), k.reshape(T, nkv, hd).transpose(1, 0, 2),
                           v.reshape(T, nkv, hd).transpose(1, 0, 2), T, T, True, 0, nq, nkv, hd).transpose(1, 0, 2).reshape(T, -1)
    _, _, qw, sc, qz = L["gptq"]["o"]
    o_ref = O.gemm(attn, O.dequant_gptq(qw, sc, qz, 128, qd, H), T, H, qd)
    _, _, qw, sc, qz = L["gptq"]["gate_up"]
    act = O.fused_silu_mul_split(O.gemm(x, O.dequant_gptq(qw, sc, qz, 128, H, 2 * I), T, 2 * I, H), I)
    _, _, qw, sc, qz = L["gptq"]["down"]
    d_ref = O.gemm(act, O.dequant_gptq(qw, sc, qz, 128, I, H), T, H, I)
    # sum order differs (two partial sums): fp32-level agreement
    assert np.max(np.abs(r["o_tp"] - o_ref)) < 1e-4 * max(1.0, np.abs(o_ref).max())
    assert np.max(np.abs(r["d_tp"] - d_ref)) < 1e-4 * max(1.0, np.abs(d_ref).max())
    assert float(r["tmax"][0]) == 1.5


def test_mapping_rejects_indivisible_heads():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    spec = ge.importlib.util.spec_from_file_location("fh_tp2", os.path.join(ROOT, "ferrum-infer-rs_amd", "tp.py"))
    tp = ge.importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    with pytest.raises(ValueError):
        tp.TransformerParallelMapping(32, 4, 128, 2048, 768, 8)         # nkv=4 not divisible by 8
    m = tp.TransformerParallelMapping(64, 8, 128, 8192, 28672, 8)       # Llama-70B TP8
    assert (m.heads_per_rank, m.kv_heads_per_rank, m.intermediate_per_rank) == (8, 1, 3584)
    assert m.q_proj_size() + 2 * m.k_proj_size() == 1280 and m.o_proj_in_size() == 1024
