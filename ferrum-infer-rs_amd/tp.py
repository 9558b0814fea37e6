"""Tensor-parallel shard math for the dense configs (host side, numpy).

Mirrors the reference's Megatron-style split (crates/ferrum-engine/src/parallel/tensor_parallel.rs:148-340,
tests crates/ferrum-models/tests/tp_sharding_test.rs): q/k/v and gate/up are column-parallel (split N by heads /
intermediate), o_proj and down_proj are row-parallel (split K), norms / embeddings / lm_head are replicated, the KV pool
is sharded by kv head, and the only exchange is an all-reduce(sum, fp16) of [T, H] after o_proj and after down_proj
(ferrum-kernels/src/backend/cuda/tp_decode.rs:363-366).  GPTQ tensors are cut on pack/group boundaries:
column slices must be multiples of 8 (qzeros packs 8 columns per word), row slices multiples of the group size.
"""
import numpy as np


class TransformerParallelMapping:
    """tensor_parallel.rs:209-262."""

    def __init__(self, num_heads, num_kv_heads, head_dim, hidden_dim, intermediate_dim, tp_size):
        for name, v in (("num_heads", num_heads), ("num_kv_heads", num_kv_heads), ("intermediate_dim", intermediate_dim)):
            if v % tp_size != 0:
                raise ValueError(f"{name} {v} must be divisible by tp_size {tp_size}")
        self.tp_size = tp_size
        self.heads_per_rank = num_heads // tp_size
        self.kv_heads_per_rank = num_kv_heads // tp_size
        self.head_dim = head_dim
        self.hidden_per_rank = hidden_dim          # hidden is not sharded
        self.intermediate_per_rank = intermediate_dim // tp_size

    def q_proj_size(self):
        return self.heads_per_rank * self.head_dim

    def k_proj_size(self):
        return self.kv_heads_per_rank * self.head_dim

    def o_proj_in_size(self):
        return self.heads_per_rank * self.head_dim


def shard_range(dim, rank, world):
    """TensorParallelConfig::shard_range."""
    per = dim // world
    return rank * per, (rank + 1) * per


def _cols(qweight, scales, qzeros, lo, hi):
    assert lo % 8 == 0 and hi % 8 == 0, "column shards must be multiples of 8 (qzeros packing)"
    return qweight[:, lo:hi], scales[:, lo:hi], qzeros[:, lo // 8:hi // 8]


def shard_gptq_columns(qweight, scales, qzeros, segments, rank, world):
    """Column-parallel split of a fused projection.  `segments` lists the widths of the fused parts
    (e.g. [q_dim, kv_dim, kv_dim] or [I, I]); each part is split evenly and the rank's pieces re-concatenated."""
    qs, ss, zs, off = [], [], [], 0
    for width in segments:
        lo, hi = shard_range(width, rank, world)
        q, s, z = _cols(qweight, scales, qzeros, off + lo, off + hi)
        qs.append(q); ss.append(s); zs.append(z)
        off += width
    return (np.ascontiguousarray(np.concatenate(qs, axis=1)), np.ascontiguousarray(np.concatenate(ss, axis=1)),
            np.ascontiguousarray(np.concatenate(zs, axis=1)))


def shard_gptq_rows(qweight, scales, qzeros, k, group, rank, world):
    """Row-parallel split (K axis) on group boundaries."""
    lo, hi = shard_range(k, rank, world)
    assert lo % group == 0 and hi % group == 0, "row shards must fall on quant-group boundaries"
    return (np.ascontiguousarray(qweight[lo // 8:hi // 8]), np.ascontiguousarray(scales[lo // group:hi // group]),
            np.ascontiguousarray(qzeros[lo // group:hi // group]))
