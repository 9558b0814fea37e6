"""Host mirror of the reference's `ModelExecutor` unified-decode contract over the C++ runner.

`HipModel` wraps `ferrum_hip_model_*` (include/ferrum_hip.h): weight hand-over as host slices
(`WeightLoader` → `B::load_gptq`, ferrum-quantization/src/loader.rs:21), `reserve_kv_slots` /
`unified_decode` / `release` (ferrum-interfaces/src/model_executor.rs:456-651) and the
steady-state decode loop.  All compute happens in libferrum_hip.so.
"""
import ctypes as C

import numpy as np

from .backend import _check, load_library


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_layers", "hidden", "num_heads", "num_kv_heads", "head_dim", "intermediate", "vocab", "max_seq_len",
        "has_qk_norm", "activation", "num_experts", "top_k", "expert_inter", "norm_topk_prob", "rope_scaling_kind",
        "sliding_window", "group_size", "kv_num_blocks", "max_seqs", "max_tokens")] + [
        ("rms_eps", C.c_float), ("_pad", C.c_float), ("rope_theta", C.c_double), ("rope_p0", C.c_double),
        ("rope_p1", C.c_double), ("rope_p2", C.c_double), ("rope_p3", C.c_double), ("tp_rank", C.c_int32),
        ("tp_world", C.c_int32), ("sliding_window_pattern", C.c_int32), ("sandwich_norms", C.c_int32),
        ("embed_scale", C.c_float), ("_pad2", C.c_float), ("rope_local_theta", C.c_double), ("expert_parallel", C.c_int32),
        ("vocab_parallel", C.c_int32)]


class BatchItem(C.Structure):
    """UnifiedBatchItem (model_executor.rs:354-386)."""
    _fields_ = [("seq_id", C.c_uint64), ("q_tokens", C.POINTER(C.c_uint32)), ("num_q_tokens", C.c_int32),
                ("pos_offset", C.c_int32), ("is_final_chunk", C.c_int32), ("_pad", C.c_int32)]


class GreedyOptions(C.Structure):
    """LogitsReturnPolicy::GreedyArgmax { token_mask, repetition_penalty } for one batch (model_executor.rs:109-150)."""
    _fields_ = [("valid_token_mask", C.POINTER(C.c_uint8)), ("mask_len", C.c_int32), ("_pad", C.c_int32),
                ("penalty_row_offsets", C.POINTER(C.c_uint32)), ("penalty_token_ids", C.POINTER(C.c_uint32)),
                ("penalties", C.POINTER(C.c_float))]


class KvSlotRequest(C.Structure):
    _fields_ = [("seq_id", C.c_uint64), ("target_len", C.c_int32), ("_pad", C.c_int32)]


class KvSlotReservation(C.Structure):
    _fields_ = [("block_size", C.c_int32), ("total_blocks", C.c_int32), ("free_blocks_before", C.c_int32),
                ("free_blocks_after", C.c_int32)]


_DEFAULTS = dict(max_seq_len=512, has_qk_norm=0, activation=0, num_experts=0, top_k=0, expert_inter=0,
                 norm_topk_prob=1, rope_scaling_kind=0, sliding_window=0, group_size=128, kv_num_blocks=256,
                 max_seqs=32, max_tokens=512, rms_eps=1e-6, rope_theta=1e6, intermediate=0, tp_rank=0, tp_world=1)

GLOBAL = {"embed": 0, "lm_head": 1, "final_norm": 2}
LAYER_DENSE = {"input_ln": 0, "post_ln": 1, "q_norm": 2, "k_norm": 3, "router": 4, "post_attn_ln": 5, "post_ffn_ln": 6,
               "qkv_bias": 7}
GPTQ = {"qkv": 0, "o": 1, "gate_up": 2, "down": 3, "expert_gate_up": 4, "expert_down": 5}


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Checkpoint:
    """Checkpoint directory reader over `ferrum_hip_checkpoint_*` (the C++ mirror of NativeSafetensorsLoader,
    ferrum-quantization/src/native_safetensors.rs:130-330,887-1000, and of the config.json mapping)."""

    def __init__(self, model_dir):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.ferrum_hip_checkpoint_open(C.byref(self.h), str(model_dir).encode()), "checkpoint_open")

    def close(self):
        if self.h:
            self.lib.ferrum_hip_checkpoint_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_tensors(self):
        return int(self.lib.ferrum_hip_checkpoint_num_tensors(self.h))

    def tensor_info(self, name):
        dt, nd = C.c_int(), C.c_int()
        shape = (C.c_int64 * 4)()
        _check(self.lib.ferrum_hip_checkpoint_tensor_info(self.h, name.encode(), C.byref(dt), C.byref(nd), shape), "tensor_info")
        return ("F32", "F16", "BF16", "I32", "I64", "other")[dt.value], tuple(shape[:nd.value])

    def read_f32(self, name):
        _, shape = self.tensor_info(name)
        out = np.empty(shape, np.float32)
        _check(self.lib.ferrum_hip_checkpoint_read_f32(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)),
                                                       C.c_size_t(out.size)), "read_f32")
        return out

    def read_i32(self, name):
        _, shape = self.tensor_info(name)
        out = np.empty(shape, np.int32)
        _check(self.lib.ferrum_hip_checkpoint_read_i32(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       C.c_size_t(out.size)), "read_i32")
        return out

    def quant_config(self):
        v = [C.c_int() for _ in range(5)]
        _check(self.lib.ferrum_hip_checkpoint_quant_config(self.h, *[C.byref(x) for x in v]), "quant_config")
        return dict(zip(("is_gptq", "bits", "group_size", "desc_act", "sym"), (x.value for x in v)))

    def read_gptq_fused(self, parts):
        arr = (C.c_char_p * len(parts))(*[p.encode() for p in parts])
        k, n, hg = C.c_int(), C.c_int(), C.c_int()
        fn = self.lib.ferrum_hip_checkpoint_read_gptq_fused
        _check(fn(self.h, arr, len(parts), None, None, None, None, C.byref(k), C.byref(n), C.byref(hg)), "read_gptq_fused")
        g = self.quant_config()["group_size"]
        K, N = k.value, n.value
        qw, sc = np.empty((K // 8, N), np.int32), np.empty((K // g, N), np.float32)
        qz, gi = np.empty((K // g, N // 8), np.int32), np.empty(K, np.int32)
        p = C.POINTER(C.c_int32)
        _check(fn(self.h, arr, len(parts), qw.ctypes.data_as(p), sc.ctypes.data_as(C.POINTER(C.c_float)), qz.ctypes.data_as(p),
                  gi.ctypes.data_as(p), None, None, None), "read_gptq_fused")
        return qw, sc, qz, (gi if hg.value else None), K, N

    def model_config(self, max_seq_len_cap=0):
        cfg = ModelConfig()
        arch = C.create_string_buffer(128)
        tied = C.c_int()
        _check(self.lib.ferrum_hip_checkpoint_model_config(self.h, max_seq_len_cap, C.byref(cfg), arch, C.c_size_t(128),
                                                           C.byref(tied)), "checkpoint_model_config")
        d = {f: getattr(cfg, f) for f, _ in ModelConfig._fields_ if not f.startswith("_pad")}
        return d, arch.value.decode(), bool(tied.value)


class Comm:
    """`BackendCollective` (ferrum-kernels/src/backend/capabilities.rs:84-109) over `ferrum_hip_comm_*`: one rank of a
    tensor-parallel group.  Transports: RCCL (`Comm.rccl`), a one-shot peer reduce between ranks of one process
    (`Comm.local_group`) or between processes through hipIpc handles (`oneshot_export` / `oneshot_attach`)."""

    def __init__(self, handle, owner=None):
        self.lib = load_library()
        self.h = handle
        self._owner = owner

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        _check(lib.ferrum_hip_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    @classmethod
    def rccl(cls, world, rank, unique_id):
        lib = load_library()
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib.ferrum_hip_comm_create_rccl(C.byref(h), world, rank, buf), "comm_create_rccl")
        return cls(h)

    @classmethod
    def bare(cls, world, rank):
        """A rank with no transport yet: attach the one-shot buffers with oneshot_export / oneshot_attach."""
        lib = load_library()
        h = C.c_void_p()
        _check(lib.ferrum_hip_comm_create_bare(C.byref(h), world, rank), "comm_create_bare")
        return cls(h)

    @classmethod
    def local_group(cls, world, max_message_bytes=1 << 20):
        lib = load_library()
        arr = (C.c_void_p * world)()
        _check(lib.ferrum_hip_comm_create_local_group(arr, world, C.c_size_t(max_message_bytes), None), "comm_create_local_group")
        return [cls(C.c_void_p(arr[r])) for r in range(world)]

    def oneshot_export(self, max_message_bytes=1 << 20):
        buf = (C.c_uint8 * 64)()
        _check(self.lib.ferrum_hip_comm_oneshot_export(self.h, C.c_size_t(max_message_bytes), buf), "comm_oneshot_export")
        return bytes(buf)

    def oneshot_attach(self, handles):
        blob = b"".join(handles)
        arr = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(self.lib.ferrum_hip_comm_oneshot_attach(self.h, arr, len(handles)), "comm_oneshot_attach")

    def oneshot_status(self):
        e, t = C.c_uint(), C.c_uint()
        _check(self.lib.ferrum_hip_comm_oneshot_status(self.h, C.byref(e), C.byref(t)), "comm_oneshot_status")
        return {"epoch": e.value, "timeouts": t.value}

    def world_size(self):
        return self.lib.ferrum_hip_comm_world_size(self.h)

    def rank(self):
        return self.lib.ferrum_hip_comm_rank(self.h)

    def all_reduce(self, buf, count, stream):
        """BackendCollective::all_reduce, ReduceOp::Sum over fp16, in place on `stream` (a torch CUDA tensor or raw pointer)."""
        ptr = C.c_void_p(buf.data_ptr()) if hasattr(buf, "data_ptr") else buf
        _check(self.lib.ferrum_hip_all_reduce_f16(self.h, ptr, C.c_size_t(count), stream), "all_reduce")

    def all_reduce_add_rms_norm(self, x, residual, w, eps, norm_out, rows, dim, stream):
        """residual += all_reduce(x); norm_out = rms_norm(residual)·w as ONE launch (one-shot transport); False when not taken."""
        fused = C.c_int(0)
        p = lambda t: C.c_void_p(t.data_ptr())
        _check(self.lib.ferrum_hip_all_reduce_add_rms_norm_f16(self.h, p(x), p(residual), p(w), C.c_float(eps), p(norm_out), rows, dim,
                                                               C.byref(fused), stream), "all_reduce_add_rms_norm")
        return bool(fused.value)

    def destroy(self):
        if self.h:
            self.lib.ferrum_hip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipModel:
    @classmethod
    def from_checkpoint(cls, model_dir, kv_num_blocks, max_seqs, max_tokens, max_seq_len=0):
        """Create → load every tensor with the reference's names and fusions → finalize."""
        ck = Checkpoint(model_dir)
        d, _, _ = ck.model_config(max_seq_len)
        d.update(kv_num_blocks=kv_num_blocks, max_seqs=max_seqs, max_tokens=max_tokens)
        m = cls(**d)
        _check(m.lib.ferrum_hip_model_load_checkpoint(m.h, ck.h), "model_load_checkpoint")
        m.finalize()
        ck.close()
        return m

    def __init__(self, **kw):
        self.lib = load_library()
        self.cfg = ModelConfig()
        d = dict(_DEFAULTS)
        d.update(kw)
        for k, v in d.items():
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        _check(self.lib.ferrum_hip_model_create(C.byref(self.h), C.byref(self.cfg)), "model_create")

    def __del__(self):
        try:
            if self.h:
                self.lib.ferrum_hip_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ── weights ──────────────────────────────────────────────────────────────
    def set_global(self, name, data):
        d = _f32(data)
        _check(self.lib.ferrum_hip_model_set_global_f32(self.h, GLOBAL[name], d.ctypes.data_as(C.POINTER(C.c_float))),
               f"set_global({name})")

    def set_layer_dense(self, layer, name, data):
        d = _f32(data)
        _check(self.lib.ferrum_hip_model_set_layer_dense_f32(self.h, layer, LAYER_DENSE[name],
                                                             d.ctypes.data_as(C.POINTER(C.c_float))), f"set_layer_dense({name})")

    def set_gptq(self, layer, name, qweight, scales, qzeros, k, n, expert=0, g_idx=None):
        qw, sc, qz = _i32(qweight), _f32(scales), _i32(qzeros)
        gi = None if g_idx is None else _i32(g_idx)
        p = C.POINTER(C.c_int32)
        _check(self.lib.ferrum_hip_model_set_gptq(self.h, layer, GPTQ[name], expert, qw.ctypes.data_as(p),
                                                  sc.ctypes.data_as(C.POINTER(C.c_float)), qz.ctypes.data_as(p),
                                                  None if gi is None else gi.ctypes.data_as(p), k, n), f"set_gptq({name})")

    def set_dense(self, layer, name, weight, k, n):
        """Unquantised projection (DenseLinear): weight [n, k] f32; name in qkv / o / gate_up / down."""
        w = _f32(np.asarray(weight).reshape(n, k))
        _check(self.lib.ferrum_hip_model_set_dense_f32(self.h, layer, GPTQ[name], w.ctypes.data_as(C.POINTER(C.c_float)), k, n),
               f"set_dense({name})")

    def init_synthetic(self, seed):
        _check(self.lib.ferrum_hip_model_init_synthetic(self.h, C.c_uint64(seed)), "init_synthetic")

    def finalize(self):
        _check(self.lib.ferrum_hip_model_finalize(self.h), "finalize")

    # ── block-level prefix cache (models/qwen3_moe/prefix_cache.rs) ───────────
    def prefix_cache_acquire(self, seq_id, tokens):
        """After reserve_kv_slots: splice cached prefix blocks in; returns the number of prompt tokens already cached."""
        t = np.ascontiguousarray(tokens, np.uint32)
        cached = C.c_int()
        _check(self.lib.ferrum_hip_model_prefix_cache_acquire(self.h, C.c_uint64(seq_id), t.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                              len(t), C.byref(cached)), "prefix_cache_acquire")
        return cached.value

    def prefix_cache_register(self, seq_id, all_tokens, prior_cached_tokens):
        t = np.ascontiguousarray(all_tokens, np.uint32)
        _check(self.lib.ferrum_hip_model_prefix_cache_register(self.h, C.c_uint64(seq_id), t.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                               len(t), prior_cached_tokens), "prefix_cache_register")

    def prefix_cache_stats(self):
        v = [C.c_uint64() for _ in range(4)]
        _check(self.lib.ferrum_hip_model_prefix_cache_stats(self.h, *[C.byref(x) for x in v]), "prefix_cache_stats")
        return dict(zip(("hits", "misses", "saved_prefill_tokens", "entries"), (int(x.value) for x in v)))

    # ── KV admission ─────────────────────────────────────────────────────────
    def reserve_kv_slots(self, requests):
        """requests: [(seq_id, target_len)] → KvSlotReservation dict; raises when the pool is exhausted."""
        arr = (KvSlotRequest * max(len(requests), 1))(*[KvSlotRequest(s, t, 0) for s, t in requests])
        out = KvSlotReservation()
        _check(self.lib.ferrum_hip_model_reserve_kv_slots(self.h, arr, len(requests), C.byref(out)), "reserve_kv_slots")
        return {f: getattr(out, f) for f, _ in KvSlotReservation._fields_}

    def kv_slot_capacity_snapshot(self):
        out = KvSlotReservation()
        _check(self.lib.ferrum_hip_model_kv_capacity_snapshot(self.h, C.byref(out)), "kv_capacity_snapshot")
        return {"block_size": out.block_size, "total_blocks": out.total_blocks, "free_blocks": out.free_blocks_after}

    def release(self, seq_id):
        _check(self.lib.ferrum_hip_model_release(self.h, C.c_uint64(seq_id)), "release")

    def block_table(self, seq_id):
        cap = (self.cfg.max_seq_len + 15) // 16
        arr = (C.c_uint32 * cap)()
        nb, kl = C.c_int(), C.c_int()
        _check(self.lib.ferrum_hip_model_block_table(self.h, C.c_uint64(seq_id), arr, cap, C.byref(nb), C.byref(kl)), "block_table")
        return list(arr[:nb.value]), kl.value

    def read_kv(self, seq_id, layer, is_v):
        _, kv_len = self.block_table(seq_id)
        out = np.zeros((kv_len, self.cfg.num_kv_heads, self.cfg.head_dim), np.float32)
        _check(self.lib.ferrum_hip_model_read_kv_f32(self.h, C.c_uint64(seq_id), layer, int(is_v),
                                                     out.ctypes.data_as(C.POINTER(C.c_float))), "read_kv")
        return out

    # ── forward ──────────────────────────────────────────────────────────────
    def unified_forward(self, items, greedy=True, want_logits=False, token_mask=None, repetition_penalties=None):
        """items: [(seq_id, tokens, pos_offset, is_final_chunk)].  Returns (tokens|None, logits|None) for the
        final-chunk items in order.  token_mask: uint8 [mask_len] shared by all rows; repetition_penalties: per sampled
        row (penalty, [de-duplicated token ids]) — the GreedyArgmax policy of the reference, applied on the device."""
        keep = [np.ascontiguousarray(t, dtype=np.uint32) for _, t, _, _ in items]
        arr = (BatchItem * len(items))()
        n_final = 0
        for i, (sid, _, pos, fin) in enumerate(items):
            arr[i] = BatchItem(sid, keep[i].ctypes.data_as(C.POINTER(C.c_uint32)), len(keep[i]), pos, int(fin), 0)
            n_final += int(bool(fin))
        toks = np.zeros(max(n_final, 1), np.uint32)
        logits = np.zeros((max(n_final, 1), self.local_vocab()[1]), np.float32) if want_logits else None   # this rank's slice when vocab_parallel
        opts, hold = None, []
        if token_mask is not None or repetition_penalties is not None:
            opts = GreedyOptions()
            if token_mask is not None:
                mk = np.ascontiguousarray(token_mask, np.uint8)
                hold.append(mk)
                opts.valid_token_mask, opts.mask_len = mk.ctypes.data_as(C.POINTER(C.c_uint8)), len(mk)
            if repetition_penalties is not None:
                assert len(repetition_penalties) == n_final
                ro = np.zeros(n_final + 1, np.uint32)
                ro[1:] = np.cumsum([len(ids) for _, ids in repetition_penalties])
                ids = np.ascontiguousarray(np.concatenate([np.asarray(i_, np.uint32) for _, i_ in repetition_penalties] + [np.zeros(0, np.uint32)]), np.uint32)
                if ids.size == 0:
                    ids = np.zeros(1, np.uint32)
                pen = np.ascontiguousarray([p for p, _ in repetition_penalties], np.float32)
                hold += [ro, ids, pen]
                opts.penalty_row_offsets = ro.ctypes.data_as(C.POINTER(C.c_uint32))
                opts.penalty_token_ids = ids.ctypes.data_as(C.POINTER(C.c_uint32))
                opts.penalties = pen.ctypes.data_as(C.POINTER(C.c_float))
        _check(self.lib.ferrum_hip_model_unified_forward_ex(
            self.h, arr, len(items), int(greedy), None if opts is None else C.byref(opts),
            toks.ctypes.data_as(C.POINTER(C.c_uint32)),
            None if logits is None else logits.ctypes.data_as(C.POINTER(C.c_float))), "unified_forward")
        return (toks[:n_final] if greedy else None), (logits[:n_final] if logits is not None else None)

    def local_vocab(self):
        """(first vocabulary row, row count) this model scores: the whole vocabulary unless cfg.vocab_parallel."""
        v0, n = C.c_int(), C.c_int()
        _check(self.lib.ferrum_hip_model_local_vocab(self.h, C.byref(v0), C.byref(n)), "local_vocab")
        return v0.value, n.value

    def decode_steps(self, seq_ids, first_tokens, steps):
        n = len(seq_ids)
        sid = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        ft = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        out = np.zeros((steps, n), np.uint32)
        _check(self.lib.ferrum_hip_model_decode_steps(self.h, sid.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      ft.ctypes.data_as(C.POINTER(C.c_uint32)), n, steps,
                                                      out.ctypes.data_as(C.POINTER(C.c_uint32))), "decode_steps")
        return out

    def enable_taps(self, on=True):
        _check(self.lib.ferrum_hip_model_enable_taps(self.h, int(on)), "enable_taps")

    def read_taps(self, tokens):
        out = np.zeros((self.cfg.num_layers, tokens, self.cfg.hidden), np.float32)
        _check(self.lib.ferrum_hip_model_read_taps(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), tokens), "read_taps")
        return out

    def time_kernel(self, which, n_seqs, max_kv_len, reps=3):
        """Mean µs per launch of a hot kernel (HIP events on the model stream) and the MoE block count."""
        names = {"moe_gate_up": 0, "moe_down": 1, "attention": 2, "qkv": 3, "o": 4, "lm_head": 5, "gate_up": 6, "down": 7, "moe_pair": 8}
        us, blocks = C.c_float(), C.c_int()
        _check(self.lib.ferrum_hip_model_time_kernel(self.h, names[which], n_seqs, max_kv_len, reps, C.byref(us),
                                                     C.byref(blocks)), "time_kernel")
        return us.value, blocks.value

    def set_comm(self, comm):
        """Attach a tensor-parallel communicator (`Comm`); None detaches.  The model does not own it."""
        self._comm = comm                                       # keep it alive as long as the model uses it
        _check(self.lib.ferrum_hip_model_set_comm(self.h, comm.h if comm is not None else None), "model_set_comm")

    def tp_init(self, unique_id):
        """RCCL rank owned by the model (ferrum_hip_model_tp_init); unique_id: 128 bytes from Comm.unique_id() on rank 0."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(self.lib.ferrum_hip_model_tp_init(self.h, buf), "model_tp_init")

    def stream(self):
        s = C.c_void_p()
        _check(self.lib.ferrum_hip_model_stream(self.h, C.byref(s)), "model_stream")
        return s
