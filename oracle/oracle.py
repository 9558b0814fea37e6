"""ctypes/numpy binding of the CPU oracle (oracle/libferrum_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py — never by the product package.  Each wrapper
names the reference function its C body restates (see ferrum_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libferrum_oracle.so")


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("ferrum_oracle.c", "ferrum_oracle_model.c", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libferrum_oracle.so"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f(a):
    return _p(a, C.c_float)


def _i(a):
    return _p(a, C.c_int32)


def _u(a):
    return _p(a, C.c_uint32)


def _declare(l):
    l.fo_lcg_u32.restype = C.c_uint32
    l.fo_lcg_f32.restype = C.c_float
    l.fo_lcg_f32.argtypes = [C.POINTER(C.c_uint64), C.c_float, C.c_float]
    l.fo_make_synthetic_gptq.restype = C.c_uint64
    l.fo_make_synthetic_gptq.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    l.fo_rope_freq.restype = C.c_double
    l.fo_rope_freq.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_double] * 4
    l.fo_scale_llama3_rope_freq.restype = C.c_double
    l.fo_scale_llama3_rope_freq.argtypes = [C.c_double] * 5
    l.fo_build_rope_cache.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + \
        [C.POINTER(C.c_float)] * 2
    l.fo_bf16_round.restype = C.c_float
    l.fo_bf16_round.argtypes = [C.c_float]
    l.fo_rms_norm.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                              C.POINTER(C.c_float), C.c_int, C.c_int]
    l.fo_layer_norm.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int]
    l.fo_gelu.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_long]
    l.fo_fused_add_rms_norm.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float, C.POINTER(C.c_float),
                                                                     C.c_int, C.c_int]
    l.fo_scale_inplace.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_long]
    l.fo_add_inplace.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_long]
    l.fo_qk_norm_rope.argtypes = [C.POINTER(C.c_float)] * 5 + [C.c_int] * 4 + [C.c_float, C.c_int]
    l.fo_cpu_attention.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_int] * 7 + [C.c_float, C.c_int, C.c_int]
    l.fo_paged_attention.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                     C.c_int, C.c_int, C.POINTER(C.c_float)]
    l.fo_repetition_penalty.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_float]
    l.fo_temperature.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float]
    l.fo_top_k.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int]
    l.fo_top_p.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float]
    l.fo_multinomial.restype = C.c_uint32
    l.fo_multinomial.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_uint32]
    l.fo_greedy_sample.restype = C.c_uint32
    l.fo_alloc_new.restype = C.c_void_p
    l.fo_alloc_new.argtypes = [C.c_uint32]
    for name in ("fo_alloc_free_obj", "fo_alloc_free", "fo_alloc_acquire", "fo_alloc_register_hash"):
        getattr(l, name).restype = None
    l.fo_alloc_free_obj.argtypes = [C.c_void_p]
    l.fo_alloc_allocate.restype = C.c_int64
    l.fo_alloc_allocate.argtypes = [C.c_void_p]
    l.fo_alloc_allocate_n.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    l.fo_alloc_free.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    l.fo_alloc_acquire.argtypes = [C.c_void_p, C.c_uint32]
    l.fo_alloc_register_hash.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    l.fo_alloc_try_acquire_by_hash.restype = C.c_int64
    l.fo_alloc_try_acquire_by_hash.argtypes = [C.c_void_p, C.c_uint64]
    for name in ("fo_alloc_free_count", "fo_alloc_peak_in_use", "fo_alloc_hash_table_size"):
        getattr(l, name).restype = C.c_uint32
        getattr(l, name).argtypes = [C.c_void_p]
    l.fo_alloc_ref_count.restype = C.c_uint32
    l.fo_alloc_ref_count.argtypes = [C.c_void_p, C.c_uint32]
    l.fo_model_new.restype = C.c_void_p
    l.fo_model_free.argtypes = [C.c_void_p]
    l.fo_model_set_global.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    l.fo_model_set_layer_dense.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    l.fo_model_set_dense.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int]
    l.fo_model_set_gptq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.c_int, C.c_int, C.c_int]
    l.fo_model_release_cache.argtypes = [C.c_void_p, C.c_int]
    l.fo_model_cache_len.argtypes = [C.c_void_p, C.c_int]
    l.fo_model_read_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    l.fo_model_enable_taps.argtypes = [C.c_void_p, C.c_int]
    l.fo_model_taps.restype = C.POINTER(C.c_float)
    l.fo_model_taps.argtypes = [C.c_void_p]
    l.fo_siphash.restype = C.c_uint64
    l.fo_siphash.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t]
    l.fo_block_hash_chain.restype = C.c_int
    l.fo_block_hash_chain.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    l.fo_model_last_route_gap.restype = C.c_float
    l.fo_model_last_route_gap.argtypes = [C.c_void_p]
    l.fo_model_last_route_gap_rel.restype = C.c_float
    l.fo_model_last_route_gap_rel.argtypes = [C.c_void_p]
    l.fo_model_forward.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float)]


def set_threads(n):
    """Worker threads of the oracle's row/head/token loops (results are bit-identical for any count; default 1 = the
    reference's single-threaded CPU path)."""
    lib().fo_set_threads(int(n))


def get_threads():
    return int(lib().fo_get_threads())


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ── deterministic inputs (gptq_parity_test.rs:28-104) ────────────────────────
class Lcg:
    def __init__(self, seed):
        self.state = C.c_uint64(seed)

    def u32(self):
        return lib().fo_lcg_u32(C.byref(self.state))

    def f32(self, lo, hi):
        return lib().fo_lcg_f32(C.byref(self.state), lo, hi)

    def array_f32(self, n, lo, hi):
        return np.array([self.f32(lo, hi) for _ in range(n)], dtype=np.float32)


def make_synthetic_gptq(k, n, group, seed, symmetric=False):
    qweight = np.zeros((k // 8, n), np.int32)
    scales = np.zeros((k // group, n), np.float32)
    qzeros = np.zeros((k // group, n // 8), np.int32)
    lib().fo_make_synthetic_gptq(k, n, group, seed, int(symmetric), _i(qweight), _f(scales), _i(qzeros))
    return qweight, scales, qzeros


def make_desc_act_g_idx(k, group):
    g = np.zeros(k, np.int32)
    lib().fo_make_desc_act_g_idx(k, group, _i(g))
    return g


# ── dense ops (cpu.rs) ───────────────────────────────────────────────────────
def gemm(a, b, m, n, k):
    a, b = f32(a), f32(b)
    out = np.zeros((m, n), np.float32)
    lib().fo_gemm(_f(a), _f(b), _f(out), m, n, k)
    return out


def rms_norm(x, w, eps):
    x, w = f32(x), f32(w)
    t, d = x.shape
    out = np.zeros_like(x)
    lib().fo_rms_norm(_f(x), _f(w), eps, _f(out), t, d)
    return out


def layer_norm(x, gamma, beta, eps):
    x, gamma, beta = f32(x), f32(gamma), f32(beta)
    t, d = x.shape
    out = np.zeros_like(x)
    lib().fo_layer_norm(_f(x), _f(gamma), _f(beta), eps, _f(out), t, d)
    return out


def gelu(x):
    x = f32(x)
    out = np.zeros_like(x)
    lib().fo_gelu(_f(x), _f(out), x.size)
    return out


def fused_add_rms_norm(residual, x, w, eps):
    """Returns (new_residual, out)."""
    r, x, w = f32(residual).copy(), f32(x), f32(w)
    t, d = r.shape
    out = np.zeros_like(r)
    lib().fo_fused_add_rms_norm(_f(r), _f(x), _f(w), eps, _f(out), t, d)
    return r, out


def embedding_lookup(table, ids):
    table = f32(table)
    ids = np.ascontiguousarray(ids, np.uint32)
    out = np.zeros((len(ids), table.shape[1]), np.float32)
    lib().fo_embedding_lookup(_f(table), _u(ids), len(ids), _f(out), table.shape[1])
    return out


def split_qkv(qkv, q_dim, kv_dim):
    qkv = f32(qkv)
    t = qkv.shape[0]
    q = np.zeros((t, q_dim), np.float32)
    k = np.zeros((t, kv_dim), np.float32)
    v = np.zeros((t, kv_dim), np.float32)
    lib().fo_split_qkv(_f(qkv), _f(q), _f(k), _f(v), t, q_dim, kv_dim)
    return q, k, v


def fused_silu_mul_split(gate_up, im):
    gate_up = f32(gate_up)
    t = gate_up.shape[0]
    out = np.zeros((t, im), np.float32)
    lib().fo_fused_silu_mul_split(_f(gate_up), _f(out), t, im)
    return out


def fused_gelu_tanh_mul_split(gate_up, im):
    gate_up = f32(gate_up)
    t = gate_up.shape[0]
    out = np.zeros((t, im), np.float32)
    lib().fo_fused_gelu_tanh_mul_split(_f(gate_up), _f(out), t, im)
    return out


def scale_inplace(buf, scale):
    b = f32(buf).copy()
    lib().fo_scale_inplace(_f(b), scale, b.size)
    return b


def add_inplace(residual, x):
    r = f32(residual).copy()
    x = f32(x)
    lib().fo_add_inplace(_f(r), _f(x), r.size)
    return r


def qk_norm_rope(inp, norm_w, cos, sin, tokens, heads, head_dim, pos_offset, eps, mode):
    """input [T,heads,hd] → output [heads,T,hd] (cpu.rs:1706-1783)."""
    inp, norm_w, cos, sin = f32(inp), f32(norm_w), f32(cos), f32(sin)
    out = np.zeros((heads, tokens, head_dim), np.float32)
    lib().fo_qk_norm_rope(_f(inp), _f(norm_w), _f(cos), _f(sin), _f(out), tokens, heads, head_dim,
                          pos_offset, eps, mode)
    return out


def transpose_head_to_token(src, tokens, heads, dim):
    src = f32(src)
    dst = np.zeros((tokens, heads, dim), np.float32)
    lib().fo_transpose_head_to_token(_f(src), _f(dst), tokens, heads, dim)
    return dst


def cpu_attention(q, k, v, q_len, kv_len, causal, pos_offset, nh, nkv, d, scale=None,
                  kv_seq_stride=0, sliding_window=0):
    """q [nh,q_len,d], k/v [nkv,stride,d] → out [nh,q_len,d] (cpu.rs:2179-2259)."""
    q, k, v = f32(q), f32(k), f32(v)
    out = np.zeros((nh, q_len, d), np.float32)
    if scale is None:
        scale = 1.0 / np.sqrt(np.float32(d))
    lib().fo_cpu_attention(_f(q), _f(k), _f(v), _f(out), q_len, kv_len, int(causal), pos_offset,
                           nh, nkv, d, float(scale), kv_seq_stride, sliding_window)
    return out


def paged_attention(query, q_tokens, nh, nkv, hd, pool_k, pool_v, block_table, block_size, kv_len):
    """ferrum-kv/src/attention.rs:30-114; pools [blocks,block_size,nkv,hd]."""
    query, pool_k, pool_v = f32(query), f32(pool_k), f32(pool_v)
    bt = np.ascontiguousarray(block_table, np.int32)
    out = np.zeros((q_tokens, nh, hd), np.float32)
    rc = lib().fo_paged_attention(_f(query), q_tokens, nh, nkv, hd, _f(pool_k), _f(pool_v), _i(bt),
                                  block_size, kv_len, _f(out))
    if rc != 0:
        raise ValueError("kv_len must be positive")
    return out


def dequant_gptq(qweight, scales, qzeros, group, k, n, g_idx=None):
    """→ w [n,k] f32 (cpu.rs:2283-2315 / gptq_parity_test.rs:168-189)."""
    qweight = np.ascontiguousarray(qweight, np.int32)
    scales = f32(scales)
    qzeros = np.ascontiguousarray(qzeros, np.int32)
    w = np.zeros((n, k), np.float32)
    gp = None
    if g_idx is not None:
        g_idx = np.ascontiguousarray(g_idx, np.int32)
        gp = _i(g_idx)
    rc = lib().fo_dequant_gptq(_i(qweight), _f(scales), _i(qzeros), gp, 4, group, k, n, _f(w))
    assert rc == 0
    return w


# ── MoE ──────────────────────────────────────────────────────────────────────
def route_topk(logits, num_experts, top_k, norm_topk_prob):
    logits = f32(logits)
    b = logits.shape[0]
    ids = np.zeros((b, top_k), np.uint32)
    w = np.zeros((b, top_k), np.float32)
    lib().fo_route_topk(_f(logits), b, num_experts, top_k, int(norm_topk_prob), _u(ids), _f(w))
    return ids, w


def bucket_plan(expert_ids, batch, num_experts, top_k):
    ids = np.ascontiguousarray(expert_ids, np.uint32)
    offsets = np.zeros(num_experts + 1, np.uint32)
    packed = np.zeros(batch * top_k, np.uint32)
    pairs = np.zeros(batch * top_k, np.int32)
    lib().fo_bucket_plan(_u(ids), batch, num_experts, top_k, _u(offsets), _u(packed), _i(pairs))
    return offsets, packed, pairs


def moe_align_block_size(expert_ids, num_experts, block_size):
    ids = np.ascontiguousarray(expert_ids, np.int32).reshape(-1)
    n = ids.size
    sorted_max = n + num_experts * block_size
    sorted_ids = np.zeros(sorted_max, np.int32)
    block_ids = np.full(sorted_max // block_size + 1, -1, np.int32)
    total = np.zeros(1, np.int32)
    nb = lib().fo_moe_align_block_size(_i(ids), n, num_experts, block_size, sorted_max, _i(sorted_ids),
                                       _i(block_ids), _i(total))
    return sorted_ids, block_ids[:nb], int(total[0])


def moe_forward_cpu(x, hidden, inter, top_k, expert_ids, expert_weights, gate_up_w, down_w):
    x = f32(x)
    b = x.shape[0]
    ids = np.ascontiguousarray(expert_ids, np.uint32)
    ew = f32(expert_weights)
    gw, dw = f32(gate_up_w), f32(down_w)
    out = np.zeros((b, hidden), np.float32)
    lib().fo_moe_forward_cpu(_f(x), b, hidden, inter, top_k, _u(ids), _f(ew), _f(gw), _f(dw), _f(out))
    return out


def compute_ids_tpe(selected, num_experts, batch, top_k):
    sel = np.ascontiguousarray(selected, np.uint32)
    tpe = np.zeros(num_experts, np.int32)
    ids = np.zeros(num_experts * max(1, batch * top_k), np.int32)
    mpe = lib().fo_compute_ids_tpe(_u(sel), num_experts, batch, top_k, _i(tpe), _i(ids))
    return tpe, ids[:num_experts * mpe], mpe


# ── sampling ─────────────────────────────────────────────────────────────────
def argmax_rows(logits):
    logits = f32(logits)
    m, n = logits.shape
    out = np.zeros(m, np.uint32)
    lib().fo_argmax_rows(_f(logits), m, n, _u(out))
    return out


def greedy_sample(logits):
    logits = f32(logits)
    return int(lib().fo_greedy_sample(_f(logits), logits.size))


def repetition_penalty(logits, prev, penalty):
    l = f32(logits).copy()
    prev = np.ascontiguousarray(prev, np.uint32)
    lib().fo_repetition_penalty(_f(l), l.size, _u(prev), prev.size, penalty)
    return l


def temperature(logits, t):
    l = f32(logits).copy()
    lib().fo_temperature(_f(l), l.size, t)
    return l


def top_k(logits, k):
    l = f32(logits).copy()
    lib().fo_top_k(_f(l), l.size, k)
    return l


def top_p(logits, p):
    l = f32(logits).copy()
    lib().fo_top_p(_f(l), l.size, p)
    return l


def multinomial(logits, rng_u32):
    l = f32(logits)
    return int(lib().fo_multinomial(_f(l), l.size, rng_u32))


# ── RoPE ─────────────────────────────────────────────────────────────────────
def build_rope_cache(theta, head_dim, max_seq, scaling_kind=0, p=(0.0, 0.0, 0.0, 0.0)):
    half = head_dim // 2
    cos = np.zeros((max_seq, half), np.float32)
    sin = np.zeros((max_seq, half), np.float32)
    lib().fo_build_rope_cache(theta, head_dim, max_seq, scaling_kind, *p, _f(cos), _f(sin))
    return cos, sin


def rope_freq(theta, head_dim, pair_idx, scaling_kind=0, p=(0.0, 0.0, 0.0, 0.0)):
    return lib().fo_rope_freq(theta, head_dim, pair_idx, scaling_kind, *p)


def bf16_round(x):
    return lib().fo_bf16_round(x)


# ── BlockAllocator (paged_pool.rs:106-365) ───────────────────────────────────
class BlockAllocator:
    def __init__(self, num_blocks):
        self._h = lib().fo_alloc_new(num_blocks)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_alloc_free_obj(self._h)
            self._h = None

    def allocate(self):
        b = lib().fo_alloc_allocate(self._h)
        if b < 0:
            raise RuntimeError("paged KV pool exhausted")
        return int(b)

    def allocate_n(self, n):
        out = np.zeros(max(n, 1), np.uint32)
        if lib().fo_alloc_allocate_n(self._h, n, _u(out)) != 0:
            raise RuntimeError("paged KV pool exhausted")
        return [int(x) for x in out[:n]]

    def free(self, blocks):
        b = np.ascontiguousarray(blocks, np.uint32)
        lib().fo_alloc_free(self._h, _u(b), b.size)

    def acquire(self, block):
        lib().fo_alloc_acquire(self._h, block)

    def register_block_hash(self, block, h):
        lib().fo_alloc_register_hash(self._h, block, h)

    def try_acquire_by_hash(self, h):
        b = lib().fo_alloc_try_acquire_by_hash(self._h, h)
        return None if b < 0 else int(b)

    def free_count(self):
        return lib().fo_alloc_free_count(self._h)

    def ref_count(self, b):
        return lib().fo_alloc_ref_count(self._h, b)

    def peak_in_use(self):
        return lib().fo_alloc_peak_in_use(self._h)

    def hash_table_size(self):
        return lib().fo_alloc_hash_table_size(self._h)


# ── model (llama_family.rs CPU path / moe_forward_cpu) ───────────────────────
class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_layers", "hidden", "num_heads", "num_kv_heads", "head_dim", "intermediate", "vocab",
        "max_seq_len", "has_qk_norm", "activation", "num_experts", "top_k", "expert_inter",
        "norm_topk_prob", "rope_scaling_kind", "sliding_window")] + [
        ("rms_eps", C.c_float), ("_pad", C.c_float), ("rope_theta", C.c_double),
        ("rope_p0", C.c_double), ("rope_p1", C.c_double), ("rope_p2", C.c_double), ("rope_p3", C.c_double),
        ("sliding_window_pattern", C.c_int32), ("sandwich_norms", C.c_int32), ("embed_scale", C.c_float), ("_pad2", C.c_float),
        ("rope_local_theta", C.c_double)]


def siphash(c, d, k0, k1, data):
    """SipHash-c-d of `data` (bytes) with key (k0, k1)."""
    return int(lib().fo_siphash(c, d, k0, k1, bytes(data), len(data)))


def block_hash_chain(tokens, block_size=16):
    """paged_pool.rs:89-98: chained content hash per full block of `tokens`."""
    t = np.ascontiguousarray(tokens, np.uint32)
    out = np.zeros(max(len(t) // block_size, 1), np.uint64)
    n = lib().fo_block_hash_chain(t.ctypes.data, len(t), block_size, out.ctypes.data)
    return out[:n].copy()


class OracleModel:
    """CPU restatement of LlamaFamilyModel<CpuBackend> / Qwen3-MoE (see ferrum_oracle_model.c)."""

    GLOBAL = {"embed": 0, "lm_head": 1, "final_norm": 2}
    LAYER_DENSE = {"input_ln": 0, "post_ln": 1, "q_norm": 2, "k_norm": 3, "router": 4, "post_attn_ln": 5, "post_ffn_ln": 6,
                   "qkv_bias": 7}
    GPTQ = {"qkv": 0, "o": 1, "gate_up": 2, "down": 3, "expert_gate_up": 4, "expert_down": 5}

    def __init__(self, **kw):
        self.cfg = ModelCfg()
        defaults = dict(max_seq_len=512, has_qk_norm=0, activation=0, num_experts=0, top_k=0,
                        expert_inter=0, norm_topk_prob=1, rope_scaling_kind=0, sliding_window=0,
                        rms_eps=1e-6, rope_theta=1e6, intermediate=0)
        defaults.update(kw)
        for k, v in defaults.items():
            setattr(self.cfg, k, v)
        self._h = lib().fo_model_new(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_model_free(self._h)
            self._h = None

    def set_global(self, name, data):
        d = f32(data)
        lib().fo_model_set_global(self._h, self.GLOBAL[name], _f(d))

    def set_layer_dense(self, layer, name, data):
        d = f32(data)
        lib().fo_model_set_layer_dense(self._h, layer, self.LAYER_DENSE[name], _f(d))

    def set_gptq(self, layer, name, qweight, scales, qzeros, group, k, n, expert=0, g_idx=None):
        qw = np.ascontiguousarray(qweight, np.int32)
        sc = f32(scales)
        qz = np.ascontiguousarray(qzeros, np.int32)
        gp = None
        if g_idx is not None:
            g_idx = np.ascontiguousarray(g_idx, np.int32)
            gp = _i(g_idx)
        rc = lib().fo_model_set_gptq(self._h, layer, self.GPTQ[name], expert, _i(qw), _f(sc), _i(qz), gp,
                                     group, k, n)
        assert rc == 0

    def set_dense(self, layer, name, weight, k, n):
        """Unquantised projection: weight [n, k] f32 (DenseLinear)."""
        which = {"qkv": 0, "o": 1, "gate_up": 2, "down": 3}[name]
        w = np.ascontiguousarray(weight, np.float32).reshape(n, k)
        rc = lib().fo_model_set_dense(self._h, layer, which, _f(w), k, n)
        if rc != 0:
            raise RuntimeError(f"fo_model_set_dense({name}) failed: {rc}")

    def forward(self, cache_id, tokens, pos_offset, all_logits=False):
        toks = np.ascontiguousarray(tokens, np.uint32)
        v = self.cfg.vocab
        last = np.zeros(v, np.float32)
        al = np.zeros((len(toks), v), np.float32) if all_logits else None
        rc = lib().fo_model_forward(self._h, cache_id, _u(toks), len(toks), pos_offset, _f(last),
                                    _f(al) if al is not None else None)
        if rc != 0:
            raise RuntimeError(f"fo_model_forward rc={rc}")
        return (last, al) if all_logits else last

    def release(self, cache_id):
        lib().fo_model_release_cache(self._h, cache_id)

    def cache_len(self, cache_id):
        return lib().fo_model_cache_len(self._h, cache_id)

    def read_kv(self, cache_id, layer, is_v):
        n = self.cache_len(cache_id)
        out = np.zeros((n, self.cfg.num_kv_heads, self.cfg.head_dim), np.float32)
        lib().fo_model_read_kv(self._h, cache_id, layer, int(is_v), _f(out))
        return out

    def last_route_gap(self):
        """Smallest k-th/(k+1)-th router-logit gap over the last forward's layers and tokens (inf for dense models)."""
        return float(lib().fo_model_last_route_gap(self._h))

    def last_route_gap_rel(self):
        """The same gap divided by that token's router-logit spread (max − min)."""
        return float(lib().fo_model_last_route_gap_rel(self._h))

    def enable_taps(self, max_tokens):
        self._tap_tokens = max_tokens
        lib().fo_model_enable_taps(self._h, max_tokens)

    def taps(self, tokens):
        p = lib().fo_model_taps(self._h)
        n = self.cfg.num_layers * self._tap_tokens * self.cfg.hidden
        a = np.ctypeslib.as_array(p, shape=(n,)).reshape(self.cfg.num_layers, self._tap_tokens, self.cfg.hidden)
        return a[:, :tokens, :].copy()
