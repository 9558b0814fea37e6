"""ctypes mirror of the reference's backend operator traits over the C ABI (include/ferrum_hip.h).

`HipBackend` exposes the ops of `Backend` / `BackendPagedKv` / `BackendQuantMarlin` /
`BackendMoeFused` (crates/ferrum-kernels/src/backend/traits.rs, capabilities.rs) with the same
names and argument meaning; `B::Buffer` is a torch CUDA tensor (device memory only — torch does
no arithmetic here), `B::Context` is a HIP stream + workspace.  Errors follow the trait: core ops
raise `RuntimeError` (the reference panics), capability ops raise `Unsupported` where the
reference returns `FerrumError::unsupported`.

There is NO fallback path: a missing `libferrum_hip.so` raises at load.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FERRUM_HIP_LIB") or os.path.join(_HERE, "lib", "libferrum_hip.so")   # (override: development builds, e.g. make EXPERIMENTS=1 into lib_exp/)
_lib = None

vp, i32p, u32p, f32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)


class Unsupported(RuntimeError):
    """Mirror of FerrumError::unsupported (the trait's "fall back" signal)."""


def build_library(force=False, jobs=8):
    """hipcc cross-compiles for gfx950 without a GPU (make -C csrc)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing — build it with __graft_entry__.build() "
                           "(the HIP extension is the product; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.ferrum_hip_last_error.restype = C.c_char_p
    lib.ferrum_hip_paged_pool_bytes.restype = C.c_size_t
    for n in ("free_count", "ref_count", "peak_in_use", "hash_table_size"):
        getattr(lib, "ferrum_hip_block_allocator_" + n).restype = C.c_uint32
    lib.ferrum_native_op_descriptor.restype = C.c_void_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc == 0:
        return
    msg = load_library().ferrum_hip_last_error().decode(errors="replace")
    if rc == 3:
        raise Unsupported(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Context:
    """B::Context: stream + split-K / split-KV workspace (traits.rs:69 new_context)."""

    def __init__(self, workspace_bytes=256 << 20):
        import torch
        self.lib = load_library()
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.ws = C.c_void_p()
        _check(self.lib.ferrum_hip_workspace_create(C.byref(self.ws), C.c_size_t(workspace_bytes)), "workspace_create")

    def sync(self):
        _check(self.lib.ferrum_hip_stream_synchronize(self.stream), "sync")

    def __del__(self):
        try:
            if getattr(self, "ws", None):
                self.lib.ferrum_hip_workspace_destroy(self.ws)
                self.ws = None
        except Exception:
            pass


class GptqLinear:
    """`GptqLinear<B>` / `Linear<B>` (ferrum-quantization/src/gptq.rs:42-112, linear.rs:109-129)."""

    def __init__(self, handle, k, n):
        self.handle, self.in_features, self.out_features = handle, k, n

    @staticmethod
    def from_raw(qweight, scales, qzeros, g_idx, bias, bits, group_size, k, n):
        return HipBackend.load_gptq(qweight, scales, qzeros, g_idx, bias, bits, group_size, k, n)

    def forward(self, ctx, inp, out, m):
        _check(ctx.lib.ferrum_hip_gptq_linear_forward_f16(self.handle, _ptr(inp), _ptr(out), m, ctx.ws, ctx.stream),
               "gptq_linear_forward")

    def __del__(self):
        try:
            if self.handle:
                load_library().ferrum_hip_gptq_free(self.handle)
                self.handle = None
        except Exception:
            pass


class ExpertStack(GptqLinear):
    """`MarlinExpertStack<B>` (ferrum-kernels/src/marlin_expert_stack.rs:35-140)."""

    def __init__(self, handle, k, n, num_experts, fused):
        super().__init__(handle, k, n)
        self.num_experts, self.fused = num_experts, fused

    def n_per_expert(self):
        return self.out_features

    def k(self):
        return self.in_features

    def gemm_phase_vllm(self, ctx, inp, sorted_token_ids, expert_ids, num_tokens_past_padded, output, prob_m,
                        moe_block_size, top_k, max_blocks, fused_silu_mul=False):
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_f16(self.handle, _ptr(inp), _ptr(sorted_token_ids), _ptr(expert_ids),
                                                     _ptr(num_tokens_past_padded), _ptr(output), prob_m, moe_block_size,
                                                     top_k, max_blocks, int(fused_silu_mul), ctx.stream),
               "moe_gemm_phase")


    def gemm_phase_batched(self, ctx, inp, dispatches, output, k, fused_silu_mul=False):
        """MarlinExpertStack::gemm_phase_batched: dispatches = [(expert, in_row_offset, out_row_offset, m)] (host)."""
        import numpy as np
        d = np.ascontiguousarray(np.asarray(dispatches, np.int32).reshape(-1, 4))
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_batched_f16(self.handle, _ptr(inp), d.ctypes.data_as(i32p), len(d), _ptr(output), k,
                                                             int(fused_silu_mul), ctx.stream), "gemm_phase_batched")

    def gemm_phase_inline_align(self, ctx, inp, expert_ids_per_pair, output, prob_m, num_experts, top_k, max_blocks,
                                fused_silu_mul=False):
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_inline_align_f16(self.handle, _ptr(inp), _ptr(expert_ids_per_pair),
                                                                  _ptr(output), prob_m, num_experts, top_k, max_blocks,
                                                                  int(fused_silu_mul), ctx.stream),
               "moe_gemm_phase_inline_align")


    def gemm_phase_expert_major(self, ctx, inp, expert_ids_per_pair, output, prob_m, num_experts, top_k, fused_silu_mul=False):
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_expert_major_f16(self.handle, _ptr(inp), _ptr(expert_ids_per_pair),
                                                                  _ptr(output), prob_m, num_experts, top_k,
                                                                  int(fused_silu_mul), ctx.stream),
               "moe_gemm_phase_expert_major")

    def gemm_phase_expert_major_pair(self, ctx, down_stack, inp, expert_ids_per_pair, act_out, output, prob_m, num_experts, top_k):
        """gate_up (+ silu·mul) and down as ONE launch; self is the fused gate_up stack."""
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_expert_major_pair_f16(self.handle, down_stack.handle, _ptr(inp),
                                                                       _ptr(expert_ids_per_pair), _ptr(act_out), _ptr(output),
                                                                       prob_m, num_experts, top_k, ctx.stream),
               "moe_gemm_phase_expert_major_pair")

    def gemm_phase_block_major_pair(self, ctx, down_stack, inp, expert_ids_per_pair, act_out, output, prob_m, num_experts, top_k,
                                    max_blocks):
        """≤ 64 pairs: gate_up (+ silu·mul) and down as ONE block-major launch; self is the fused gate_up stack."""
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_block_major_pair_f16(self.handle, down_stack.handle, _ptr(inp),
                                                                      _ptr(expert_ids_per_pair), _ptr(act_out), _ptr(output),
                                                                      prob_m, num_experts, top_k, max_blocks, ctx.stream),
               "moe_gemm_phase_block_major_pair")

    def pair_timeouts(self, ctx):
        import ctypes
        t = ctypes.c_uint(0)
        _check(ctx.lib.ferrum_hip_moe_pair_status(self.handle, ctypes.byref(t)), "moe_pair_status")
        return t.value

    def gemm_phase_merge_route(self, ctx, inp, cand, stats, output, tokens, num_parts, top_k, norm_topk_prob, num_experts,
                               max_blocks, expert_ids_out, expert_weights_out, sorted_out, block_ids_out, total_out,
                               fused_silu_mul=False):
        _check(ctx.lib.ferrum_hip_moe_gemm_phase_merge_route_f16(
            self.handle, _ptr(inp), _ptr(cand), _ptr(stats), _ptr(output), tokens, num_parts, top_k, int(norm_topk_prob),
            num_experts, max_blocks, int(fused_silu_mul), _ptr(expert_ids_out), _ptr(expert_weights_out), _ptr(sorted_out),
            _ptr(block_ids_out), _ptr(total_out), ctx.stream), "moe_gemm_phase_merge_route")


def _np_i32(a):
    import numpy as np
    return np.ascontiguousarray(a, dtype=np.int32)


def _np_f32(a):
    import numpy as np
    return np.ascontiguousarray(a, dtype=np.float32)


class HipBackend:
    """Static ops, names and argument order as in the reference traits."""

    @staticmethod
    def new_context(workspace_bytes=256 << 20):
        return Context(workspace_bytes)

    @staticmethod
    def sync(ctx):
        ctx.sync()

    # ── Backend (traits.rs:190-1591) ─────────────────────────────────────────
    @staticmethod
    def gemm(ctx, a, b, out, m, n, k):
        import torch
        fn = ctx.lib.ferrum_hip_gemm_f16_f32out if out.dtype == torch.float32 else ctx.lib.ferrum_hip_gemm_f16
        _check(fn(_ptr(a), _ptr(b), _ptr(out), m, n, k, ctx.ws, ctx.stream), "gemm")

    @staticmethod
    def dense_repack_f16t(ctx, w_rowmajor, n, k):
        """Load-time transform of a dense fp16 weight [N,K] into MFMA-fragment-major tiles (DenseLinear)."""
        import torch
        ctx.lib.ferrum_hip_dense_f16t_bytes.restype = C.c_size_t
        nbytes = ctx.lib.ferrum_hip_dense_f16t_bytes(n, k)
        out = torch.empty(nbytes // 2, dtype=torch.float16, device=w_rowmajor.device)
        _check(ctx.lib.ferrum_hip_dense_repack_f16t(_ptr(w_rowmajor), _ptr(out), n, k, ctx.stream), "dense_repack_f16t")
        return out

    @staticmethod
    def gemm_f16t(ctx, a, b_tiled, out, m, n, k):
        import torch
        fn = ctx.lib.ferrum_hip_gemm_f16t_f32out if out.dtype == torch.float32 else ctx.lib.ferrum_hip_gemm_f16t
        _check(fn(_ptr(a), _ptr(b_tiled), _ptr(out), m, n, k, ctx.ws, ctx.stream), "gemm_f16t")

    @staticmethod
    def rms_norm(ctx, x, w, eps, out, tokens, dim):
        _check(ctx.lib.ferrum_hip_rms_norm_f16(_ptr(x), _ptr(w), C.c_float(eps), _ptr(out), tokens, dim, ctx.stream), "rms_norm")

    @staticmethod
    def fused_add_rms_norm(ctx, residual, x, w, eps, out, tokens, dim):
        _check(ctx.lib.ferrum_hip_fused_add_rms_norm_f16(_ptr(residual), _ptr(x), _ptr(w), C.c_float(eps), _ptr(out), tokens,
                                                         dim, ctx.stream), "fused_add_rms_norm")

    @staticmethod
    def embedding_lookup(ctx, table, ids, out, dim):
        _check(ctx.lib.ferrum_hip_embedding_lookup_f16(_ptr(table), _ptr(ids), _ptr(out), ids.numel(), dim, ctx.stream),
               "embedding_lookup")

    @staticmethod
    def layer_norm(ctx, x, gamma, beta, eps, out, tokens, dim):
        _check(ctx.lib.ferrum_hip_layer_norm_f16(_ptr(x), _ptr(gamma), _ptr(beta), C.c_float(eps), _ptr(out), tokens, dim, ctx.stream), "layer_norm")

    @staticmethod
    def gelu(ctx, x, out, length):
        _check(ctx.lib.ferrum_hip_gelu_f16(_ptr(x), _ptr(out), C.c_size_t(length), ctx.stream), "gelu")

    @staticmethod
    def fused_silu_mul_split(ctx, gate_up, out, tokens, im):
        _check(ctx.lib.ferrum_hip_fused_silu_mul_split_f16(_ptr(gate_up), _ptr(out), tokens, im, ctx.stream), "fused_silu_mul_split")

    @staticmethod
    def fused_gelu_tanh_mul_split(ctx, gate_up, out, tokens, im):
        _check(ctx.lib.ferrum_hip_fused_gelu_tanh_mul_split_f16(_ptr(gate_up), _ptr(out), tokens, im, ctx.stream),
               "fused_gelu_tanh_mul_split")

    @staticmethod
    def scale_inplace(ctx, buf, scale, length):
        _check(ctx.lib.ferrum_hip_scale_inplace_f16(_ptr(buf), C.c_float(scale), C.c_size_t(length), ctx.stream), "scale_inplace")

    @staticmethod
    def add_inplace(ctx, residual, x, length):
        _check(ctx.lib.ferrum_hip_add_inplace_f16(_ptr(residual), _ptr(x), C.c_size_t(length), ctx.stream), "add_inplace")

    @staticmethod
    def add_bias(ctx, data, bias, rows, cols):
        _check(ctx.lib.ferrum_hip_add_bias_f16(_ptr(data), _ptr(bias), rows, cols, ctx.stream), "add_bias")

    @staticmethod
    def argmax_rows_f16(ctx, logits, m, n):
        """Returns the m token ids on the host (traits.rs:1534); logits fp16 or fp32."""
        import torch
        out = torch.empty(m, dtype=torch.int32, device=logits.device)
        fn = ctx.lib.ferrum_hip_argmax_rows_f32_ws if logits.dtype == torch.float32 else ctx.lib.ferrum_hip_argmax_rows_f16_ws
        _check(fn(_ptr(logits), _ptr(out), None, 0, m, n, ctx.ws, ctx.stream), "argmax_rows")
        ctx.sync()
        return out.cpu().numpy().astype("uint32")

    @staticmethod
    def argmax_rows_f16_masked(ctx, logits, valid_token_mask, mask_len, m, n):
        import torch
        out = torch.empty(m, dtype=torch.int32, device=logits.device)
        fn = ctx.lib.ferrum_hip_argmax_rows_f32_ws if logits.dtype == torch.float32 else ctx.lib.ferrum_hip_argmax_rows_f16_ws
        _check(fn(_ptr(logits), _ptr(out), _ptr(valid_token_mask), mask_len, m, n, ctx.ws, ctx.stream), "argmax_rows_masked")
        ctx.sync()
        return out.cpu().numpy().astype("uint32")

    @staticmethod
    def argmax_rows_f16_sparse_repetition_penalty(ctx, logits, valid_token_mask, row_offsets, token_ids,
                                                  repetition_penalties, total_token_ids, m, n):
        import torch
        fn = (ctx.lib.ferrum_hip_apply_repetition_penalties_sparse_f32 if logits.dtype == torch.float32
              else ctx.lib.ferrum_hip_apply_repetition_penalties_sparse_f16)
        _check(fn(_ptr(logits), _ptr(row_offsets), _ptr(token_ids), _ptr(repetition_penalties), m, n, ctx.stream),
               "apply_repetition_penalties_sparse")
        if valid_token_mask is not None:
            mask, mask_len = valid_token_mask
            return HipBackend.argmax_rows_f16_masked(ctx, logits, mask, mask_len, m, n)
        return HipBackend.argmax_rows_f16(ctx, logits, m, n)

    # ── BackendPagedKv (traits.rs:1622-1904) ─────────────────────────────────
    @staticmethod
    def supports_paged_kv():
        return True

    @staticmethod
    def supports_varlen_qkv():
        return True

    # ── contiguous-KV lane (kv_layer.rs:370-513) ──────────────────────────────
    @staticmethod
    def split_qkv(ctx, qkv, q, k, v, tokens, q_dim, kv_dim):
        _check(ctx.lib.ferrum_hip_split_qkv_f16(_ptr(qkv), _ptr(q), _ptr(k), _ptr(v), tokens, q_dim, kv_dim, ctx.stream), "split_qkv")

    @staticmethod
    def qk_norm_rope(ctx, inp, norm_w, cos, sin, out, tokens, heads, head_dim, pos_offset, eps, mode):
        _check(ctx.lib.ferrum_hip_qk_norm_rope_f16(_ptr(inp), _ptr(norm_w), _ptr(cos), _ptr(sin), _ptr(out), tokens, heads,
                                                   head_dim, pos_offset, C.c_float(eps), mode, ctx.stream), "qk_norm_rope")

    @staticmethod
    def kv_cache_append_head_major(ctx, cache_k, cache_v, cache_len, capacity, new_k, new_v, new_tokens, nkv, hd):
        _check(ctx.lib.ferrum_hip_kv_cache_append_head_major_f16(_ptr(cache_k), _ptr(cache_v), cache_len, capacity, _ptr(new_k),
                                                                 _ptr(new_v), new_tokens, nkv, hd, ctx.stream),
               "kv_cache_append_head_major")

    @staticmethod
    def transpose_head_to_token(ctx, src, dst, tokens, heads, dim):
        _check(ctx.lib.ferrum_hip_transpose_head_to_token_f16(_ptr(src), _ptr(dst), tokens, heads, dim, ctx.stream), "transpose_head_to_token")

    @staticmethod
    def transpose_token_to_head(ctx, src, dst, tokens, heads, dim):
        _check(ctx.lib.ferrum_hip_transpose_token_to_head_f16(_ptr(src), _ptr(dst), tokens, heads, dim, ctx.stream), "transpose_token_to_head")

    @staticmethod
    def copy_slice(ctx, src, src_offset, dst, dst_offset, length):
        _check(ctx.lib.ferrum_hip_copy_slice_f16(_ptr(src), C.c_size_t(src_offset), _ptr(dst), C.c_size_t(dst_offset),
                                                 C.c_size_t(length), ctx.stream), "copy_slice")

    @staticmethod
    def scaled_add_inplace(ctx, dst, src, scale, length):
        _check(ctx.lib.ferrum_hip_scaled_add_inplace_f16(_ptr(dst), _ptr(src), C.c_float(scale), C.c_size_t(length), ctx.stream),
               "scaled_add_inplace")

    @staticmethod
    def flash_attention(ctx, q, k, v, out, batch, q_len, kv_len, pos_offset, num_heads, num_kv_heads, head_dim, causal=True,
                        scale=None, kv_seq_stride=0, sliding_window=0):
        sc = (1.0 / head_dim ** 0.5) if scale is None else scale
        _check(ctx.lib.ferrum_hip_flash_attention_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), batch, q_len, kv_len, pos_offset,
                                                      num_heads, num_kv_heads, head_dim, int(causal), C.c_float(sc), kv_seq_stride,
                                                      sliding_window, ctx.stream), "flash_attention")

    @staticmethod
    def alloc_paged_pool(num_blocks, kv_heads, head_dim, device="cuda"):
        """Zero-initialised pool in the native tile layout (csrc/kv_layout.h)."""
        import torch
        return torch.zeros(num_blocks * kv_heads * 16 * head_dim, dtype=torch.float16, device=device)

    @staticmethod
    def split_qkv_norm_rope_into_paged_cache_varlen(ctx, qkv, q_norm_w, k_norm_w, cos, sin, q_out, cache_k, cache_v,
                                                    cu_seqlens_q, pos_offsets, block_tables, num_seqs, m_total, q_heads,
                                                    kv_heads, head_dim, eps, qk_mode, block_size, max_blocks_per_seq):
        _check(ctx.lib.ferrum_hip_split_qkv_norm_rope_into_paged_cache_varlen_f16(
            _ptr(qkv), _ptr(q_norm_w), _ptr(k_norm_w), _ptr(cos), _ptr(sin), _ptr(q_out), _ptr(cache_k), _ptr(cache_v),
            _ptr(cu_seqlens_q), _ptr(pos_offsets), _ptr(block_tables), num_seqs, m_total, q_heads, kv_heads, head_dim,
            C.c_float(eps), qk_mode, block_size, max_blocks_per_seq, ctx.stream),
            "split_qkv_norm_rope_into_paged_cache_varlen")

    @staticmethod
    def paged_varlen_attention(ctx, q, k_pool, v_pool, out, cu_seqlens_q, pos_offsets, block_tables, num_seqs,
                               total_q_tokens, max_kv_len, num_heads, num_kv_heads, head_dim, sliding_window, block_size,
                               max_num_blocks_per_seq, max_q_len=0):
        _check(ctx.lib.ferrum_hip_paged_varlen_attention_f16(
            _ptr(q), _ptr(k_pool), _ptr(v_pool), _ptr(out), _ptr(cu_seqlens_q), _ptr(pos_offsets), _ptr(block_tables),
            num_seqs, total_q_tokens, max_kv_len, num_heads, num_kv_heads, head_dim, sliding_window, block_size,
            max_num_blocks_per_seq, max_q_len, ctx.ws, ctx.stream), "paged_varlen_attention")

    @staticmethod
    def paged_batched_decode_attention(ctx, q, k_pool, v_pool, out, block_tables, valid_kv_lens, num_seqs, max_kv_len,
                                       num_heads, num_kv_heads, head_dim, block_size, max_num_blocks_per_seq):
        _check(ctx.lib.ferrum_hip_paged_batched_decode_attention_f16(
            _ptr(q), _ptr(k_pool), _ptr(v_pool), _ptr(out), _ptr(block_tables), _ptr(valid_kv_lens), num_seqs, max_kv_len,
            num_heads, num_kv_heads, head_dim, block_size, max_num_blocks_per_seq, ctx.ws, ctx.stream),
            "paged_batched_decode_attention")

    @staticmethod
    def paged_decode_attention_fused_qkv(ctx, qkv, q_norm_w, k_norm_w, cos, sin, eps, qk_mode, k_pool, v_pool, out,
                                         block_tables, valid_kv_lens, num_seqs, max_kv_len, num_heads, num_kv_heads,
                                         head_dim, block_size, max_num_blocks_per_seq, sliding_window=0):
        """split_qkv_norm_rope_into_paged_cache_varlen (one token per sequence) + paged_batched_decode_attention."""
        _check(ctx.lib.ferrum_hip_paged_decode_attention_fused_qkv_f16(
            _ptr(qkv), _ptr(q_norm_w), _ptr(k_norm_w), _ptr(cos), _ptr(sin), C.c_float(eps), qk_mode, _ptr(k_pool),
            _ptr(v_pool), _ptr(out), _ptr(block_tables), _ptr(valid_kv_lens), num_seqs, max_kv_len, num_heads,
            num_kv_heads, head_dim, sliding_window, block_size, max_num_blocks_per_seq, ctx.ws, ctx.stream),
            "paged_decode_attention_fused_qkv")

    @staticmethod
    def paged_kv_read(ctx, cache_k, cache_v, block_table, kv_len, kv_heads, head_dim, block_size=16):
        import torch
        k = torch.empty(kv_len, kv_heads, head_dim, dtype=torch.float16, device=cache_k.device)
        v = torch.empty_like(k)
        _check(ctx.lib.ferrum_hip_paged_kv_read_f16(_ptr(cache_k), _ptr(cache_v), _ptr(block_table), kv_len, kv_heads,
                                                    head_dim, block_size, _ptr(k), _ptr(v), ctx.stream), "paged_kv_read")
        return k, v

    # ── BackendQuantMarlin (capabilities.rs:121-193) ─────────────────────────
    @staticmethod
    def load_gptq(qweight, scales, qzeros, g_idx, bias_host, bits, group_size, k, n):
        lib = load_library()
        qw, sc, qz = _np_i32(qweight), _np_f32(scales), _np_i32(qzeros)
        gi = None if g_idx is None else _np_i32(g_idx)
        bh = None if bias_host is None else _np_f32(bias_host)
        h = C.c_void_p()
        rc = lib.ferrum_hip_gptq_load(C.byref(h), qw.ctypes.data_as(i32p), sc.ctypes.data_as(f32p), qz.ctypes.data_as(i32p),
                                      None if gi is None else gi.ctypes.data_as(i32p),
                                      None if bh is None else bh.ctypes.data_as(f32p), bits, group_size, k, n)
        _check(rc, "load_gptq")
        return GptqLinear(h, k, n)

    @staticmethod
    def load_gptq_stacked(qweights, scales, qzeros, g_idx, bits, group_size, k, n_per_expert, fuse_gate_up=False):
        lib = load_library()
        e = len(qweights)
        qw = [_np_i32(x) for x in qweights]
        sc = [_np_f32(x) for x in scales]
        qz = [_np_i32(x) for x in qzeros]
        qwp = (i32p * e)(*[x.ctypes.data_as(i32p) for x in qw])
        scp = (f32p * e)(*[x.ctypes.data_as(f32p) for x in sc])
        qzp = (i32p * e)(*[x.ctypes.data_as(i32p) for x in qz])
        gi = None if g_idx is None else _np_i32(g_idx)
        h = C.c_void_p()
        rc = lib.ferrum_hip_gptq_load_stacked(C.byref(h), qwp, scp, qzp, None if gi is None else gi.ctypes.data_as(i32p),
                                              bits, group_size, k, n_per_expert, e, int(fuse_gate_up))
        _check(rc, "load_gptq_stacked")
        return ExpertStack(h, k, n_per_expert, e, fuse_gate_up)

    # ── BackendMoeFused (capabilities.rs:305-724) ────────────────────────────
    @staticmethod
    def route_topk_softmax(ctx, logits, expert_ids, expert_weights, tokens, num_experts, top_k, norm_topk_prob):
        import torch
        fn = (ctx.lib.ferrum_hip_moe_route_topk_softmax_f32 if logits.dtype == torch.float32
              else ctx.lib.ferrum_hip_moe_route_topk_softmax_f16)
        _check(fn(_ptr(logits), _ptr(expert_ids), _ptr(expert_weights), tokens, num_experts, top_k, int(norm_topk_prob),
                  ctx.stream), "route_topk_softmax")

    @staticmethod
    def moe_align_block_size_pair_ids(ctx, expert_ids, sorted_token_ids, block_ids, total_post_pad, batch_x_topk,
                                      num_experts, block_size, sorted_max):
        _check(ctx.lib.ferrum_hip_moe_align_block_size(_ptr(expert_ids), _ptr(sorted_token_ids), _ptr(block_ids),
                                                       _ptr(total_post_pad), batch_x_topk, num_experts, block_size,
                                                       sorted_max, ctx.stream), "moe_align_block_size")

    @staticmethod
    def moe_align_block_size(ctx, expert_ids, sorted_token_ids, block_ids, total_post_pad, batch_x_topk, num_experts,
                             block_size, sorted_max):
        """capabilities.rs:429 — sorted_token_ids hold unpadded packed rows."""
        _check(ctx.lib.ferrum_hip_moe_align_block_size_packed_rows(_ptr(expert_ids), _ptr(sorted_token_ids), _ptr(block_ids),
                                                                   _ptr(total_post_pad), batch_x_topk, num_experts,
                                                                   block_size, sorted_max, ctx.stream), "moe_align_block_size")

    @staticmethod
    def moe_build_pairs_by_token(ctx, expert_ids, pairs_by_token, packed_token_idx, expert_offsets, batch_x_topk,
                                 num_experts, top_k):
        _check(ctx.lib.ferrum_hip_moe_build_pairs_by_token(_ptr(expert_ids), _ptr(pairs_by_token), _ptr(packed_token_idx),
                                                           _ptr(expert_offsets), batch_x_topk, num_experts, top_k,
                                                           ctx.stream), "moe_build_pairs_by_token")

    @staticmethod
    def moe_combine_pairs(ctx, packed_down, pairs_by_token, pair_weights, out, batch, hidden, top_k, total_pairs):
        """BackendMoeFused::moe_combine with the trait's arguments (capabilities.rs:684)."""
        _check(ctx.lib.ferrum_hip_moe_combine_pairs_f16(_ptr(packed_down), _ptr(pairs_by_token), _ptr(pair_weights), _ptr(out),
                                                        batch, hidden, top_k, total_pairs, ctx.stream), "moe_combine")

    @staticmethod
    def weighted_sum_batched_offset(ctx, slots, weights, weights_offset, out, out_offset, batch, top_k, hidden):
        _check(ctx.lib.ferrum_hip_weighted_sum_batched_f16(_ptr(slots), _ptr(weights), C.c_size_t(weights_offset), _ptr(out),
                                                           C.c_size_t(out_offset), batch, top_k, hidden, ctx.stream),
               "weighted_sum_batched")

    @staticmethod
    def weighted_sum_batched(ctx, slots, weights, out, batch, top_k, hidden):
        HipBackend.weighted_sum_batched_offset(ctx, slots, weights, 0, out, 0, batch, top_k, hidden)

    @staticmethod
    def paged_decode_attention(ctx, q, k_pool, v_pool, out, block_tables, context_lens, num_seqs, num_heads, num_kv_heads,
                               head_dim, block_size, max_num_blocks_per_seq, q_len):
        _check(ctx.lib.ferrum_hip_paged_decode_attention_f16(
            _ptr(q), _ptr(k_pool), _ptr(v_pool), _ptr(out), _ptr(block_tables), _ptr(context_lens), num_seqs, num_heads,
            num_kv_heads, head_dim, block_size, max_num_blocks_per_seq, q_len, ctx.ws, ctx.stream), "paged_decode_attention")

    # ── BackendGraph (capabilities.rs:35-70) ─────────────────────────────────
    @staticmethod
    def begin_graph_capture(ctx):
        _check(ctx.lib.ferrum_hip_graph_begin_capture(ctx.stream), "begin_graph_capture")

    @staticmethod
    def end_graph_capture(ctx):
        g = C.c_void_p()
        _check(ctx.lib.ferrum_hip_graph_end_capture(ctx.stream, C.byref(g)), "end_graph_capture")
        return g

    @staticmethod
    def replay_graph(ctx, graph):
        _check(ctx.lib.ferrum_hip_graph_replay(graph, ctx.stream), "replay_graph")

    @staticmethod
    def reset_graph(ctx, graph):
        _check(ctx.lib.ferrum_hip_graph_destroy(graph), "reset_graph")

    @staticmethod
    def moe_combine(ctx, down, weights, out, tokens, top_k, hidden, accumulate=False):
        _check(ctx.lib.ferrum_hip_moe_combine_f16(_ptr(down), _ptr(weights), _ptr(out), tokens, top_k, hidden,
                                                  int(accumulate), ctx.stream), "moe_combine")


    # ── fused chains (csrc/fused.hip) ────────────────────────────────────────
    @staticmethod
    def fused_add_rms_norm_route(ctx, residual, x, w, eps, norm_out, router_w, num_experts, top_k, norm_topk_prob,
                                 expert_ids, expert_weights, logits_out, tokens, hidden):
        """fused_add_rms_norm → router gemm → route_topk_softmax in one launch."""
        _check(ctx.lib.ferrum_hip_fused_add_rms_norm_route_f16(
            _ptr(residual), _ptr(x), _ptr(w), C.c_float(eps), _ptr(norm_out), _ptr(router_w), num_experts, top_k,
            int(norm_topk_prob), _ptr(expert_ids), _ptr(expert_weights), _ptr(logits_out), tokens, hidden, ctx.stream),
            "fused_add_rms_norm_route")

    @staticmethod
    def fused_add_rms_norm_route_parts(ctx, residual_in, residual_out, x, x_slabs, num_slabs, slab_stride, ld_slab, w, eps,
                                       norm_out, router_w_tiled, num_experts, top_k, num_parts, cand, stats, logits_out,
                                       tokens, hidden):
        _check(ctx.lib.ferrum_hip_fused_add_rms_norm_route_parts_f16(
            _ptr(residual_in), _ptr(residual_out), _ptr(x), _ptr(x_slabs), num_slabs, C.c_long(slab_stride), ld_slab, _ptr(w),
            C.c_float(eps), _ptr(norm_out), _ptr(router_w_tiled), num_experts, top_k, num_parts, _ptr(cand), _ptr(stats),
            _ptr(logits_out), tokens, hidden, ctx.stream), "fused_add_rms_norm_route_parts")

    @staticmethod
    def fused_add_rms_norm_route_split(ctx, residual_in, residual_out, x, x_slabs, num_slabs, slab_stride, ld_slab, w, eps,
                                       norm_out, router_w_tiled, num_experts, top_k, norm_topk_prob, num_parts, cand, stats,
                                       arrive, expert_ids, expert_weights, logits_out, tokens, hidden):
        """Split-router form of `fused_add_rms_norm_route` (grid tokens × parts, merge inside the launch)."""
        _check(ctx.lib.ferrum_hip_fused_add_rms_norm_route_split_f16(
            _ptr(residual_in), _ptr(residual_out), _ptr(x), _ptr(x_slabs), num_slabs, C.c_long(slab_stride), ld_slab, _ptr(w),
            C.c_float(eps), _ptr(norm_out), _ptr(router_w_tiled), num_experts, top_k, norm_topk_prob, num_parts, _ptr(cand),
            _ptr(stats), _ptr(arrive), _ptr(expert_ids), _ptr(expert_weights), _ptr(logits_out), tokens, hidden, ctx.stream),
            "fused_add_rms_norm_route_split")

    @staticmethod
    def moe_combine_add_rms_norm(ctx, down, weights, residual, next_norm_w, eps, norm_out, tokens, top_k, hidden):
        """moe_combine → add_inplace → (next layer's) rms_norm in one launch."""
        _check(ctx.lib.ferrum_hip_moe_combine_add_rms_norm_f16(
            _ptr(down), _ptr(weights), _ptr(residual), _ptr(next_norm_w), C.c_float(eps), _ptr(norm_out), tokens, top_k,
            hidden, ctx.stream), "moe_combine_add_rms_norm")


class BlockAllocator:
    """Host `BlockAllocator` (ferrum-models/src/common/paged_pool.rs:106-365) via the C ABI."""

    def __init__(self, num_blocks):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.ferrum_hip_block_allocator_create(C.byref(self.h), C.c_uint32(num_blocks)), "block_allocator_create")

    def __del__(self):
        try:
            if self.h:
                self.lib.ferrum_hip_block_allocator_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def allocate(self):
        b = C.c_uint32()
        _check(self.lib.ferrum_hip_block_allocator_allocate(self.h, C.byref(b)), "allocate")
        return b.value

    def allocate_n(self, n):
        arr = (C.c_uint32 * max(n, 1))()
        _check(self.lib.ferrum_hip_block_allocator_allocate_n(self.h, C.c_uint32(n), arr), "allocate_n")
        return list(arr[:n])

    def free(self, blocks):
        arr = (C.c_uint32 * max(len(blocks), 1))(*blocks)
        _check(self.lib.ferrum_hip_block_allocator_free(self.h, arr, C.c_uint32(len(blocks))), "free")

    def acquire(self, block):
        _check(self.lib.ferrum_hip_block_allocator_acquire(self.h, C.c_uint32(block)), "acquire")

    def register_block_hash(self, block, h):
        _check(self.lib.ferrum_hip_block_allocator_register_hash(self.h, C.c_uint32(block), C.c_uint64(h)), "register_hash")

    def try_acquire_by_hash(self, h):
        out = C.c_int64()
        _check(self.lib.ferrum_hip_block_allocator_try_acquire_by_hash(self.h, C.c_uint64(h), C.byref(out)), "try_acquire")
        return None if out.value < 0 else out.value

    def free_count(self):
        return self.lib.ferrum_hip_block_allocator_free_count(self.h)

    def ref_count(self, b):
        return self.lib.ferrum_hip_block_allocator_ref_count(self.h, C.c_uint32(b))

    def peak_in_use(self):
        return self.lib.ferrum_hip_block_allocator_peak_in_use(self.h)

    def hash_table_size(self):
        return self.lib.ferrum_hip_block_allocator_hash_table_size(self.h)
