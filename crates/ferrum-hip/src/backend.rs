//! `Backend` + `BackendGraph` for `HipBackend` (crates/ferrum-kernels/src/backend/traits.rs:30-1600, capabilities.rs:35-70).
//!
//! Every REQUIRED method of the trait (traits.rs: `make_timer`, `new_context`, `sync`, `alloc_typed`, `from_slice_typed`,
//! `write_typed`, `gemm`, `rms_norm`, `fused_add_rms_norm`, `flash_attention`, `copy_slice`, `embedding_lookup`, `split_qkv`,
//! `fused_silu_mul_split`, `qk_norm_rope`, `kv_cache_append_head_major`, `transpose_head_to_token`, `add_inplace`, `add_bias`,
//! `layer_norm`, `gelu`, `alloc`, `to_vec`, `from_slice`) is implemented below over one C entry point each; the defaulted
//! methods the LLM decode path relies on are overridden with their device forms.  `tests/test_abi_symbols.py` parses the
//! required-method list out of the reference's `traits.rs` and asserts that each has an `fn` here.
use crate::{check, ffi, must};
use ferrum_kernels::backend::timer::BackendTimer;
use ferrum_kernels::backend::{AttnConfig, Backend, BackendGraph, Dtype, HostDtype, SrcDtype};
use ferrum_types::Result;
use half::f16;
use std::collections::HashMap;
use std::os::raw::{c_int, c_void};
use std::ptr;

/// Zero-sized backend selector, like `CudaBackend`.
pub struct HipBackend;

/// `B::Buffer`: device memory owned by the buffer, dtype-tagged like the CUDA lane's `CudaBuf`.
///
/// `f32_shadow`: buffers made by `from_slice` (norm weights, RoPE tables — small) also keep the caller's f32 values on the
/// device: the RoPE entry points take f32 cos / sin tables (llama_family.rs:5220-5237 computes them in f64 and stores f32;
/// the CUDA lane rounds them to f16, this backend does not have to).
pub struct HipBuf {
    pub(crate) ptr: *mut c_void,
    pub(crate) bytes: usize,
    pub(crate) dtype: Dtype,
    pub(crate) f32_shadow: *mut c_void,
}
unsafe impl Send for HipBuf {}
unsafe impl Sync for HipBuf {}
impl Drop for HipBuf {
    fn drop(&mut self) {
        unsafe {
            ffi::ferrum_hip_free(self.ptr);
            if !self.f32_shadow.is_null() {
                ffi::ferrum_hip_free(self.f32_shadow);
            }
        }
    }
}

fn dtype_bytes(d: Dtype) -> usize {
    match d {
        Dtype::F32 | Dtype::U32 | Dtype::I32 => 4,
        Dtype::F16 => 2,
        Dtype::I8 => 1,
    }
}

fn raw_alloc(bytes: usize) -> *mut c_void {
    let mut p = ptr::null_mut();
    must(unsafe { ffi::ferrum_hip_alloc(&mut p, bytes.max(16)) }, "alloc"); // zero-initialised, like B::alloc
    p
}

/// Blocking host → device copy on the default stream (weights, index tensors, test data: not the decode loop).
fn upload(dst: *mut c_void, src: *const c_void, bytes: usize) {
    if bytes == 0 {
        return;
    }
    must(unsafe { ffi::ferrum_hip_memcpy_h2d(dst, src, bytes, ptr::null_mut()) }, "memcpy_h2d");
    must(unsafe { ffi::ferrum_hip_stream_synchronize(ptr::null_mut()) }, "sync");
}

/// `B::Context`: stream + split-K / split-KV workspace + the graphs captured on it (keyed like `end_graph_capture(key)`)
/// + a small device scratch for host-side index slices (`embedding_lookup` takes `ids: &[u32]`).
pub struct HipCtx {
    pub(crate) stream: *mut c_void,
    pub(crate) ws: *mut ffi::FerrumHipWorkspace,
    pub(crate) graphs: HashMap<u64, *mut ffi::FerrumHipGraph>,
    pub(crate) comm: *mut ffi::FerrumHipComm,
    ids_scratch: *mut c_void,
    ids_scratch_bytes: usize,
    retired_scratch: Vec<*mut c_void>, // outgrown scratch a captured graph may still read: freed with the context
    capturing: bool,
}
unsafe impl Send for HipCtx {}
impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe {
            for (_, g) in self.graphs.drain() {
                ffi::ferrum_hip_graph_destroy(g);
            }
            ffi::ferrum_hip_stream_synchronize(self.stream);
            if !self.ids_scratch.is_null() {
                ffi::ferrum_hip_free(self.ids_scratch);
            }
            for p in self.retired_scratch.drain(..) {
                ffi::ferrum_hip_free(p);
            }
            ffi::ferrum_hip_workspace_destroy(self.ws);
            ffi::ferrum_hip_stream_destroy(self.stream);
        }
    }
}
impl HipCtx {
    fn stage_ids(&mut self, ids: &[u32]) -> *const u32 {
        let bytes = ids.len() * 4;
        if bytes > self.ids_scratch_bytes {
            if !self.ids_scratch.is_null() {
                self.retired_scratch.push(self.ids_scratch);
            }
            self.ids_scratch_bytes = bytes.next_power_of_two().max(4096);
            self.ids_scratch = raw_alloc(self.ids_scratch_bytes);
        }
        must(unsafe { ffi::ferrum_hip_memcpy_h2d(self.ids_scratch, ids.as_ptr() as *const c_void, bytes, self.stream) }, "memcpy_h2d");
        self.ids_scratch as *const u32
    }
}

/// `B::Timer` (backend/timer.rs:88-109): two device events recorded on the context's stream; `elapsed_ms` waits for the end event.
pub struct HipEventTimer {
    start: *mut c_void,
    end: *mut c_void,
    started: bool,
    ended: bool,
}
unsafe impl Send for HipEventTimer {}
impl Drop for HipEventTimer {
    fn drop(&mut self) {
        unsafe {
            ffi::ferrum_hip_event_destroy(self.start);
            ffi::ferrum_hip_event_destroy(self.end);
        }
    }
}
impl BackendTimer<HipBackend> for HipEventTimer {
    fn new() -> Self {
        let (mut start, mut end) = (ptr::null_mut(), ptr::null_mut());
        must(unsafe { ffi::ferrum_hip_event_create(&mut start) }, "event_create");
        must(unsafe { ffi::ferrum_hip_event_create(&mut end) }, "event_create");
        HipEventTimer { start, end, started: false, ended: false }
    }
    fn record_start(&mut self, ctx: &mut HipCtx) {
        must(unsafe { ffi::ferrum_hip_event_record(self.start, ctx.stream) }, "event_record");
        self.started = true;
        self.ended = false;
    }
    fn record_end(&mut self, ctx: &mut HipCtx) {
        must(unsafe { ffi::ferrum_hip_event_record(self.end, ctx.stream) }, "event_record");
        self.ended = true;
    }
    fn elapsed_ms(&self) -> f64 {
        if !(self.started && self.ended) {
            return 0.0;
        }
        let mut ms = 0.0f32;
        must(unsafe { ffi::ferrum_hip_event_elapsed_ms(self.start, self.end, &mut ms) }, "event_elapsed_ms");
        ms as f64
    }
}

impl Backend for HipBackend {
    type Buffer = HipBuf;
    type Context = HipCtx;
    type Timer = HipEventTimer;

    fn make_timer() -> HipEventTimer {
        <HipEventTimer as BackendTimer<HipBackend>>::new()
    }
    fn new_context() -> HipCtx {
        let mut stream = ptr::null_mut();
        let mut ws = ptr::null_mut();
        must(unsafe { ffi::ferrum_hip_stream_create(&mut stream) }, "stream_create");
        must(unsafe { ffi::ferrum_hip_workspace_create(&mut ws, 256 << 20) }, "workspace_create");
        HipCtx { stream, ws, graphs: HashMap::new(), comm: ptr::null_mut(), ids_scratch: ptr::null_mut(), ids_scratch_bytes: 0,
                 retired_scratch: Vec::new(), capturing: false }
    }
    fn sync(ctx: &mut HipCtx) {
        must(unsafe { ffi::ferrum_hip_stream_synchronize(ctx.stream) }, "sync");
    }
    fn graph_capture_in_flight(ctx: &HipCtx) -> bool {
        ctx.capturing
    }
    fn activation_elem_size_bytes() -> usize {
        2 // fp16 activations
    }

    // ── typed buffers (index tensors are U32 / I32 buffers, activations F16, fp32 scratch F32) ──
    fn alloc_typed(dtype: Dtype, n: usize) -> HipBuf {
        let bytes = n * dtype_bytes(dtype);
        HipBuf { ptr: raw_alloc(bytes), bytes, dtype, f32_shadow: ptr::null_mut() }
    }
    fn from_slice_typed<T: HostDtype>(data: &[T]) -> HipBuf {
        let bytes = std::mem::size_of_val(data);
        let buf = HipBuf { ptr: raw_alloc(bytes), bytes, dtype: T::DTYPE, f32_shadow: ptr::null_mut() };
        upload(buf.ptr, data.as_ptr() as *const c_void, bytes);
        buf
    }
    fn write_typed<T: HostDtype>(ctx: &mut HipCtx, dst: &mut HipBuf, data: &[T]) {
        let bytes = std::mem::size_of_val(data);
        assert!(bytes <= dst.bytes, "write_typed: {} bytes into a {}-byte buffer", bytes, dst.bytes);
        assert!(T::DTYPE == dst.dtype, "write_typed: dtype mismatch");
        // stream-ordered; the host slice must outlive the copy only until the call returns (pageable source: the runtime stages it)
        must(unsafe { ffi::ferrum_hip_memcpy_h2d(dst.ptr, data.as_ptr() as *const c_void, bytes, ctx.stream) }, "memcpy_h2d");
    }
    fn alloc(len: usize) -> HipBuf {
        Self::alloc_typed(Dtype::F16, len) // activation dtype, zero-initialised
    }
    fn zero_buffer(ctx: &mut HipCtx, buf: &mut HipBuf, len: usize) -> Result<()> {
        check(unsafe { ffi::ferrum_hip_memset_zero(buf.ptr, len * dtype_bytes(buf.dtype), ctx.stream) })
    }
    fn from_slice(data: &[f32]) -> HipBuf {
        let h: Vec<f16> = data.iter().map(|&v| f16::from_f32(v)).collect();
        let mut buf = Self::from_slice_typed::<f16>(&h);
        if data.len() <= (4 << 20) {
            buf.f32_shadow = raw_alloc(data.len() * 4);
            upload(buf.f32_shadow, data.as_ptr() as *const c_void, data.len() * 4);
        }
        buf
    }
    fn to_vec(buf: &HipBuf, len: usize) -> Vec<f32> {
        // (callers sync their context first: sync_before_host_readback — the copy below runs on the default stream)
        match buf.dtype {
            Dtype::F32 => {
                let mut out = vec![0f32; len];
                must(unsafe { ffi::ferrum_hip_memcpy_d2h(out.as_mut_ptr() as *mut c_void, buf.ptr, len * 4, ptr::null_mut()) }, "memcpy_d2h");
                must(unsafe { ffi::ferrum_hip_stream_synchronize(ptr::null_mut()) }, "sync");
                out
            }
            Dtype::F16 => {
                let mut h = vec![f16::ZERO; len];
                must(unsafe { ffi::ferrum_hip_memcpy_d2h(h.as_mut_ptr() as *mut c_void, buf.ptr, len * 2, ptr::null_mut()) }, "memcpy_d2h");
                must(unsafe { ffi::ferrum_hip_stream_synchronize(ptr::null_mut()) }, "sync");
                h.iter().map(|v| v.to_f32()).collect()
            }
            _ => panic!("to_vec: integer buffer"),
        }
    }
    fn sync_before_host_readback(ctx: &mut HipCtx) {
        Self::sync(ctx);
    }
    fn from_weight_bytes(raw: &[u8], src_dtype: SrcDtype) -> HipBuf {
        // checkpoint bytes → fp16 device weights (bf16 / f32 sources are rounded once on the host)
        let h: Vec<f16> = match src_dtype {
            SrcDtype::F16 => raw.chunks_exact(2).map(|b| f16::from_bits(u16::from_le_bytes([b[0], b[1]]))).collect(),
            SrcDtype::BF16 => raw.chunks_exact(2).map(|b| f16::from_f32(f32::from_bits((u16::from_le_bytes([b[0], b[1]]) as u32) << 16))).collect(),
            SrcDtype::F32 => raw.chunks_exact(4).map(|b| f16::from_f32(f32::from_le_bytes([b[0], b[1], b[2], b[3]]))).collect(),
        };
        Self::from_slice_typed::<f16>(&h)
    }

    // ── core ops ──
    fn gemm(ctx: &mut HipCtx, a: &HipBuf, b: &HipBuf, out: &mut HipBuf, m: usize, n: usize, k: usize) {
        must(unsafe { ffi::ferrum_hip_gemm_f16(a.ptr, b.ptr, out.ptr, m as c_int, n as c_int, k as c_int, ctx.ws, ctx.stream) }, "gemm");
    }
    fn rms_norm(ctx: &mut HipCtx, x: &HipBuf, w: &HipBuf, eps: f32, out: &mut HipBuf, tokens: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_rms_norm_f16(x.ptr, w.ptr, eps, out.ptr, tokens as c_int, dim as c_int, ctx.stream) }, "rms_norm");
    }
    fn fused_add_rms_norm(ctx: &mut HipCtx, residual: &mut HipBuf, x: &HipBuf, w: &HipBuf, eps: f32, out: &mut HipBuf, tokens: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_fused_add_rms_norm_f16(residual.ptr, x.ptr, w.ptr, eps, out.ptr, tokens as c_int, dim as c_int, ctx.stream) },
             "fused_add_rms_norm");
    }
    fn layer_norm(ctx: &mut HipCtx, x: &HipBuf, gamma: &HipBuf, beta: &HipBuf, eps: f32, out: &mut HipBuf, tokens: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_layer_norm_f16(x.ptr, gamma.ptr, beta.ptr, eps, out.ptr, tokens as c_int, dim as c_int, ctx.stream) }, "layer_norm");
    }
    fn gelu(ctx: &mut HipCtx, x: &HipBuf, out: &mut HipBuf, len: usize) {
        must(unsafe { ffi::ferrum_hip_gelu_f16(x.ptr, out.ptr, len, ctx.stream) }, "gelu");
    }
    fn flash_attention(ctx: &mut HipCtx, q: &HipBuf, k: &HipBuf, v: &HipBuf, out: &mut HipBuf, batch: usize, q_len: usize, kv_len: usize,
                       pos_offset: usize, cfg: &AttnConfig) {
        must(unsafe {
            ffi::ferrum_hip_flash_attention_f16(q.ptr, k.ptr, v.ptr, out.ptr, batch as c_int, q_len as c_int, kv_len as c_int, pos_offset as c_int,
                                                cfg.num_heads as c_int, cfg.num_kv_heads as c_int, cfg.head_dim as c_int, cfg.causal as c_int, cfg.scale,
                                                cfg.kv_seq_stride as c_int, cfg.sliding_window as c_int, ctx.stream)
        }, "flash_attention");
    }
    fn copy_slice(ctx: &mut HipCtx, src: &HipBuf, src_offset: usize, dst: &mut HipBuf, dst_offset: usize, len: usize) {
        must(unsafe { ffi::ferrum_hip_copy_slice_f16(src.ptr, src_offset, dst.ptr, dst_offset, len, ctx.stream) }, "copy_slice");
    }
    fn embedding_lookup(ctx: &mut HipCtx, table: &HipBuf, ids: &[u32], out: &mut HipBuf, dim: usize) {
        let dev_ids = ctx.stage_ids(ids);
        must(unsafe { ffi::ferrum_hip_embedding_lookup_f16(table.ptr, dev_ids, out.ptr, ids.len() as c_int, dim as c_int, ctx.stream) },
             "embedding_lookup");
    }
    fn embedding_lookup_dev(ctx: &mut HipCtx, table: &HipBuf, ids: &HipBuf, out: &mut HipBuf, batch: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_embedding_lookup_f16(table.ptr, ids.ptr as *const u32, out.ptr, batch as c_int, dim as c_int, ctx.stream) },
             "embedding_lookup_dev");
    }
    fn split_qkv(ctx: &mut HipCtx, qkv: &HipBuf, q: &mut HipBuf, k: &mut HipBuf, v: &mut HipBuf, tokens: usize, q_dim: usize, kv_dim: usize) {
        must(unsafe { ffi::ferrum_hip_split_qkv_f16(qkv.ptr, q.ptr, k.ptr, v.ptr, tokens as c_int, q_dim as c_int, kv_dim as c_int, ctx.stream) }, "split_qkv");
    }
    fn fused_silu_mul_split(ctx: &mut HipCtx, gate_up: &HipBuf, out: &mut HipBuf, tokens: usize, im: usize) {
        must(unsafe { ffi::ferrum_hip_fused_silu_mul_split_f16(gate_up.ptr, out.ptr, tokens as c_int, im as c_int, ctx.stream) }, "fused_silu_mul_split");
    }
    fn fused_gelu_tanh_mul_split(ctx: &mut HipCtx, gate_up: &HipBuf, out: &mut HipBuf, tokens: usize, im: usize) {
        must(unsafe { ffi::ferrum_hip_fused_gelu_tanh_mul_split_f16(gate_up.ptr, out.ptr, tokens as c_int, im as c_int, ctx.stream) },
             "fused_gelu_tanh_mul_split");
    }
    fn qk_norm_rope(ctx: &mut HipCtx, input: &HipBuf, norm_w: &HipBuf, cos: &HipBuf, sin: &HipBuf, output: &mut HipBuf, tokens: usize, heads: usize,
                    head_dim: usize, pos_offset: usize, eps: f32, mode: i32) {
        // the entry point reads f32 tables: the shadow `from_slice` kept (RoPE caches are built with it, llama_family.rs:5220-5237)
        assert!(!cos.f32_shadow.is_null() && !sin.f32_shadow.is_null(), "qk_norm_rope: cos / sin must come from Backend::from_slice");
        must(unsafe {
            ffi::ferrum_hip_qk_norm_rope_f16(input.ptr, norm_w.ptr, cos.f32_shadow as *const f32, sin.f32_shadow as *const f32, output.ptr, tokens as c_int,
                                             heads as c_int, head_dim as c_int, pos_offset as c_int, eps, mode as c_int, ctx.stream)
        }, "qk_norm_rope");
    }
    fn kv_cache_append_head_major(ctx: &mut HipCtx, cache_k: &mut HipBuf, cache_v: &mut HipBuf, cache_len: usize, cache_capacity: usize,
                                  new_k_head_major: &HipBuf, new_v_head_major: &HipBuf, new_tokens: usize, nkv: usize, hd: usize) {
        must(unsafe {
            ffi::ferrum_hip_kv_cache_append_head_major_f16(cache_k.ptr, cache_v.ptr, cache_len as c_int, cache_capacity as c_int, new_k_head_major.ptr,
                                                           new_v_head_major.ptr, new_tokens as c_int, nkv as c_int, hd as c_int, ctx.stream)
        }, "kv_cache_append_head_major");
    }
    fn transpose_head_to_token(ctx: &mut HipCtx, src: &HipBuf, dst: &mut HipBuf, tokens: usize, heads: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_transpose_head_to_token_f16(src.ptr, dst.ptr, tokens as c_int, heads as c_int, dim as c_int, ctx.stream) },
             "transpose_head_to_token");
    }
    fn transpose_token_to_head(ctx: &mut HipCtx, src: &HipBuf, dst: &mut HipBuf, tokens: usize, heads: usize, dim: usize) {
        must(unsafe { ffi::ferrum_hip_transpose_token_to_head_f16(src.ptr, dst.ptr, tokens as c_int, heads as c_int, dim as c_int, ctx.stream) },
             "transpose_token_to_head");
    }
    fn add_inplace(ctx: &mut HipCtx, residual: &mut HipBuf, x: &HipBuf, len: usize) {
        must(unsafe { ffi::ferrum_hip_add_inplace_f16(residual.ptr, x.ptr, len, ctx.stream) }, "add_inplace");
    }
    fn scaled_add_inplace(ctx: &mut HipCtx, dst: &mut HipBuf, src: &HipBuf, scale: f32, len: usize) {
        must(unsafe { ffi::ferrum_hip_scaled_add_inplace_f16(dst.ptr, src.ptr, scale, len, ctx.stream) }, "scaled_add_inplace");
    }
    fn add_bias(ctx: &mut HipCtx, data: &mut HipBuf, bias: &HipBuf, rows: usize, cols: usize) {
        must(unsafe { ffi::ferrum_hip_add_bias_f16(data.ptr, bias.ptr, rows as c_int, cols as c_int, ctx.stream) }, "add_bias");
    }
    fn scale_inplace(ctx: &mut HipCtx, buf: &mut HipBuf, scale: f32, len: usize) {
        must(unsafe { ffi::ferrum_hip_scale_inplace_f16(buf.ptr, scale, len, ctx.stream) }, "scale_inplace");
    }

    // ── device-side greedy sampling (traits.rs:1534-1591): first maximum, optional token mask / sparse repetition penalty ──
    fn argmax_rows_f16(ctx: &mut HipCtx, logits: &HipBuf, m: usize, n: usize) -> Result<Vec<u32>> {
        argmax_rows(ctx, logits, ptr::null(), 0, m, n)
    }
    fn argmax_rows_f16_masked(ctx: &mut HipCtx, logits: &HipBuf, valid_token_mask: &HipBuf, mask_len: usize, m: usize, n: usize) -> Result<Vec<u32>> {
        argmax_rows(ctx, logits, valid_token_mask.ptr as *const u8, mask_len, m, n)
    }
    fn argmax_rows_f16_sparse_repetition_penalty(ctx: &mut HipCtx, logits: &mut HipBuf, valid_token_mask: Option<(&HipBuf, usize)>, row_offsets: &HipBuf,
                                                 token_ids: &HipBuf, repetition_penalties: &HipBuf, _total_token_ids: usize, m: usize, n: usize) -> Result<Vec<u32>> {
        check(unsafe {
            ffi::ferrum_hip_apply_repetition_penalties_sparse_f16(logits.ptr, row_offsets.ptr as *const u32, token_ids.ptr as *const u32,
                                                                  repetition_penalties.ptr as *const f32, m as c_int, n as c_int, ctx.stream)
        })?;
        let (mask, mask_len) = valid_token_mask.map_or((ptr::null(), 0), |(b, l)| (b.ptr as *const u8, l));
        argmax_rows(ctx, logits, mask, mask_len, m, n)
    }
}

fn argmax_rows(ctx: &mut HipCtx, logits: &HipBuf, mask: *const u8, mask_len: usize, m: usize, n: usize) -> Result<Vec<u32>> {
    let ids = HipBackend::alloc_typed(Dtype::U32, m);
    check(unsafe { ffi::ferrum_hip_argmax_rows_f16(logits.ptr, ids.ptr as *mut u32, mask, mask_len as c_int, m as c_int, n as c_int, ctx.stream) })?;
    let mut out = vec![0u32; m];
    check(unsafe { ffi::ferrum_hip_memcpy_d2h(out.as_mut_ptr() as *mut c_void, ids.ptr, m * 4, ctx.stream) })?;
    check(unsafe { ffi::ferrum_hip_stream_synchronize(ctx.stream) })?;
    Ok(out)
}

impl BackendGraph for HipBackend {
    fn begin_graph_capture(ctx: &mut HipCtx) -> Result<()> {
        check(unsafe { ffi::ferrum_hip_graph_begin_capture(ctx.stream) })?;
        ctx.capturing = true;
        Ok(())
    }
    fn end_graph_capture(ctx: &mut HipCtx, key: u64) -> Result<()> {
        let mut g = ptr::null_mut();
        ctx.capturing = false;
        check(unsafe { ffi::ferrum_hip_graph_end_capture(ctx.stream, &mut g) })?;
        if let Some(old) = ctx.graphs.insert(key, g) {
            unsafe { ffi::ferrum_hip_graph_destroy(old) };
        }
        Ok(())
    }
    fn replay_graph(ctx: &mut HipCtx, key: u64) -> Result<bool> {
        match ctx.graphs.get(&key) {
            Some(&g) => check(unsafe { ffi::ferrum_hip_graph_replay(g, ctx.stream) }).map(|_| true),
            None => Ok(false),
        }
    }
    fn reset_graph(ctx: &mut HipCtx, key: u64) {
        if let Some(g) = ctx.graphs.remove(&key) {
            unsafe { ffi::ferrum_hip_graph_destroy(g) };
        }
    }
    fn reset_all_graphs(ctx: &mut HipCtx) {
        for (_, g) in ctx.graphs.drain() {
            unsafe { ffi::ferrum_hip_graph_destroy(g) };
        }
    }
}
