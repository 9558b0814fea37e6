//! `BackendQuantMarlin` (capabilities.rs:121-193): GPTQ-INT4 weights handed over as host slices, owned by the returned object.
use crate::{backend::{HipBackend, HipBuf, HipCtx}, check, ffi, must};
use ferrum_kernels::{backend::BackendQuantMarlin, linear::Linear, marlin_expert_stack::MarlinExpertStack};
use ferrum_types::Result;
use std::os::raw::c_int;
use std::ptr;
use std::sync::Arc;

pub struct HipGptqLinear { pub(crate) h: *mut ffi::FerrumHipGptq, k: usize, n: usize }
unsafe impl Send for HipGptqLinear {}
unsafe impl Sync for HipGptqLinear {}
impl Drop for HipGptqLinear { fn drop(&mut self) { unsafe { ffi::ferrum_hip_gptq_free(self.h) }; } }

impl Linear<HipBackend> for HipGptqLinear {
    fn in_features(&self) -> usize { self.k }
    fn out_features(&self) -> usize { self.n }
    fn forward(&self, ctx: &mut HipCtx, input: &HipBuf, out: &mut HipBuf, m: usize) {
        must(unsafe { ffi::ferrum_hip_gptq_linear_forward_f16(self.h, input.ptr, out.ptr, m as c_int, ctx.ws, ctx.stream) }, "gptq_linear_forward");
    }
}

pub struct HipExpertStack { pub(crate) h: *mut ffi::FerrumHipGptq, k: usize, n: usize, e: usize, pub(crate) fused_silu: bool }
unsafe impl Send for HipExpertStack {}
unsafe impl Sync for HipExpertStack {}
impl Drop for HipExpertStack { fn drop(&mut self) { unsafe { ffi::ferrum_hip_gptq_free(self.h) }; } }

impl BackendQuantMarlin for HipBackend {
    fn load_gptq(qweight: &[i32], scales: &[f32], qzeros: &[i32], g_idx: Option<&[i32]>, bias: Option<&[f32]>, bits: u32, group_size: usize,
                 k: usize, n: usize) -> Result<Box<dyn Linear<Self> + Send + Sync>> {
        let mut h = ptr::null_mut();
        check(unsafe {
            ffi::ferrum_hip_gptq_load(&mut h, qweight.as_ptr(), scales.as_ptr(), qzeros.as_ptr(), g_idx.map_or(ptr::null(), |g| g.as_ptr()),
                                      bias.map_or(ptr::null(), |b| b.as_ptr()), bits as c_int, group_size as c_int, k as c_int, n as c_int)
        })?;
        Ok(Box::new(HipGptqLinear { h, k, n }))
    }
    fn load_gptq_stacked(qweights: &[&[i32]], scales: &[&[f32]], qzeros: &[&[i32]], g_idx: Option<&[i32]>, bits: u32, group_size: usize, k: usize,
                         n_per_expert: usize) -> Result<Arc<dyn MarlinExpertStack<Self>>> {
        let (qw, sc, qz): (Vec<_>, Vec<_>, Vec<_>) = (qweights.iter().map(|s| s.as_ptr()).collect(), scales.iter().map(|s| s.as_ptr()).collect(),
                                                       qzeros.iter().map(|s| s.as_ptr()).collect());
        let mut h = ptr::null_mut();
        check(unsafe {
            ffi::ferrum_hip_gptq_load_stacked(&mut h, qw.as_ptr(), sc.as_ptr(), qz.as_ptr(), g_idx.map_or(ptr::null(), |g| g.as_ptr()), bits as c_int,
                                              group_size as c_int, k as c_int, n_per_expert as c_int, qweights.len() as c_int, 0)
        })?;
        Ok(Arc::new(HipExpertStack { h, k, n: n_per_expert, e: qweights.len(), fused_silu: false }))
    }
}

impl MarlinExpertStack<HipBackend> for HipExpertStack {
    fn n_per_expert(&self) -> usize { self.n }
    fn k(&self) -> usize { self.k }
    fn num_experts(&self) -> usize { self.e }
    fn requires_vllm_moe(&self) -> bool { false }
    fn as_any(&self) -> &dyn std::any::Any { self }
    fn zero_workspace(&self, _ctx: &mut HipCtx) -> Result<()> { Ok(()) }       // no lock workspace: outputs are written, not atomically added
    // marlin_expert_stack.rs:63
    fn gemm_phase_batched(&self, ctx: &mut HipCtx, input: &HipBuf, dispatches: &[(usize, usize, usize, usize)], output: &mut HipBuf, k: usize) -> Result<()> {
        let flat: Vec<i32> = dispatches.iter().flat_map(|&(e, i, o, m)| [e as i32, i as i32, o as i32, m as i32]).collect();
        check(unsafe {
            ffi::ferrum_hip_moe_gemm_phase_batched_f16(self.h, input.ptr, flat.as_ptr(), dispatches.len() as c_int, output.ptr, k as c_int,
                                                       self.fused_silu as c_int, ctx.stream)
        })
    }
    // marlin_expert_stack.rs:86
    fn gemm_phase_vllm(&self, ctx: &mut HipCtx, input: &HipBuf, sorted_token_ids: &HipBuf, expert_ids: &HipBuf, num_tokens_past_padded: &HipBuf,
                       output: &mut HipBuf, prob_m: usize, moe_block_size: usize, top_k: usize) -> Result<()> {
        let max_blocks = prob_m * top_k / moe_block_size + self.e.min(prob_m * top_k);
        check(unsafe {
            ffi::ferrum_hip_moe_gemm_phase_f16(self.h, input.ptr, sorted_token_ids.ptr as *const i32, expert_ids.ptr as *const i32,
                                               num_tokens_past_padded.ptr as *const i32, output.ptr, prob_m as c_int, moe_block_size as c_int,
                                               top_k as c_int, max_blocks as c_int, self.fused_silu as c_int, ctx.stream)
        })
    }
    // make_expert_linear: a column slice of one expert is a plain GPTQ linear over the same tiles (not needed by the Qwen3-MoE path)
}
