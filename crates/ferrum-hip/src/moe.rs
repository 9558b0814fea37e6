//! `BackendMoeFused` (capabilities.rs:305-724): routing, bucket plan, combine — all device-side, graph-capturable.
use crate::{backend::{HipBackend, HipBuf, HipCtx}, check, ffi, must};
use ferrum_kernels::backend::BackendMoeFused;
use ferrum_types::Result;
use std::os::raw::c_int;

impl BackendMoeFused for HipBackend {
    // capabilities.rs:334
    fn route_topk_softmax(ctx: &mut HipCtx, logits: &HipBuf, expert_ids: &mut HipBuf, expert_weights: &mut HipBuf, batch: usize, num_experts: usize,
                          top_k: usize, norm_topk_prob: bool) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_moe_route_topk_softmax_f16(logits.ptr, expert_ids.ptr as *mut i32, expert_weights.ptr as *mut f32, batch as c_int,
                                                       num_experts as c_int, top_k as c_int, norm_topk_prob as c_int, ctx.stream)
        })
    }
    // capabilities.rs:410 — stable counting sort = MoeBucketPlan::rebuild_into, bit for bit
    fn moe_build_pairs_by_token(ctx: &mut HipCtx, expert_ids: &HipBuf, pairs_by_token: &mut HipBuf, packed_token_idx: &mut HipBuf,
                                expert_offsets: &mut HipBuf, batch_x_topk: usize, num_experts: usize, top_k: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_moe_build_pairs_by_token(expert_ids.ptr as *const i32, pairs_by_token.ptr as *mut i32, packed_token_idx.ptr as *mut i32,
                                                     expert_offsets.ptr as *mut i32, batch_x_topk as c_int, num_experts as c_int, top_k as c_int, ctx.stream)
        })
    }
    // capabilities.rs:429 (packed rows) and :449 (pair ids)
    fn moe_align_block_size(ctx: &mut HipCtx, ids: &HipBuf, sorted: &mut HipBuf, block_ids: &mut HipBuf, total: &mut HipBuf, batch_x_topk: usize,
                            num_experts: usize, block_size: usize, sorted_max_size: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_moe_align_block_size_packed_rows(ids.ptr as *const i32, sorted.ptr as *mut i32, block_ids.ptr as *mut i32, total.ptr as *mut i32,
                                                             batch_x_topk as c_int, num_experts as c_int, block_size as c_int, sorted_max_size as c_int, ctx.stream)
        })
    }
    fn moe_align_block_size_pair_ids(ctx: &mut HipCtx, ids: &HipBuf, sorted: &mut HipBuf, block_ids: &mut HipBuf, total: &mut HipBuf,
                                     batch_x_topk: usize, num_experts: usize, block_size: usize, sorted_max_size: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_moe_align_block_size(ids.ptr as *const i32, sorted.ptr as *mut i32, block_ids.ptr as *mut i32, total.ptr as *mut i32,
                                                 batch_x_topk as c_int, num_experts as c_int, block_size as c_int, sorted_max_size as c_int, ctx.stream)
        })
    }
    // capabilities.rs:560,580
    fn weighted_sum_batched(ctx: &mut HipCtx, slots: &HipBuf, weights: &HipBuf, out: &mut HipBuf, batch: usize, top_k: usize, hidden: usize) -> Result<()> {
        Self::weighted_sum_batched_offset(ctx, slots, weights, 0, out, 0, batch, top_k, hidden)
    }
    fn weighted_sum_batched_offset(ctx: &mut HipCtx, slots: &HipBuf, weights: &HipBuf, weights_offset: usize, out: &mut HipBuf, out_offset: usize,
                                   batch: usize, top_k: usize, hidden: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_weighted_sum_batched_f16(slots.ptr, weights.ptr as *const f32, weights_offset, out.ptr, out_offset, batch as c_int,
                                                     top_k as c_int, hidden as c_int, ctx.stream)
        })
    }
    // capabilities.rs:684 (infallible in the trait)
    fn moe_combine(ctx: &mut HipCtx, packed_down: &HipBuf, pairs_by_token: &HipBuf, pair_weights: &HipBuf, out: &mut HipBuf, batch: usize,
                   hidden: usize, top_k: usize, total_pairs: usize) {
        must(unsafe {
            ffi::ferrum_hip_moe_combine_pairs_f16(packed_down.ptr, pairs_by_token.ptr as *const i32, pair_weights.ptr as *const f32, out.ptr,
                                                  batch as c_int, hidden as c_int, top_k as c_int, total_pairs as c_int, ctx.stream)
        }, "moe_combine");
    }
}
