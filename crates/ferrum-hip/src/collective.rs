//! `BackendCollective` (capabilities.rs:84-109): RCCL over xGMI, or the one-shot peer reduce for decode-sized messages.
use crate::{backend::{HipBackend, HipBuf, HipCtx}, ffi, must};
use ferrum_kernels::backend::{BackendCollective, ReduceOp};

impl BackendCollective for HipBackend {
    fn world_size(ctx: &HipCtx) -> usize { unsafe { ffi::ferrum_hip_comm_world_size(ctx.comm) as usize } }
    fn rank(ctx: &HipCtx) -> usize { unsafe { ffi::ferrum_hip_comm_rank(ctx.comm) as usize } }
    fn all_reduce(ctx: &mut HipCtx, buf: &mut HipBuf, len: usize, op: ReduceOp) {
        assert!(matches!(op, ReduceOp::Sum), "HipBackend::all_reduce: only Sum (the tensor-parallel decode path, tp_decode.rs:363-366)");
        must(unsafe { ffi::ferrum_hip_all_reduce_f16(ctx.comm, buf.ptr, len, ctx.stream) }, "all_reduce");
    }
    fn all_gather(ctx: &mut HipCtx, local: &HipBuf, global: &mut HipBuf, local_len: usize) {
        must(unsafe { ffi::ferrum_hip_all_gather_f16(ctx.comm, local.ptr, global.ptr, local_len, ctx.stream) }, "all_gather");
    }
    fn broadcast(ctx: &mut HipCtx, buf: &mut HipBuf, len: usize, src_rank: usize) {
        must(unsafe { ffi::ferrum_hip_broadcast_f16(ctx.comm, buf.ptr, len, src_rank as _, ctx.stream) }, "broadcast");
    }
}
