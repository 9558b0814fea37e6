//! Whole-model delegation: `ModelExecutor::{reserve_kv_slots, unified_decode, release}`
//! (ferrum-interfaces/src/model_executor.rs:456-651) forwarded to the C++ runner, which owns the fused launch chains and
//! the hipGraph decode loop.
use crate::{check, ffi};
use ferrum_interfaces::model_executor::{KvSlotRequest, KvSlotReservation, LogitsReturnPolicy, UnifiedBatch};
use ferrum_types::Result;
use std::os::raw::c_int;
use std::ptr;

pub struct HipModelExecutor { pub(crate) m: *mut ffi::FerrumHipModel }
unsafe impl Send for HipModelExecutor {}
impl Drop for HipModelExecutor { fn drop(&mut self) { unsafe { ffi::ferrum_hip_model_destroy(self.m) }; } }

impl HipModelExecutor {
    pub fn reserve_kv_slots(&mut self, reqs: &[KvSlotRequest]) -> Result<KvSlotReservation> {
        let c: Vec<ffi::FerrumHipKvSlotRequest> =
            reqs.iter().map(|r| ffi::FerrumHipKvSlotRequest { seq_id: r.seq_id, target_len: r.target_len as i32, _pad: 0 }).collect();
        let mut out = ffi::FerrumHipKvSlotReservation { block_size: 0, total_blocks: 0, free_blocks_before: 0, free_blocks_after: 0 };
        check(unsafe { ffi::ferrum_hip_model_reserve_kv_slots(self.m, c.as_ptr(), c.len() as c_int, &mut out) })?;
        Ok(KvSlotReservation { block_size: out.block_size as usize, total_blocks: out.total_blocks as usize,
                               free_blocks_before: out.free_blocks_before as usize, free_blocks_after: out.free_blocks_after as usize })
    }
    /// `LogitsReturnPolicy::GreedyArgmax` ↔ greedy = 1 (device argmax, ids only); `FullLogits` ↔ logits_out.
    pub fn unified_decode(&mut self, batch: &UnifiedBatch, policy: &LogitsReturnPolicy, out_tokens: &mut [u32], logits_out: Option<&mut [f32]>) -> Result<()> {
        let items: Vec<ffi::FerrumHipBatchItem> = batch.items.iter().map(|it| ffi::FerrumHipBatchItem {
            seq_id: it.seq_id, q_tokens: it.q_tokens.as_ptr(), num_q_tokens: it.q_tokens.len() as i32, pos_offset: it.pos_offset as i32,
            is_final_chunk: it.is_final_chunk as i32, _pad: 0 }).collect();
        let greedy = matches!(policy, LogitsReturnPolicy::GreedyArgmax { .. }) as c_int;
        check(unsafe {
            ffi::ferrum_hip_model_unified_forward_ex(self.m, items.as_ptr(), items.len() as c_int, greedy, ptr::null(), out_tokens.as_mut_ptr(),
                                                     logits_out.map_or(ptr::null_mut(), |l| l.as_mut_ptr()))
        })
    }
    pub fn release(&mut self, seq_id: u64) -> Result<()> { check(unsafe { ffi::ferrum_hip_model_release(self.m, seq_id) }) }
}
