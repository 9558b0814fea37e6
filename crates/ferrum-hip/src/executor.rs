//! Whole-model delegation.
//!
//! The reference's engine talks to a `ModelExecutor` (ferrum-interfaces/src/model_executor.rs:456-651); its own implementation
//! of that trait, `LlmExecutor` (ferrum-models/src/executor/llm_executor.rs:187-204), is generic over a boxed
//! `DecoderOnlyLLM` (ferrum-models/src/common/llm.rs:45-293) and already carries the admission / fallback / profiling
//! logic.  The drop-in seam is therefore `DecoderOnlyLLM`: `HipDecoderModel` implements it over the C++ runner
//! (`ferrum_hip_model_*`: fused launch chains, hipGraph decode loop, block allocator), and `HipModelExecutor` is
//! `LlmExecutor` around it with every `ModelExecutor` method forwarded — so `ferrum serve` keeps its executor semantics
//! bit for bit and only the model underneath changes.
use crate::{check, ffi, last_error};
use async_trait::async_trait;
use ferrum_interfaces::model_executor::{
    DecodeInput, DecodeOutput, ExecutorCapabilities, ExecutorStatus, KvSlotAllocation, KvSlotCapacitySnapshot, KvSlotRequest, KvSlotReservation,
    LogitsReturnPolicy, ModelExecutor, PrefillInput, PrefillOutput, UnifiedBatch,
};
use ferrum_interfaces::{KvCacheHandle, TensorRef};
use ferrum_models::common::llm::{DecoderOnlyLLM, LlmRuntimeConfig};
use ferrum_models::executor::llm_executor::LlmExecutor;
use ferrum_types::{FerrumError, ModelInfo, Result};
use std::collections::HashMap;
use std::os::raw::c_int;
use std::ptr;
use std::sync::Arc;

/// `DecoderOnlyLLM` over the C++ runner.  Cache ids are strings in the reference and u64 in the C ABI: the map below owns the
/// translation (ids are never reused while a sequence lives).
pub struct HipDecoderModel {
    pub(crate) m: *mut ffi::FerrumHipModel,
    cfg: LlmRuntimeConfig,
    ids: HashMap<String, u64>,
    lens: HashMap<String, usize>,     // tokens the runner holds per cache (positions are explicit in the trait; kept for truncate / asserts)
    next_id: u64,
}
unsafe impl Send for HipDecoderModel {}
unsafe impl Sync for HipDecoderModel {}
impl Drop for HipDecoderModel {
    fn drop(&mut self) {
        unsafe { ffi::ferrum_hip_model_destroy(self.m) };
    }
}

impl HipDecoderModel {
    /// `m`: a finalized runner (weights handed over through `ferrum_hip_model_set_*` or `ferrum_hip_model_load_checkpoint`).
    pub fn from_raw(m: *mut ffi::FerrumHipModel, cfg: LlmRuntimeConfig) -> Self {
        HipDecoderModel { m, cfg, ids: HashMap::new(), lens: HashMap::new(), next_id: 1 }
    }
    fn id_of(&mut self, cache_id: &str) -> u64 {
        if let Some(&v) = self.ids.get(cache_id) {
            return v;
        }
        let v = self.next_id;
        self.next_id += 1;
        self.ids.insert(cache_id.to_string(), v);
        v
    }
    fn local_vocab(&self) -> usize {
        let (mut v0, mut n) = (0 as c_int, 0 as c_int);
        unsafe { ffi::ferrum_hip_model_local_vocab(self.m, &mut v0, &mut n) };
        n as usize
    }

    /// One forward over `items` = (cache id, tokens, pos_offset, is_final_chunk).  `policies[i]` of the FINAL items decides what
    /// comes back: `FullLogits` → the row's logits; `GreedyArgmax { token_mask, repetition_penalty }` → the reference's greedy
    /// sentinel `vec![token_id as f32]` (qwen3_moe_forward_unified.rs:407), sampled on the device with the mask and the sparse
    /// penalty applied there (`FerrumHipGreedyOptions`).  A batch mixes the two only through FullLogits for everybody (the
    /// caller's `force_full_logits`, llm_executor.rs:938-941).
    fn forward(&mut self, items: &[(String, Vec<u32>, usize, bool)], policies: Option<&[LogitsReturnPolicy]>) -> Result<Vec<Option<Vec<f32>>>> {
        let c_items: Vec<ffi::FerrumHipBatchItem> = items.iter().map(|(cid, toks, pos, fin)| ffi::FerrumHipBatchItem {
            seq_id: self.id_of(cid), q_tokens: toks.as_ptr(), num_q_tokens: toks.len() as i32, pos_offset: *pos as i32, is_final_chunk: *fin as i32, _pad: 0,
        }).collect();
        let finals: Vec<usize> = items.iter().enumerate().filter(|(_, it)| it.3).map(|(i, _)| i).collect();
        let greedy = policies.map_or(false, |p| !finals.is_empty() && finals.iter().all(|&i| matches!(p[i], LogitsReturnPolicy::GreedyArgmax { .. })));
        let vocab = self.local_vocab();
        let mut out_tokens = vec![0u32; finals.len().max(1)];
        let mut logits = if greedy { Vec::new() } else { vec![0f32; finals.len().max(1) * vocab] };
        // GreedyArgmax options: ONE mask shared by the batch (the engine hands the same Arc to every row, model_executor.rs:66-98)
        // and the per-row sparse repetition penalties in CSR form
        let mut mask_bytes: Option<(u64, Arc<[i8]>)> = None;     // (fingerprint, bytes): `valid_token_mask[id] != 0` ⇔ id may be selected
        let (mut row_offsets, mut token_ids, mut penalties) = (vec![0u32], Vec::<u32>::new(), Vec::<f32>::new());
        let mut any_penalty = false;
        if greedy {
            let pol = policies.unwrap();
            for &i in &finals {
                if let LogitsReturnPolicy::GreedyArgmax { token_mask, repetition_penalty } = &pol[i] {
                    if let Some(mk) = token_mask {
                        match &mask_bytes {
                            None => mask_bytes = Some((mk.fingerprint, mk.valid_token_mask.clone())),
                            Some((fp, prev)) if *fp == mk.fingerprint && prev.len() == mk.valid_token_mask.len() => {}
                            Some(_) => return Err(FerrumError::unsupported("unified_forward: rows of one batch carry different token masks")),
                        }
                    } else if mask_bytes.is_some() {
                        return Err(FerrumError::unsupported("unified_forward: rows of one batch carry different token masks"));
                    }
                    match repetition_penalty {
                        Some(rp) if !rp.is_empty() => {
                            token_ids.extend_from_slice(&rp.token_ids);
                            penalties.push(rp.penalty);
                            any_penalty = true;
                        }
                        _ => penalties.push(1.0),
                    }
                    row_offsets.push(token_ids.len() as u32);
                }
            }
        }
        let opts = ffi::FerrumHipGreedyOptions {
            valid_token_mask: mask_bytes.as_ref().map_or(ptr::null(), |(_, m)| m.as_ptr() as *const u8),
            mask_len: mask_bytes.as_ref().map_or(0, |(_, m)| m.len() as c_int),
            _pad: 0,
            penalty_row_offsets: if any_penalty { row_offsets.as_ptr() } else { ptr::null() },
            penalty_token_ids: if any_penalty { token_ids.as_ptr() } else { ptr::null() },
            penalties: if any_penalty { penalties.as_ptr() } else { ptr::null() },
        };
        let use_opts = greedy && (mask_bytes.is_some() || any_penalty);
        assert!(out_tokens.len() >= finals.len() && (greedy || logits.len() >= finals.len() * vocab));
        check(unsafe {
            ffi::ferrum_hip_model_unified_forward_ex(self.m, c_items.as_ptr(), c_items.len() as c_int, greedy as c_int,
                                                     if use_opts { &opts } else { ptr::null() }, out_tokens.as_mut_ptr(),
                                                     if greedy { ptr::null_mut() } else { logits.as_mut_ptr() })
        })?;
        for (cid, toks, pos, _) in items {
            self.lens.insert(cid.clone(), pos + toks.len());
        }
        let mut res: Vec<Option<Vec<f32>>> = vec![None; items.len()];
        for (j, &i) in finals.iter().enumerate() {
            res[i] = Some(if greedy { vec![out_tokens[j] as f32] } else { logits[j * vocab..(j + 1) * vocab].to_vec() });
        }
        Ok(res)
    }
}

impl DecoderOnlyLLM for HipDecoderModel {
    fn config(&self) -> &LlmRuntimeConfig {
        &self.cfg
    }
    // common/llm.rs:133 — atomic over the batch: nothing is taken when one request does not fit (paged_pool.rs:416-442)
    fn reserve_kv_slots(&mut self, requests: &[KvSlotRequest]) -> std::result::Result<Option<KvSlotReservation>, FerrumError> {
        let mut before = Vec::with_capacity(requests.len());
        let c: Vec<ffi::FerrumHipKvSlotRequest> = requests.iter().map(|r| {
            let id = self.id_of(&r.cache_id);
            before.push(self.blocks_of(id));
            ffi::FerrumHipKvSlotRequest { seq_id: id, target_len: r.target_len as i32, _pad: 0 }
        }).collect();
        let mut out = ffi::FerrumHipKvSlotReservation { block_size: 0, total_blocks: 0, free_blocks_before: 0, free_blocks_after: 0 };
        check(unsafe { ffi::ferrum_hip_model_reserve_kv_slots(self.m, c.as_ptr(), c.len() as c_int, &mut out) })?;
        let allocations = requests.iter().zip(c.iter()).zip(before.iter()).map(|((r, cr), &b)| {
            let after = self.blocks_of(cr.seq_id);
            KvSlotAllocation { cache_id: r.cache_id.clone(), blocks_before: b, blocks_after: after, new_blocks: after - b }
        }).collect();
        Ok(Some(KvSlotReservation { block_size: out.block_size as usize, total_blocks: out.total_blocks as usize,
                                    free_blocks_before: out.free_blocks_before as usize, free_blocks_after: out.free_blocks_after as usize, allocations }))
    }
    fn kv_slot_capacity_snapshot(&self) -> Option<KvSlotCapacitySnapshot> {
        let mut out = ffi::FerrumHipKvSlotReservation { block_size: 0, total_blocks: 0, free_blocks_before: 0, free_blocks_after: 0 };
        if unsafe { ffi::ferrum_hip_model_kv_capacity_snapshot(self.m, &mut out) } != ffi::FERRUM_HIP_OK {
            return None;
        }
        Some(KvSlotCapacitySnapshot { block_size: out.block_size as usize, total_blocks: out.total_blocks as usize, free_blocks: out.free_blocks_after as usize })
    }
    // common/llm.rs:160,165 — the infallible single-sequence forms panic on misuse like the other backends
    fn prefill(&mut self, cache_id: &str, tokens: &[u32]) -> Vec<f32> {
        let items = [(cache_id.to_string(), tokens.to_vec(), 0usize, true)];
        self.forward(&items, None).unwrap_or_else(|e| panic!("prefill: {e}")).pop().unwrap().unwrap()
    }
    fn decode(&mut self, cache_id: &str, token: u32, pos: u32) -> Vec<f32> {
        let items = [(cache_id.to_string(), vec![token], pos as usize, true)];
        self.forward(&items, None).unwrap_or_else(|e| panic!("decode: {e}")).pop().unwrap().unwrap()
    }
    fn decode_batch(&mut self, batch: &[(String, u32, u32)]) -> Vec<Vec<f32>> {
        let items: Vec<_> = batch.iter().map(|(c, t, p)| (c.clone(), vec![*t], *p as usize, true)).collect();
        self.forward(&items, None).unwrap_or_else(|e| panic!("decode_batch: {e}")).into_iter().map(|o| o.unwrap()).collect()
    }
    fn decode_batch_with_logits_policy(&mut self, batch: &[(String, u32, u32)], policies: &[LogitsReturnPolicy]) -> Vec<Vec<f32>> {
        let items: Vec<_> = batch.iter().map(|(c, t, p)| (c.clone(), vec![*t], *p as usize, true)).collect();
        self.forward(&items, Some(policies)).unwrap_or_else(|e| panic!("decode_batch: {e}")).into_iter().map(|o| o.unwrap()).collect()
    }
    // common/llm.rs:244,260
    fn unified_forward(&mut self, items: &[(String, Vec<u32>, usize, bool)]) -> std::result::Result<Vec<Option<Vec<f32>>>, FerrumError> {
        self.forward(items, None)
    }
    fn unified_forward_with_logits_policy(&mut self, items: &[(String, Vec<u32>, usize, bool)], policies: &[LogitsReturnPolicy])
                                          -> std::result::Result<Vec<Option<Vec<f32>>>, FerrumError> {
        if policies.len() != items.len() {
            return Err(FerrumError::model(format!("unified_forward: {} policies for {} items", policies.len(), items.len())));
        }
        self.forward(items, Some(policies))
    }
    fn unified_forward_can_return_full_logits(&self) -> bool {
        true
    }
    fn release(&mut self, cache_id: &str) {
        if let Some(id) = self.ids.remove(cache_id) {
            self.lens.remove(cache_id);
            if unsafe { ffi::ferrum_hip_model_release(self.m, id) } != ffi::FERRUM_HIP_OK {
                panic!("release: {}", last_error());
            }
        }
    }
    fn reset(&mut self) {
        let ids: Vec<String> = self.ids.keys().cloned().collect();
        for c in ids {
            self.release(&c);
        }
    }
}

impl HipDecoderModel {
    fn blocks_of(&self, id: u64) -> usize {
        let (mut nb, mut kv) = (0 as c_int, 0 as c_int);
        // (unknown sequence → 0 blocks; the entry point reports the count without copying when capacity is 0)
        if unsafe { ffi::ferrum_hip_model_block_table(self.m, id, ptr::null_mut(), 0, &mut nb, &mut kv) } != ffi::FERRUM_HIP_OK {
            return 0;
        }
        nb as usize
    }
}

/// `ModelExecutor` (model_executor.rs:456-651) — the reference's `LlmExecutor` around `HipDecoderModel`; every method is
/// forwarded, so admission (`reserve_kv_slots`), the unified mixed batch (`unified_decode`), the legacy `prefill` / `decode`
/// pair and the capability / status reports behave exactly as they do for the CUDA lane.
pub struct HipModelExecutor {
    inner: LlmExecutor,
}

impl HipModelExecutor {
    pub fn new(model: HipDecoderModel, info: ModelInfo) -> Self {
        HipModelExecutor { inner: LlmExecutor::new(Box::new(model), info) }
    }
}

#[async_trait]
impl ModelExecutor for HipModelExecutor {
    fn info(&self) -> &ModelInfo {
        self.inner.info()
    }
    fn supports_native_unified_decode(&self) -> bool {
        true // one forward over the mixed batch: ferrum_hip_model_unified_forward_ex
    }
    fn kv_capacity(&self) -> Option<usize> {
        self.inner.kv_capacity()
    }
    fn reserve_kv_slots(&self, requests: &[KvSlotRequest]) -> Result<Option<KvSlotReservation>> {
        self.inner.reserve_kv_slots(requests)
    }
    fn kv_slot_capacity_snapshot(&self) -> Option<KvSlotCapacitySnapshot> {
        self.inner.kv_slot_capacity_snapshot()
    }
    async fn prefill(&self, input: &PrefillInput) -> Result<PrefillOutput> {
        self.inner.prefill(input).await
    }
    async fn batch_prefill(&self, inputs: &[PrefillInput]) -> Result<Vec<PrefillOutput>> {
        self.inner.batch_prefill(inputs).await
    }
    async fn decode(&self, input: &DecodeInput) -> Result<DecodeOutput> {
        self.inner.decode(input).await
    }
    async fn batch_decode(&self, inputs: &[DecodeInput]) -> Result<Vec<DecodeOutput>> {
        self.inner.batch_decode(inputs).await
    }
    async fn unified_decode(&self, batch: &UnifiedBatch) -> Result<Vec<Option<Vec<f32>>>> {
        self.inner.unified_decode(batch).await
    }
    async fn forward(&self, input: &TensorRef) -> Result<TensorRef> {
        self.inner.forward(input).await
    }
    async fn truncate_kv(&self, kv_cache: &Arc<dyn KvCacheHandle>, new_len: usize) -> Result<()> {
        self.inner.truncate_kv(kv_cache, new_len).await
    }
    async fn forward_verify(&self, inputs: &[DecodeInput]) -> Result<Vec<DecodeOutput>> {
        self.inner.forward_verify(inputs).await
    }
    fn capabilities(&self) -> ExecutorCapabilities {
        self.inner.capabilities()
    }
    fn status(&self) -> ExecutorStatus {
        self.inner.status()
    }
    fn cache_metrics_snapshot(&self) -> Option<serde_json::Value> {
        self.inner.cache_metrics_snapshot()
    }
    async fn warmup(&mut self) -> Result<()> {
        self.inner.warmup().await
    }
    async fn shutdown(&mut self) -> Result<()> {
        self.inner.shutdown().await
    }
    fn release_cache(&self, cache_id: &str) {
        self.inner.release_cache(cache_id)
    }
}
