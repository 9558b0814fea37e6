// Native paged-KV tile layout (block_size 16).  One (block, kv_head) tile is 16·head_dim fp16
// (4 KiB at head_dim 128) and is shaped so that ONE coalesced 1-KiB wave load is an MFMA
// operand fragment of v_mfma_f32_16x16x32_f16 (see attention.hip):
//   K tile : [hd/8 chunks][16 keys][8 dims]             lane (a,b) of load s ↔ key b, chunk 4s+a
//   V tile : [hd/32][4 key-quads][16 d][2 d-sub][4 keys]  lane (a,b) of load p ↔ keys 4a..4a+3,
//                                                        dims 32p+b and 32p+16+b
// Block ids / block tables are the reference's (ferrum-models/src/common/paged_pool.rs:106-459);
// only the bytes inside a block differ from the CUDA lane's [slot][head][dim]
// (kernels/split_qkv_norm_rope_into_paged_cache.cu:100-112).
#pragma once

namespace fh {

constexpr int KV_BLOCK = 16;

__host__ __device__ inline int kv_tile_elems(int head_dim) { return KV_BLOCK * head_dim; }
__host__ __device__ inline int k_tile_off(int slot, int d) { return ((d >> 3) * KV_BLOCK + slot) * 8 + (d & 7); }
__host__ __device__ inline int v_tile_off(int slot, int d) {
    int dt = d >> 4, dtp = dt >> 1, dsub = dt & 1, b = d & 15, a = slot >> 2, kk = slot & 3;
    return (((dtp * 4 + a) * 16 + b) * 2 + dsub) * 4 + kk;
}

}  // namespace fh
