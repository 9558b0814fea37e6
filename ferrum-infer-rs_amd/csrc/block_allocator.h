// Host-side physical KV block bookkeeping — the product's own implementation of the reference's
// `BlockAllocator` contract (ferrum-models/src/common/paged_pool.rs:106-365): block ids start at 0,
// the free list is LIFO, `allocate` prefers the most recently freed block that carries no prefix-cache
// hash (rposition + swap_remove, :182-193), ref counts gate physical release, and a hash table lets
// soft-freed blocks be resurrected.  "Bit-exact KV-block indexing" means this class hands out the
// same ids in the same order as the reference for the same call sequence.
#pragma once
#include <stdint.h>

#include <unordered_map>
#include <vector>

namespace fh {

// SipHash-c-d (Aumasson & Bernstein), little-endian message words, 64-bit output.
inline uint64_t siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t* data, size_t len) {
    uint64_t v0 = k0 ^ 0x736f6d6570736575ull, v1 = k1 ^ 0x646f72616e646f6dull;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ull, v3 = k1 ^ 0x7465646279746573ull;
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    const size_t full = len / 8;
    for (size_t i = 0; i < full; i++) {
        uint64_t m = 0;
        for (int j = 0; j < 8; j++) m |= (uint64_t)data[8 * i + j] << (8 * j);
        v3 ^= m;
        for (int r = 0; r < c_rounds; r++) round();
        v0 ^= m;
    }
    uint64_t b = (uint64_t)(len & 0xff) << 56;
    for (size_t j = 0; j < (len & 7); j++) b |= (uint64_t)data[8 * full + j] << (8 * j);
    v3 ^= b;
    for (int r = 0; r < c_rounds; r++) round();
    v0 ^= b;
    v2 ^= 0xff;
    for (int r = 0; r < d_rounds; r++) round();
    return v0 ^ v1 ^ v2 ^ v3;
}

// Content hash of one KV block chained from its parent (paged_pool.rs:76-84): Rust's `DefaultHasher::new()` is
// SipHash-1-3 with a zero key; `parent.hash()` feeds the u64 and each `TokenId` its u32, native (little) endian.
inline uint64_t block_hash(uint64_t parent, const uint32_t* tokens, int n) {
    std::vector<uint8_t> msg(8 + 4 * (size_t)n);
    for (int j = 0; j < 8; j++) msg[j] = (uint8_t)(parent >> (8 * j));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) msg[8 + 4 * i + j] = (uint8_t)(tokens[i] >> (8 * j));
    return siphash(1, 3, 0, 0, msg.data(), msg.size());
}

// block_hash_chain (paged_pool.rs:89-98): one hash per FULL block; a trailing partial block is dropped.
inline std::vector<uint64_t> block_hash_chain(const uint32_t* tokens, int n, int block_size) {
    std::vector<uint64_t> out;
    uint64_t parent = 0;
    for (int i = 0; i + block_size <= n; i += block_size) {
        parent = block_hash(parent, tokens + i, block_size);
        out.push_back(parent);
    }
    return out;
}

class BlockAllocator {
public:
    explicit BlockAllocator(uint32_t num_blocks)
        : capacity_(num_blocks), ref_counts_(num_blocks, 0), has_hash_(num_blocks, 0), block_hash_(num_blocks, 0) {
        free_list_.reserve(num_blocks);
        for (uint32_t i = 0; i < num_blocks; i++) free_list_.push_back(num_blocks - 1 - i);   // pop yields 0 first
    }

    bool allocate(uint32_t* out) {
        if (free_list_.empty()) return false;
        uint32_t b = pop_preferring_unhashed();
        evict_hash_if_any(b);
        ref_counts_[b] = 1;
        track_peak();
        *out = b;
        return true;
    }

    // all-or-nothing
    bool allocate_n(uint32_t n, uint32_t* out) {
        if (free_list_.size() < n) return false;
        for (uint32_t i = 0; i < n; i++) {
            uint32_t b = pop_preferring_unhashed();
            evict_hash_if_any(b);
            ref_counts_[b] = 1;
            out[i] = b;
        }
        track_peak();
        return true;
    }

    void free(const uint32_t* blocks, uint32_t n) {
        for (uint32_t i = 0; i < n; i++) {
            uint32_t b = blocks[i];
            if (ref_counts_[b] == 0) continue;   // double free: ignore (reference debug-asserts)
            if (--ref_counts_[b] == 0) free_list_.push_back(b);
        }
    }

    void acquire(uint32_t block) { ref_counts_[block]++; }

    void register_block_hash(uint32_t block, uint64_t hash) {
        if (has_hash_[block]) {
            uint64_t old = block_hash_[block];
            if (old == hash) return;
            auto it = table_.find(old);
            if (it != table_.end() && it->second == block) table_.erase(it);
        }
        has_hash_[block] = 1;
        block_hash_[block] = hash;
        table_[hash] = block;   // last writer wins
    }

    // returns the block id or -1 on a miss
    int64_t try_acquire_by_hash(uint64_t hash) {
        auto it = table_.find(hash);
        if (it == table_.end()) return -1;
        uint32_t block = it->second;
        if (ref_counts_[block] == 0) {
            // soft-free: pull it out of the free list (search from the back, swap_remove)
            size_t pos = free_list_.size();
            for (size_t i = free_list_.size(); i-- > 0;)
                if (free_list_[i] == block) { pos = i; break; }
            if (pos == free_list_.size()) return -1;
            free_list_[pos] = free_list_.back();
            free_list_.pop_back();
            ref_counts_[block] = 1;
            track_peak();
        } else {
            if (ref_counts_[block] == 0xFFFF) return -1;   // would overflow the u16 count: treated as a miss
            ref_counts_[block]++;
        }
        return (int64_t)block;
    }

    uint32_t free_count() const { return (uint32_t)free_list_.size(); }
    uint32_t capacity() const { return capacity_; }
    uint32_t ref_count(uint32_t b) const { return ref_counts_[b]; }
    uint32_t peak_in_use() const { return peak_; }
    uint32_t hash_table_size() const { return (uint32_t)table_.size(); }

private:
    uint32_t pop_preferring_unhashed() {
        size_t pos = free_list_.size() - 1;
        for (size_t i = free_list_.size(); i-- > 0;)
            if (!has_hash_[free_list_[i]]) { pos = i; break; }
        uint32_t b = free_list_[pos];
        free_list_[pos] = free_list_.back();
        free_list_.pop_back();
        return b;
    }
    void evict_hash_if_any(uint32_t block) {
        if (!has_hash_[block]) return;
        has_hash_[block] = 0;
        auto it = table_.find(block_hash_[block]);
        if (it != table_.end() && it->second == block) table_.erase(it);
    }
    void track_peak() {
        uint32_t in_use = capacity_ - (uint32_t)free_list_.size();
        if (in_use > peak_) peak_ = in_use;
    }

    uint32_t capacity_;
    uint32_t peak_ = 0;
    std::vector<uint32_t> free_list_;
    std::vector<uint16_t> ref_counts_;
    std::vector<uint8_t> has_hash_;
    std::vector<uint64_t> block_hash_;
    std::unordered_map<uint64_t, uint32_t> table_;
};

}  // namespace fh
