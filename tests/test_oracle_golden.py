"""Pins the CPU oracle against the reference's own known-answer tests
(SURVEY.md §8c).  Each test names the reference test it re-derives."""
import numpy as np
import pytest


# ── ferrum-quantization/tests/gptq_parity_test.rs ────────────────────────────
def test_lcg_matches_reference_recurrence(oracle):
    # gptq_parity_test.rs:28-33: state = state*6364136223846793005 + 1442695040888963407; >>33
    st = 0xDEADBEEF
    lcg = oracle.Lcg(0xDEADBEEF)
    for _ in range(5):
        st = (st * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        assert lcg.u32() == (st >> 33) & 0xFFFFFFFF


def _py_dequant(qw, sc, qz, k, n, g, g_idx=None):
    # independent numpy restatement of dequant_reference (gptq_parity_test.rs:147-166)
    w = np.zeros((n, k), np.float32)
    kk = np.arange(k)
    shifts = (kk % 8) * 4
    q = (qw.view(np.uint32)[kk // 8, :] >> shifts[:, None].astype(np.uint32)) & 0xF      # [k,n]
    grp = (kk // g) if g_idx is None else g_idx
    cols = np.arange(n)
    z = ((qz.view(np.uint32)[grp][:, cols // 8] >> ((cols % 8) * 4).astype(np.uint32)) & 0xF) + 1
    w = ((q.astype(np.int32) - z.astype(np.int32)).astype(np.float32) * sc[grp, :]).T
    return np.ascontiguousarray(w)


def test_gptq_cpu_selfcheck(oracle):
    # gptq_parity_test.rs:108-145 cpu_selfcheck: K=256,N=128,g=128,seed 0xDEADBEEF, m=2,
    # input sin(0.001 i); load_gptq+forward vs from-scratch dequant + gemm: max|Δ| < 1e-3
    k, n, g, m = 256, 128, 128, 2
    qw, sc, qz = oracle.make_synthetic_gptq(k, n, g, 0xDEADBEEF)
    w = oracle.dequant_gptq(qw, sc, qz, g, k, n)
    w_ref = _py_dequant(qw, sc, qz, k, n, g)
    assert np.array_equal(w, w_ref)
    x = np.sin(np.arange(m * k, dtype=np.float32) * np.float32(0.001)).astype(np.float32).reshape(m, k)
    out = oracle.gemm(x, w, m, n, k)
    ref = (x.astype(np.float64) @ w_ref.astype(np.float64).T).astype(np.float32)
    assert np.max(np.abs(out - ref)) < 1e-3
    # scales are drawn from [0.01, 0.1) (:71-74)
    assert sc.min() >= 0.01 and sc.max() < 0.1


def test_gptq_symmetric_qzeros_encode_zero_point_8(oracle):
    # gptq_parity_test.rs:256-268
    qw, sc, qz = oracle.make_synthetic_gptq(512, 256, 128, 0x51A7E501, symmetric=True)
    assert np.all(qz.view(np.uint32) == 0x77777777)
    w = oracle.dequant_gptq(qw, sc, qz, 128, 512, 256)
    assert (w > 0).any() and (w < 0).any()
    # zero point 8: q - 8 ∈ [-8, 7]
    assert np.all(np.abs(w) <= 8 * sc.max() + 1e-6)


def test_gptq_desc_act_uses_g_idx(oracle):
    # gptq_parity_test.rs:232-254 and :270-321 (perm/gather ≡ g_idx reference)
    k, n, g = 512, 256, 128
    qw, sc, qz = oracle.make_synthetic_gptq(k, n, g, 0x5A170A7)
    g_idx = oracle.make_desc_act_g_idx(k, g)
    counts = np.bincount(g_idx, minlength=k // g)
    assert np.all(counts == g)
    seq = oracle.dequant_gptq(qw, sc, qz, g, k, n)
    da = oracle.dequant_gptq(qw, sc, qz, g, k, n, g_idx=g_idx)
    assert np.max(np.abs(seq - da)) > 1e-3
    assert np.array_equal(da, _py_dequant(qw, sc, qz, k, n, g, g_idx))
    # perm/gather transform
    perm = np.argsort(g_idx, kind="stable")
    m = 3
    x = np.sin(np.arange(m * k, dtype=np.float32) * np.float32(0.0041)).reshape(m, k).astype(np.float32)
    out_ref = oracle.gemm(x, da, m, n, k)
    sorted_w = da[:, perm]
    out_perm = oracle.gemm(x[:, perm], sorted_w, m, n, k)
    assert np.max(np.abs(out_ref - out_perm)) < 1e-3


# ── ferrum-kv/src/attention.rs tests ─────────────────────────────────────────
def _pool(num_blocks, bs, nkv, hd):
    return np.zeros((num_blocks, bs, nkv, hd), np.float32), np.zeros((num_blocks, bs, nkv, hd), np.float32)


def test_paged_single_token_decode_attention(oracle):
    # attention.rs:162-195: 3 cached tokens, K=1, V=pos+1, q=1 → mean(V)=2.0
    nh, hd, bs = 2, 4, 16
    pk, pv = _pool(4, bs, nh, hd)
    bt = [1]  # ferrum-kv BlockPool ids start from 1 (blocks/pool.rs:206); any id works
    for pos in range(3):
        pk[1, pos] = 1.0
        pv[1, pos] = pos + 1
    q = np.ones((1, nh, hd), np.float32)
    out = oracle.paged_attention(q, 1, nh, nh, hd, pk, pv, bt, bs, 3)
    assert out.shape == (1, nh, hd)
    assert np.all(np.abs(out - 2.0) < 1e-5)


def test_paged_prefill_causal_masking(oracle):
    # attention.rs:197-254
    nh, hd, bs = 1, 2, 16
    pk, pv = _pool(2, bs, nh, hd)
    data = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], np.float32)
    for pos in range(3):
        pk[0, pos, 0] = data[pos]
        pv[0, pos, 0] = data[pos]
    out = oracle.paged_attention(data.reshape(3, 1, 2), 3, 1, 1, hd, pk, pv, [0], bs, 3).reshape(-1)
    assert abs(out[0] - 1.0) < 1e-5 and abs(out[1]) < 1e-5
    assert out[2] < 0.5 and out[3] > 0.5
    # exact value for token 1: softmax([0, 1/sqrt(2)])
    w1 = 1.0 / (1.0 + np.exp(-1.0 / np.sqrt(2.0)))
    assert abs(out[3] - w1) < 1e-6


def test_paged_attention_across_blocks(oracle):
    # attention.rs:256-288: block_size 2, 4 tokens over 2 blocks
    nh, hd, bs = 1, 2, 2
    pk, pv = _pool(4, bs, nh, hd)
    bt = [3, 1]
    for pos in range(4):
        pk[bt[pos // bs], pos % bs, 0] = pos + 1
        pv[bt[pos // bs], pos % bs, 0] = (pos + 1) * 10.0
    out = oracle.paged_attention(np.ones((1, 1, 2), np.float32), 1, 1, 1, hd, pk, pv, bt, bs, 4)
    assert 10.0 < out[0, 0, 0] < 40.0
    s = np.array([1, 2, 3, 4], np.float64) * 2 / np.sqrt(2.0)
    p = np.exp(s - s.max()); p /= p.sum()
    assert abs(out[0, 0, 0] - (p * np.array([10, 20, 30, 40])).sum()) < 1e-4


def test_paged_attention_rejects_empty(oracle):
    pk, pv = _pool(1, 2, 1, 2)
    with pytest.raises(ValueError):
        oracle.paged_attention(np.ones((1, 1, 2), np.float32), 1, 1, 1, 2, pk, pv, [0], 2, 0)


def test_cpu_attention_equals_paged_reference(oracle):
    # the two reference formulations (online softmax cpu.rs:2179 vs 3-pass attention.rs:30)
    rng = np.random.default_rng(0)
    nh, nkv, hd, T, bs = 4, 2, 16, 7, 4
    q = rng.standard_normal((T, nh, hd)).astype(np.float32)
    k = rng.standard_normal((T, nkv, hd)).astype(np.float32)
    v = rng.standard_normal((T, nkv, hd)).astype(np.float32)
    pk, pv = _pool(4, bs, nkv, hd)
    bt = [2, 0]
    for p in range(T):
        pk[bt[p // bs], p % bs] = k[p]
        pv[bt[p // bs], p % bs] = v[p]
    paged = oracle.paged_attention(q, T, nh, nkv, hd, pk, pv, bt, bs, T)
    contig = oracle.cpu_attention(q.transpose(1, 0, 2), k.transpose(1, 0, 2), v.transpose(1, 0, 2),
                                  T, T, True, 0, nh, nkv, hd)
    assert np.max(np.abs(paged - contig.transpose(1, 0, 2))) < 1e-5


# ── ferrum-models/src/common/paged_pool.rs tests ─────────────────────────────
def test_allocator_basic(oracle):
    # paged_pool.rs:465-481
    a = oracle.BlockAllocator(4)
    assert a.free_count() == 4
    assert [a.allocate() for _ in range(4)] == [0, 1, 2, 3]
    with pytest.raises(RuntimeError):
        a.allocate()
    a.free([1, 3])
    assert a.free_count() == 2
    assert a.allocate() == 3 and a.allocate() == 1


def test_allocator_atomic_n_failure_and_peak(oracle):
    # paged_pool.rs:483-502
    a = oracle.BlockAllocator(3)
    a.allocate(); a.allocate()
    with pytest.raises(RuntimeError):
        a.allocate_n(2)
    assert a.free_count() == 1
    b = oracle.BlockAllocator(8)
    blocks = b.allocate_n(5)
    assert b.peak_in_use() == 5
    b.free(blocks)
    assert b.peak_in_use() == 5
    b.allocate_n(3)
    assert b.peak_in_use() == 5


def test_allocator_refcounts(oracle):
    # paged_pool.rs:504-569
    a = oracle.BlockAllocator(4)
    b = a.allocate()
    assert a.ref_count(b) == 1
    a.acquire(b); a.acquire(b)
    assert a.ref_count(b) == 3
    a.free([b])
    assert a.ref_count(b) == 2 and a.free_count() == 3
    a.free([b]); a.free([b])
    assert a.ref_count(b) == 0 and a.free_count() == 4
    assert a.allocate() == b  # LIFO reuse


def test_allocator_hash_table(oracle):
    # paged_pool.rs:605-689
    a = oracle.BlockAllocator(4)
    b = a.allocate()
    a.register_block_hash(b, 0xDEADBEEF)
    assert a.hash_table_size() == 1
    assert a.try_acquire_by_hash(0xDEADBEEF) == b and a.ref_count(b) == 2
    # soft-free resurrection
    a2 = oracle.BlockAllocator(4)
    b = a2.allocate()
    a2.register_block_hash(b, 0xABCD1234)
    a2.free([b])
    assert a2.free_count() == 4 and a2.ref_count(b) == 0
    assert a2.try_acquire_by_hash(0xABCD1234) == b
    assert a2.ref_count(b) == 1 and a2.free_count() == 3
    assert a2.try_acquire_by_hash(99) is None
    # evict on realloc
    a3 = oracle.BlockAllocator(2)
    b1 = a3.allocate(); a3.allocate()
    a3.register_block_hash(b1, 0x1234)
    a3.free([b1])
    assert a3.hash_table_size() == 1
    assert a3.allocate() == b1 and a3.hash_table_size() == 0
    assert a3.try_acquire_by_hash(0x1234) is None
    # prefers un-hashed free block
    a4 = oracle.BlockAllocator(3)
    cached = a4.allocate()
    a4.register_block_hash(cached, 0xCAFEBABE)
    a4.free([cached])
    fresh = a4.allocate()
    assert fresh != cached and a4.hash_table_size() == 1
    assert a4.try_acquire_by_hash(0xCAFEBABE) == cached
    # replace hash
    a5 = oracle.BlockAllocator(4)
    b = a5.allocate()
    a5.register_block_hash(b, 1); a5.register_block_hash(b, 2)
    assert a5.hash_table_size() == 1
    assert a5.try_acquire_by_hash(1) is None and a5.try_acquire_by_hash(2) == b


# ── ferrum-models/src/moe/router.rs ──────────────────────────────────────────
def _route_sort_based(logits, top_k, norm):
    # independent restatement of the sort-based `route` (router.rs:76-95 contract:
    # stable max-subtract softmax, first-tie-wins top-K, optional sum-renorm)
    ids, ws = [], []
    for row in logits:
        e = np.exp((row - row.max()).astype(np.float32)).astype(np.float32)
        s = np.float32(0)
        for v in e:
            s = np.float32(s + v)
        p = (e * np.float32(np.float32(1.0) / s)).astype(np.float32)
        order = sorted(range(len(p)), key=lambda i: (-p[i], i))[:top_k]
        w = p[order]
        if norm:
            t = np.float32(0)
            for v in w:
                t = np.float32(t + v)
            w = (w * np.float32(np.float32(1.0) / t)).astype(np.float32)
        ids.append(order); ws.append(w)
    return np.array(ids, np.uint32), np.array(ws, np.float32)


@pytest.mark.parametrize("batch,ne,k,norm,seed", [(32, 128, 8, True, 0xDEADBEEF), (1, 128, 8, True, 0x1234),
                                                  (64, 128, 8, True, 0x5678), (8, 64, 4, False, 0xC0FFEE),
                                                  (4, 16, 1, True, 0x42), (4, 16, 1, False, 0x42)])
def test_router_parity_shapes(oracle, batch, ne, k, norm, seed):
    # router.rs:203-244 shapes/seeds (logits regenerated with the LCG: the reference uses rand::StdRng)
    lcg = oracle.Lcg(seed)
    logits = lcg.array_f32(batch * ne, -3.0, 3.0).reshape(batch, ne)
    ids, w = oracle.route_topk(logits, ne, k, norm)
    ids2, w2 = _route_sort_based(logits, k, norm)
    assert np.array_equal(ids, ids2)
    assert np.max(np.abs(w - w2)) < 1e-6
    if norm:
        assert np.all(np.abs(w.sum(axis=1) - 1.0) < 1e-5)


def test_router_ties_take_lowest_index(oracle):
    # router.rs:111-112,159-178: strict `>` keeps the first tied entry; all-equal logits (:252)
    logits = np.full((2, 128), 0.5, np.float32)
    ids, w = oracle.route_topk(logits, 128, 8, True)
    assert np.array_equal(ids, np.tile(np.arange(8, dtype=np.uint32), (2, 1)))
    assert np.all(np.abs(w - 0.125) < 1e-7)


def test_bucket_plan_stable_counting_sort(oracle):
    # dispatch.rs:1408-1461
    ids = np.array([[1, 3], [1, 0], [3, 1]], np.uint32)
    offsets, packed, pairs = oracle.bucket_plan(ids, 3, 4, 2)
    assert list(offsets) == [0, 1, 4, 4, 6]
    assert list(packed) == [1, 0, 1, 2, 0, 2]
    assert list(pairs) == [1, 4, 2, 0, 5, 3]


def test_compute_ids_tpe_example(oracle):
    # ferrum-kernels/src/moe_host.rs:60-75
    tpe, ids, mpe = oracle.compute_ids_tpe([1, 3, 1, 0], 4, 2, 2)
    assert list(tpe) == [1, 2, 0, 1] and mpe == 2
    assert ids[0] == 3 and ids[2] == 0 and ids[3] == 2 and ids[6] == 1


def test_moe_align_block_size(oracle):
    # moe_align_block_size_pair_ids.cu: padded per-expert regions, sentinel = T·k
    ids = np.array([1, 3, 1, 0, 3, 1], np.int32)
    sorted_ids, block_ids, total = oracle.moe_align_block_size(ids, 4, 4)
    assert total == 12 and list(block_ids) == [0, 1, 3]
    assert list(sorted_ids[:12]) == [3, 6, 6, 6, 0, 2, 5, 6, 1, 4, 6, 6]
    assert np.all(sorted_ids[12:] == 6)


def test_moe_forward_matches_bucketed_order(oracle):
    # ferrum-models/tests/moe_bucketed_parity_test.rs: bucketed ≡ per-token on CPU
    rng = np.random.default_rng(1)
    B, H, I, E, K = 5, 32, 16, 6, 2
    x = rng.standard_normal((B, H)).astype(np.float32)
    gw = (rng.standard_normal((E, 2 * I, H)) * 0.2).astype(np.float32)
    dw = (rng.standard_normal((E, H, I)) * 0.2).astype(np.float32)
    logits = rng.standard_normal((B, E)).astype(np.float32)
    ids, w = oracle.route_topk(logits, E, K, True)
    out = oracle.moe_forward_cpu(x, H, I, K, ids, w, gw, dw)
    # bucketed: group rows per expert, batched gemm, weighted combine in k order
    offsets, packed, pairs = oracle.bucket_plan(ids, B, E, K)
    down = np.zeros((B * K, H), np.float32)
    for e in range(E):
        rows = packed[offsets[e]:offsets[e + 1]]
        if len(rows) == 0:
            continue
        gu = oracle.gemm(x[rows], gw[e], len(rows), 2 * I, H)
        act = oracle.fused_silu_mul_split(gu, I)
        down[offsets[e]:offsets[e + 1]] = oracle.gemm(act, dw[e], len(rows), H, I)
    out2 = np.zeros((B, H), np.float32)
    for b in range(B):
        for k in range(K):
            out2[b] += w[b, k] * down[pairs[b * K + k]]
    assert np.array_equal(out, out2)


# ── llama_family.rs KATs ─────────────────────────────────────────────────────
def test_bf16_round_known_values(oracle):
    # llama_family.rs:5951-5959
    r = oracle.bf16_round(float(np.float32(np.sqrt(1152.0))))
    bits = np.float32(r).view(np.uint32)
    assert bits & 0xFFFF == 0
    assert abs(r - 33.941125) < 0.25
    assert oracle.bf16_round(32.0) == 32.0


def test_linear_rope_scaling_divides_frequency(oracle):
    # llama_family.rs:6044-6054
    scaled = oracle.rope_freq(10000.0, 4, 0, 1, (8.0, 0, 0, 0))
    unscaled = oracle.rope_freq(10000.0, 4, 0, 0)
    assert abs(scaled - unscaled / 8.0) < 1e-12


def test_llama3_rope_scaling_bands(oracle):
    # llama_family.rs:5262-5282: high-freq untouched, low-freq /factor, smooth in between
    f, lo, hi, orig = 8.0, 1.0, 4.0, 8192.0
    base0 = oracle.rope_freq(500000.0, 128, 0)
    assert oracle.rope_freq(500000.0, 128, 0, 2, (f, lo, hi, orig)) == base0
    base63 = oracle.rope_freq(500000.0, 128, 63)
    assert abs(oracle.rope_freq(500000.0, 128, 63, 2, (f, lo, hi, orig)) - base63 / f) < 1e-18
    mid = None
    for i in range(64):
        b = oracle.rope_freq(500000.0, 128, i)
        wl = 2 * np.pi / b
        if orig / hi <= wl <= orig / lo:
            mid = i
            smooth = (orig / wl - lo) / (hi - lo)
            expect = (1 - smooth) * b / f + smooth * b
            assert abs(oracle.rope_freq(500000.0, 128, i, 2, (f, lo, hi, orig)) - expect) < 1e-15
    assert mid is not None


def test_rope_cache_f64_angles(oracle):
    # llama_family.rs:5220-5237: angle = pos·freq in f64, stored f32
    cos, sin = oracle.build_rope_cache(1e6, 128, 40)
    pos, i = 37, 5
    ang = pos * (1.0 / (1e6 ** (2 * i / 128)))
    assert cos[pos, i] == np.float32(np.cos(ang)) and sin[pos, i] == np.float32(np.sin(ang))


def test_gelu_tanh_mul_split_matches_reference(oracle):
    # llama_family.rs:6056-6074
    gate_up = np.array([[-1.0, 0.5, 2.0, 1.0, 2.0, 3.0]], np.float32)
    v = oracle.fused_gelu_tanh_mul_split(gate_up, 3)[0]
    for i in range(3):
        x = np.float32(gate_up[0, i])
        gelu = np.float32(0.5) * x * (np.float32(1.0) + np.tanh(np.float32(0.79788456) * (x + np.float32(0.044715) * x * x * x)))
        assert abs(v[i] - gelu * gate_up[0, 3 + i]) < 1e-6


def test_scale_inplace_scales(oracle):
    # llama_family.rs:6076-6086
    v = oracle.scale_inplace(np.array([1.0, -2.0, 0.5], np.float32), 33.9375)
    assert list(v) == [33.9375, -67.875, 16.96875]


# ── sampler.rs / traits.rs tie-breaks ────────────────────────────────────────
def test_argmax_first_max_vs_greedy_last_max(oracle):
    # traits.rs:1534-1555 (first max) vs sampler.rs:359-378 (Iterator::max_by → last max)
    logits = np.array([[0.0, 5.0, 1.0, 5.0, -1.0]], np.float32)
    assert oracle.argmax_rows(logits)[0] == 1
    assert oracle.greedy_sample(logits[0]) == 3


def test_repetition_penalty_semantics(oracle):
    # sampler.rs:327-345: v>0 → v/p, else v·p; duplicates and out-of-range ids ignored
    l = oracle.repetition_penalty(np.array([2.0, -2.0, 3.0, 0.0], np.float32), [0, 1, 0, 9], 1.1)
    assert abs(l[0] - 2.0 / 1.1) < 1e-6 and abs(l[1] + 2.2) < 1e-6 and l[2] == 3.0 and l[3] == 0.0


def test_top_k_top_p_temperature(oracle):
    # sampler.rs:196-309
    l = np.array([1.0, 4.0, 2.0, 3.0, 0.5], np.float32)
    tk = oracle.top_k(l, 2)
    assert np.isneginf(tk[[0, 2, 4]]).all() and tk[1] == 4.0 and tk[3] == 3.0
    tp = oracle.top_p(l, 0.7)
    p = np.exp(l - l.max()); p /= p.sum()
    assert p[1] < 0.7 < p[1] + p[3]
    assert tp[1] == 4.0 and tp[3] == 3.0 and np.isneginf(tp[[0, 2, 4]]).all()
    assert np.allclose(oracle.temperature(l, 0.5), l / 0.5)
    assert np.array_equal(oracle.temperature(l, 0.0), l)
    # multinomial: threshold = u32/u32::MAX, first idx with cumulative >= threshold
    assert oracle.multinomial(l, 0) == 0
    assert oracle.multinomial(l, 0xFFFFFFFF) in (3, 4)


# ── op-level identities from cpu.rs ──────────────────────────────────────────
def test_rms_norm_and_fused_add(oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 64)).astype(np.float32)
    r = rng.standard_normal((3, 64)).astype(np.float32)
    w = rng.standard_normal(64).astype(np.float32)
    out = oracle.rms_norm(x, w, 1e-6)
    ref = x / np.sqrt((x.astype(np.float64) ** 2).mean(axis=1, keepdims=True) + 1e-6) * w
    assert np.max(np.abs(out - ref)) < 1e-5
    r2, o2 = oracle.fused_add_rms_norm(r, x, w, 1e-6)
    assert np.array_equal(r2, r + x)
    assert np.array_equal(o2, oracle.rms_norm(r + x, w, 1e-6))


def test_qk_norm_rope_modes(oracle):
    rng = np.random.default_rng(3)
    T, Hh, hd = 3, 2, 8
    x = rng.standard_normal((T, Hh, hd)).astype(np.float32)
    w = rng.standard_normal(hd).astype(np.float32)
    cos, sin = oracle.build_rope_cache(10000.0, hd, 16)
    o0 = oracle.qk_norm_rope(x, w, cos, sin, T, Hh, hd, 5, 1e-6, 0)
    assert np.array_equal(o0, x.transpose(1, 0, 2))
    o2 = oracle.qk_norm_rope(x, w, cos, sin, T, Hh, hd, 5, 1e-6, 2)
    half = hd // 2
    for t in range(T):
        c, s = cos[5 + t], sin[5 + t]
        x0, x1 = x[t, :, :half], x[t, :, half:]
        assert np.allclose(o2[:, t, :half], x0 * c - x1 * s, atol=1e-6)
        assert np.allclose(o2[:, t, half:], x1 * c + x0 * s, atol=1e-6)
    o1 = oracle.qk_norm_rope(x, w, cos, sin, T, Hh, hd, 5, 1e-6, 1)
    xn = x / np.sqrt((x ** 2).mean(axis=-1, keepdims=True) + 1e-6) * w
    o1b = oracle.qk_norm_rope(xn.astype(np.float32), w, cos, sin, T, Hh, hd, 5, 1e-6, 2)
    assert np.allclose(o1, o1b, atol=1e-5)
    o3 = oracle.qk_norm_rope(x, w, cos, sin, T, Hh, hd, 5, 1e-6, 3)
    for t in range(T):
        c, s = cos[5 + t], sin[5 + t]
        assert np.allclose(o3[:, t, 0::2], x[t, :, 0::2] * c - x[t, :, 1::2] * s, atol=1e-6)


# ── KV block content hashes (paged_pool.rs:60-98) ────────────────────────────
def test_siphash_published_vectors_and_block_hash_chain_properties(oracle):
    O = oracle
    """SipHash-c-d pinned by the paper's SipHash-2-4 vectors (key 00..0f: empty message and 00..0e); the block chain is
    the 1-3 variant with a zero key (Rust DefaultHasher) and must satisfy the reference's chain tests
    (paged_pool.rs:578-606): chained, prefix-stable, trailing partial block dropped."""
    import struct
    k0, k1 = struct.unpack("<QQ", bytes(range(16)))
    assert O.siphash(2, 4, k0, k1, b"") == 0x726FDB47DD0E0E31
    assert O.siphash(2, 4, k0, k1, bytes(range(15))) == 0xA129CA6149BE45E5
    chain = O.block_hash_chain([1, 2, 3, 4, 5, 6, 7, 8], 4)
    assert len(chain) == 2 and chain[0] != chain[1]
    ca = O.block_hash_chain([10, 20, 30, 40, 50, 60, 70, 80], 4)
    cb = O.block_hash_chain([10, 20, 30, 40, 99, 99, 99, 99], 4)
    assert ca[0] == cb[0] and ca[1] != cb[1]
    assert len(O.block_hash_chain([1, 2, 3, 4, 5, 6, 7], 4)) == 1
    # chaining: hash[1] depends on hash[0] — the same second block under a different first block hashes differently
    assert O.block_hash_chain([9, 9, 9, 9, 5, 6, 7, 8], 4)[1] != chain[1]
    # explicit construction: parent u64 LE ‖ token u32 LE, SipHash-1-3, zero key
    msg = struct.pack("<Q4I", 0, 1, 2, 3, 4)
    assert O.siphash(1, 3, 0, 0, msg) == int(chain[0])


def test_layer_norm_and_gelu_follow_cpu_rs(oracle):
    O = oracle
    """cpu.rs:2081-2122: layer_norm (f64 mean / variance, f32 affine) and the exact-form GELU whose erf is the reference's own
    Abramowitz-Stegun polynomial (cpu.rs:2263-2273: |error| <= 1.5e-7 against the true erf)."""
    from scipy.special import erf
    x = np.array([[1.0, 2.0, 3.0, 4.0], [-2.0, 0.0, 2.0, 0.0]], np.float32)
    g = np.array([1.0, 0.5, 2.0, 1.0], np.float32)
    b = np.array([0.0, 1.0, -1.0, 0.5], np.float32)
    out = O.layer_norm(x, g, b, 0.0)
    # row 0: mean 2.5, var 1.25 -> (x - 2.5) / sqrt(1.25); row 1: mean 0, var 2
    exp0 = (x[0] - 2.5) / np.sqrt(1.25) * g + b
    exp1 = x[1] / np.sqrt(2.0) * g + b
    assert np.allclose(out[0], exp0, atol=1e-6) and np.allclose(out[1], exp1, atol=1e-6)
    v = np.linspace(-6, 6, 97).astype(np.float32)
    ref = 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    assert np.max(np.abs(O.gelu(v[None])[0] - ref)) < 1e-6
    assert O.gelu(np.zeros((1, 3), np.float32)).tolist() == [[0.0, 0.0, 0.0]]
