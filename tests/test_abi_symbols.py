"""C-ABI surface (no GPU needed): the library loads, exports every symbol include/ferrum_hip.h
declares, carries the native-operator descriptor, and the host-only pieces (BlockAllocator,
argument validation) behave like the reference's."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    return ge.build()


def _declared():
    text = open(os.path.join(ROOT, "include", "ferrum_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ferrum_(?:hip|native)_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkg):
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    declared = _declared()
    assert len(declared) > 60
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in ferrum_hip.h but not exported: {missing}"
    # nothing CUDA-flavoured or oracle-flavoured leaks into the product
    assert not [s for s in exported if s.startswith("fo_")]


def test_native_operator_descriptor(pkg):
    # ferrum-native-ops/src/abi.rs:3-13 + resolver.rs:478-481 fixture contract
    lib = pkg.load_library()
    assert lib.ferrum_native_op_init() == 0

    class Desc(C.Structure):
        _fields_ = [("abi_version", C.c_uint32), ("operator_name", C.c_char_p), ("operator_abi_version", C.c_char_p)]
    d = C.cast(lib.ferrum_native_op_descriptor(), C.POINTER(Desc)).contents
    assert d.abi_version == 1 and d.operator_name == b"ferrum_hip_decode" and d.operator_abi_version == b"1"


def test_product_does_not_link_the_oracle(pkg):
    out = subprocess.check_output(["ldd", pkg.LIB_PATH], text=True)
    assert "oracle" not in out
    src = os.path.join(ROOT, "ferrum-infer-rs_amd")
    for dirpath, _, files in os.walk(src):
        for f in files:
            if f.endswith((".py", ".hip", ".cc", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("no CPU fallback", ""), f


def test_block_allocator_rejects_out_of_range_ids_and_ref_count_overflow(pkg):
    """The reference indexes its vectors and uses checked_add (a bad id or a wrapped count panics, paged_pool.rs:333-365);
    over the C ABI the same mistakes must come back as errors and leave the allocator untouched."""
    from ferrum_infer_rs_amd.backend import BlockAllocator
    a = BlockAllocator(4)
    b0 = a.allocate()
    for bad in (4, 5, 2 ** 31):
        with pytest.raises(RuntimeError):
            a.free([b0, bad])                                    # nothing of the call is applied
        with pytest.raises(RuntimeError):
            a.acquire(bad)
        with pytest.raises(RuntimeError):
            a.register_block_hash(bad, 7)
        assert a.ref_count(bad) == 0
    assert a.ref_count(b0) == 1 and a.free_count() == 3
    for _ in range(0xFFFF - 1):
        a.acquire(b0)
    assert a.ref_count(b0) == 0xFFFF
    with pytest.raises(RuntimeError):
        a.acquire(b0)                                            # u16 count would wrap
    assert a.ref_count(b0) == 0xFFFF


def test_block_allocator_matches_reference_sequences(pkg, oracle):
    # paged_pool.rs:465-481 + randomised differential test against the oracle restatement
    from ferrum_infer_rs_amd.backend import BlockAllocator
    a = BlockAllocator(4)
    assert [a.allocate() for _ in range(4)] == [0, 1, 2, 3]
    with pytest.raises(RuntimeError):
        a.allocate()
    a.free([1, 3])
    assert a.allocate() == 3 and a.allocate() == 1
    rng = np.random.default_rng(0)
    prod, ref = BlockAllocator(64), oracle.BlockAllocator(64)
    live, hashed = [], {}
    for step in range(3000):
        op = rng.integers(0, 6)
        if op <= 1 and prod.free_count() > 0:
            x, y = prod.allocate(), ref.allocate()
            assert x == y
            live.append(x)
        elif op == 2 and live:
            k = int(rng.integers(1, min(4, len(live)) + 1))
            idx = sorted(rng.choice(len(live), size=k, replace=False), reverse=True)
            blocks = [live.pop(i) for i in idx]
            prod.free(blocks); ref.free(blocks)
        elif op == 3 and live:
            b = live[int(rng.integers(len(live)))]
            h = int(rng.integers(1, 40))
            prod.register_block_hash(b, h); ref.register_block_hash(b, h)
            hashed[h] = b
        elif op == 4:
            h = int(rng.integers(1, 40))
            x, y = prod.try_acquire_by_hash(h), ref.try_acquire_by_hash(h)
            assert x == y
            if x is not None:
                live.append(x)
        elif op == 5:
            n = int(rng.integers(0, 5))
            if n <= prod.free_count():
                x, y = prod.allocate_n(n), ref.allocate_n(n)
                assert x == y
                live.extend(x)
            else:
                with pytest.raises(RuntimeError):
                    prod.allocate_n(n)
        assert prod.free_count() == ref.free_count()
        assert prod.hash_table_size() == ref.hash_table_size()
    assert prod.peak_in_use() == ref.peak_in_use()


def test_gptq_load_rejects_unsupported_shapes(pkg):
    # capability ops signal `unsupported` (capabilities.rs:147) — checked before any device work
    from ferrum_infer_rs_amd.backend import HipBackend, Unsupported
    qw = np.zeros((8, 16), np.int32); sc = np.ones((1, 16), np.float32); qz = np.zeros((1, 2), np.int32)
    with pytest.raises(Unsupported):
        HipBackend.load_gptq(qw, sc, qz, None, None, 8, 128, 128, 16)     # bits != 4
    with pytest.raises(Unsupported):
        HipBackend.load_gptq(qw, sc, qz, None, None, 4, 64, 64, 16)       # K % 128
    with pytest.raises(Unsupported):
        HipBackend.load_gptq(qw, sc, qz, None, None, 4, 32, 128, 16)      # group 32


def test_model_create_validates_config(pkg):
    from ferrum_infer_rs_amd import HipModel
    with pytest.raises(RuntimeError):
        HipModel(num_layers=1, hidden=100, num_heads=2, num_kv_heads=1, head_dim=128, intermediate=128, vocab=64)
    with pytest.raises(RuntimeError):
        HipModel(num_layers=1, hidden=128, num_heads=3, num_kv_heads=2, head_dim=128, intermediate=128, vocab=64)


def test_block_hash_chain_matches_restatement_and_published_siphash(pkg, oracle):
    """paged_pool.rs:60-98 through the product's C ABI: same chain as the independent restatement on random prompts,
    SipHash-2-4 paper vectors through the product's own routine."""
    import struct
    lib = pkg.load_library()
    lib.ferrum_hip_siphash.restype = C.c_uint64
    lib.ferrum_hip_siphash.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t]
    k0, k1 = struct.unpack("<QQ", bytes(range(16)))
    assert lib.ferrum_hip_siphash(2, 4, k0, k1, b"", 0) == 0x726FDB47DD0E0E31
    assert lib.ferrum_hip_siphash(2, 4, k0, k1, bytes(range(15)), 15) == 0xA129CA6149BE45E5
    rng = np.random.default_rng(5)
    for n in (0, 15, 16, 17, 100, 256, 1000):
        toks = rng.integers(0, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
        out = np.zeros(n // 16 + 1, np.uint64)
        cnt = C.c_int()
        assert lib.ferrum_hip_block_hash_chain(toks.ctypes.data_as(C.POINTER(C.c_uint32)), n, 16,
                                               out.ctypes.data_as(C.POINTER(C.c_uint64)), len(out), C.byref(cnt)) == 0
        ref = oracle.block_hash_chain(toks, 16)
        assert cnt.value == n // 16 == len(ref) and np.array_equal(out[:cnt.value], ref)


def test_native_operator_manifest_matches_the_binary(pkg, tmp_path):
    """tools/make_manifest.py emits the reference's manifest schema (native_operator.rs:39-58); the resolver's fail-closed
    checks (resolver.rs:136-318) are reproduced here: sha256 of the file, every declared export visible to nm."""
    import hashlib
    import json
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "make_manifest.py")])
    m = json.load(open(os.path.join(os.path.dirname(pkg.LIB_PATH), "native_operator_manifest.json")))
    assert m["schema_version"] == 1 and m["ferrum_native_abi_version"] == "1" and m["linkage"] == "dynamic"
    assert m["binary_sha256"] == hashlib.sha256(open(pkg.LIB_PATH, "rb").read()).hexdigest()
    assert len(m["inputs_sha256"]) == 64 and m["compute_capabilities"] == ["gfx950"]
    out = subprocess.check_output(["nm", "-g", "--defined-only", pkg.LIB_PATH], text=True)
    visible = set(line.split()[-1] for line in out.splitlines())
    assert "ferrum_native_op_init" in m["exports"] and "ferrum_native_op_descriptor" in m["exports"]
    assert all(e in visible for e in m["exports"])


def test_host_sampler_chain_matches_restatement(pkg, oracle):
    """sampler.rs:186-467 through the product's C ABI vs the independent restatement, bit for bit, on random logits with
    ties, -inf entries, repeated previous tokens and the edge parameters (k ≥ n, p outside (0,1), penalty 1, T ≤ 0)."""
    lib = pkg.load_library()
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)

    class Params(C.Structure):
        _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("repetition_penalty", C.c_float),
                    ("previous_tokens", up), ("num_previous_tokens", C.c_int32), ("greedy", C.c_int32),
                    ("random_u32", C.c_uint32), ("_pad", C.c_uint32)]
    rng = np.random.default_rng(17)
    for trial in range(60):
        n = int(rng.choice([5, 64, 1000, 4096]))
        logits = (rng.standard_normal(n) * 3).astype(np.float32)
        logits[rng.integers(0, n, size=max(1, n // 10))] = np.float32(1.5)        # ties
        if trial % 4 == 0:
            logits[rng.integers(0, n, size=n // 3)] = -np.inf
        prev = rng.integers(0, n + 3, size=int(rng.integers(0, 40))).astype(np.uint32)   # duplicates and out-of-range ids
        temp = float(rng.choice([0.0, 1.0, 0.7, 1.3]))
        k = int(rng.choice([0, 1, 5, n // 2, n, n + 7]))
        p = float(rng.choice([0.0, 0.3, 0.9, 1.0, 1.5]))
        pen = float(rng.choice([1.0, 1.2, 0.8]))
        u = int(rng.integers(0, 2**32 - 1))
        # reference order: penalty, top-k, top-p, temperature (lowest priority), then the sampler
        ref = oracle.repetition_penalty(logits, prev, pen)
        ref = oracle.top_k(ref, k) if k > 0 else ref
        ref = oracle.top_p(ref, p)
        ref = oracle.temperature(ref, temp)
        for greedy in (1, 0):
            work = logits.copy()
            prm = Params(temp, k, p, pen, prev.ctypes.data_as(up), len(prev), greedy, u, 0)
            tok = C.c_uint32()
            rc = lib.ferrum_hip_sampler_sample(work.ctypes.data_as(fp), n, C.byref(prm), C.byref(tok))
            assert np.array_equal(work, ref, equal_nan=True), trial
            if greedy:
                assert rc == 0 and tok.value == oracle.greedy_sample(ref)
            elif np.isfinite(ref).any():
                assert rc == 0 and tok.value == oracle.multinomial(ref, u), trial
            else:
                assert rc != 0                                                   # "No valid tokens for sampling"
    # greedy = last maximum, device-style argmax = first (both reference behaviours)
    tie = np.array([1.0, 7.0, 7.0, 3.0], np.float32)
    tok = C.c_uint32()
    assert lib.ferrum_hip_sampler_greedy(tie.ctypes.data_as(fp), 4, C.byref(tok)) == 0 and tok.value == 2
    assert int(oracle.argmax_rows(tie[None])[0]) == 1


def test_rust_ffi_crate_matches_the_library(pkg):
    """crates/ferrum-hip (cannot be compiled here — no Rust toolchain): src/ffi.rs is generated from
    include/ferrum_hip.h and must (a) be current, (b) declare exactly the library's exported entry points, one to one,
    and (c) every entry point the hand-written trait impls and TRAIT_MAP name must exist in the library."""
    gen = os.path.join(ROOT, "tools", "gen_ffi_rs.py")
    assert subprocess.run(["python", gen, "--check"]).returncode == 0, "ffi.rs is stale: run python tools/gen_ffi_rs.py"
    crate = os.path.join(ROOT, "crates", "ferrum-hip", "src")
    ffi = open(os.path.join(crate, "ffi.rs")).read()
    declared = re.findall(r"pub fn (ferrum_\w+)\(", ffi)
    assert len(declared) == len(set(declared))
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("ferrum_"))
    assert sorted(declared) == exported, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))
    used = set()
    for f in os.listdir(crate):
        if f.endswith(".rs") and f != "ffi.rs":
            src = open(os.path.join(crate, f)).read()
            used |= set(re.findall(r"ffi::(ferrum_\w+)", src)) | set(re.findall(r'"(ferrum_hip_\w+)"', src))
    assert len(used) > 40
    assert not (used - set(exported)), sorted(used - set(exported))
    # the #[repr(C)] structs carry every field of the C structs, in order
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ferrum_hip.h")).read(), flags=re.S)
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(\w+);", hdr, flags=re.S):
        names = []
        for decl in m.group(1).split(";"):
            decl = re.sub(r"\s+", " ", decl.strip())
            if decl:                                             # "int32_t a, b, c" / "const char* name": the last word of each declarator
                first, *rest = decl.split(",")
                names += [first.split()[-1].lstrip("*")] + [r.strip().lstrip("*") for r in rest]
        body = re.search(r"pub struct %s \{(.*?)\n\}" % m.group(2), ffi, flags=re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == names, m.group(2)


# ── the trait impls of crates/ferrum-hip against the reference's trait sources ──────────────────────────────────────────
REF = "/root/reference/crates"


def _trait_methods(path, trait):
    """(name, number of parameters, has a default body) of every fn of `pub trait <trait>` in a Rust source file."""
    src = open(path).read()
    i = src.index("pub trait " + trait)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            break
        k += 1
    body = src[j + 1:k]
    out = []
    for m in re.finditer(r"\n    (?:async )?(?:unsafe )?fn (\w+)", body):
        p, pd, params, start = m.end(), 0, None, None
        while True:                                              # the parameter list, then `;` (required) or `{` (default body)
            ch = body[p]
            if ch == "(" and pd == 0 and params is None:
                start = p
            if ch in "([<":
                pd += 1
            elif ch in ")]>" and not (ch == ">" and body[p - 1] == "-"):
                pd -= 1
                if ch == ")" and pd == 0 and params is None:
                    params = body[start + 1:p]
            elif ch == ";" and pd <= 0:
                default = False
                break
            elif ch == "{" and pd <= 0:
                default = True
                break
            p += 1
        out.append((m.group(1), _count_params(params), default))
    return out


def _count_params(params):
    params = re.sub(r"//[^\n]*", "", params)
    depth, n, cur = 0, 0, ""
    for ch in params:
        if ch in "([<{":
            depth += 1
        elif ch in ")]>}":
            depth -= 1
        if ch == "," and depth == 0:
            n += bool(cur.strip())
            cur = ""
        else:
            cur += ch
    return n + bool(cur.strip())


def _crate_fns():
    """name → set of parameter counts of every `fn` in crates/ferrum-hip/src (ffi.rs excluded)."""
    crate = os.path.join(ROOT, "crates", "ferrum-hip", "src")
    fns = {}
    for f in os.listdir(crate):
        if not f.endswith(".rs") or f == "ffi.rs":
            continue
        src = open(os.path.join(crate, f)).read()
        for m in re.finditer(r"\bfn (\w+)\s*(?:<[^>]*>)?\s*\(", src):
            p, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[p], 0)
                p += 1
            fns.setdefault(m.group(1), set()).add(_count_params(src[m.end():p - 1]))
    return fns


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_rust_crate_implements_every_required_trait_method():
    """crates/ferrum-hip cannot be compiled here (no Rust toolchain), so the next best check runs instead: the REQUIRED methods
    (no default body) of the reference's `Backend`, `BackendTimer`, `DecoderOnlyLLM` and `ModelExecutor` traits — parsed out of
    the reference's own sources — must each have an `fn` of the same name and the same number of parameters in the crate, and
    so must the defaulted methods the drop-in needs to override (paged KV, graph capture, unified forward, admission)."""
    fns = _crate_fns()
    need = {
        ("ferrum-kernels/src/backend/traits.rs", "Backend"): (),
        ("ferrum-kernels/src/backend/timer.rs", "BackendTimer"): (),
        ("ferrum-models/src/common/llm.rs", "DecoderOnlyLLM"): ("reserve_kv_slots", "kv_slot_capacity_snapshot", "unified_forward",
                                                                "unified_forward_with_logits_policy", "decode_batch"),
        ("ferrum-interfaces/src/model_executor.rs", "ModelExecutor"): ("unified_decode", "reserve_kv_slots", "kv_slot_capacity_snapshot",
                                                                       "release_cache", "supports_native_unified_decode"),
        ("ferrum-kernels/src/backend/traits.rs", "BackendPagedKv"): ("split_qkv_norm_rope_into_paged_cache_varlen", "paged_varlen_attention",
                                                                      "paged_batched_decode_attention", "paged_decode_attention"),
        ("ferrum-kernels/src/backend/capabilities.rs", "BackendGraph"): ("begin_graph_capture", "end_graph_capture", "replay_graph", "reset_graph"),
    }
    checked = 0
    for (rel, trait), also in need.items():
        methods = _trait_methods(os.path.join(REF, rel), trait)
        required = [(n, c) for n, c, d in methods if not d]
        if trait == "Backend":
            assert len(required) == 24, (len(required), [n for n, _ in required])          # the count VERDICT r2 quotes
        wanted = required + [(n, c) for n, c, d in methods if d and n in also]
        assert set(also) <= {n for n, _, _ in methods}, (trait, set(also) - {n for n, _, _ in methods})
        for name, count in wanted:
            assert name in fns, f"{trait}::{name} has no fn in crates/ferrum-hip/src"
            assert count in fns[name], f"{trait}::{name}: the reference takes {count} parameters, the crate's fn takes {sorted(fns[name])}"
            checked += 1
    assert checked >= 45, checked
