"""Checkpoint reader (csrc/loader.cc) against HF-style GPTQ checkpoint directories written by tests/modelgen.py.
CPU part: shard index, dtype conversion, fused GPTQ reads, quantize-config forms, config.json mapping and the g_idx
validation cases the reference tests hold (ferrum-quantization/src/native_safetensors.rs:1549-1575).  GPU part: a model
loaded through ferrum_hip_model_load_checkpoint produces the same logits as one fed the identical arrays directly."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    p = ge.load_package()
    p.load_library()
    return p


def _tiny(moe, **kw):
    from tests import modelgen
    return modelgen.TinyModel(moe, layers=2, seed=7, **kw)


@pytest.mark.parametrize("moe,shards,fused,dtype,embedded", [(False, 1, False, "F16", False), (True, 3, False, "F32", True),
                                                            (False, 2, True, "BF16", False)])
def test_reader_roundtrip(pkg, tmp_path, moe, shards, fused, dtype, embedded):
    from tests import modelgen
    tm = _tiny(moe)
    arch = "Qwen3MoeForCausalLM" if moe else "Qwen3ForCausalLM"
    tensors = modelgen.write_checkpoint(tm, str(tmp_path), arch, shards=shards, fused_names=fused, dense_dtype=dtype,
                                        embedded_quant_config=embedded, qzeros_noise=True)
    ck = pkg.Checkpoint(str(tmp_path))
    assert ck.num_tensors() == len(tensors)
    assert ck.quant_config() == dict(is_gptq=1, bits=4, group_size=128, desc_act=0, sym=1)
    # dense tensors convert exactly (values are fp16-representable; bf16 path truncates the same way the writer did)
    emb = ck.read_f32("model.embed_tokens.weight")
    ref = tm.glob["embed"] if dtype != "BF16" else (tm.glob["embed"].view(np.uint32) & 0xFFFF0000).view(np.float32)
    assert ck.tensor_info("model.embed_tokens.weight") == (dtype, ref.shape) and np.array_equal(emb, ref)
    # fused qkv: same arrays the model holds, qzeros canonicalised to 0x77777777 although the files carry other codes
    p = "model.layers.1.self_attn."
    parts = [p + "qkv_proj"] if fused else [p + "q_proj", p + "k_proj", p + "v_proj"]
    qw, sc, qz, gi, k, n = ck.read_gptq_fused(parts)
    k0, n0, qw0, sc0, qz0 = tm.layers[1]["gptq"]["qkv"]
    assert (k, n) == (k0, n0) and gi is None
    assert np.array_equal(qw, qw0.reshape(k // 8, n)) and np.array_equal(sc, sc0.reshape(k // 128, n))
    assert np.all(qz == 0x77777777)
    raw = ck.read_i32(parts[0] + ".qzeros")
    assert np.all(raw == (0x77777777 ^ 0x11111111))
    # config mapping
    d, a, tied = ck.model_config(64)
    assert a == arch and not tied
    for key in ("num_layers", "hidden", "num_heads", "num_kv_heads", "head_dim", "vocab", "num_experts", "top_k", "expert_inter"):
        assert d[key] == tm.cfg[key], key
    assert d["has_qk_norm"] == 1 and d["max_seq_len"] == 64 and d["group_size"] == 128
    assert d["intermediate"] == (0 if moe else tm.cfg["intermediate"])
    with pytest.raises(RuntimeError, match="not in index"):
        ck.read_f32("model.layers.9.nope")


def test_config_mapping_defaults_and_rope_scaling(pkg, tmp_path):
    """Field extraction of definition.rs:225-375 / llama_family.rs:596-680,768-810 on hand-written config.json files."""
    from safetensors.numpy import save_file
    save_file({"lm_head.weight": np.zeros((4, 4), np.float16)}, str(tmp_path / "model.safetensors"))

    def cfg(j):
        json.dump(j, open(tmp_path / "config.json", "w"))
        return pkg.Checkpoint(str(tmp_path)).model_config(0)

    d, arch, tied = cfg({"architectures": ["LlamaForCausalLM"], "hidden_size": 4096, "num_attention_heads": 32,
                         "num_hidden_layers": 32, "vocab_size": 128256, "intermediate_size": 14336, "num_key_value_heads": 8,
                         "max_position_embeddings": 131072, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                         "rope_scaling": {"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                          "original_max_position_embeddings": 8192, "rope_type": "llama3"}})
    assert (d["head_dim"], d["num_kv_heads"], d["rope_scaling_kind"]) == (128, 8, 2) and not tied
    assert (d["rope_p0"], d["rope_p1"], d["rope_p2"], d["rope_p3"]) == (8.0, 1.0, 4.0, 8192.0)
    assert d["max_seq_len"] == 131072 and abs(d["rms_eps"] - 1e-5) < 1e-12 and d["has_qk_norm"] == 0
    # defaults: kv heads = heads, head_dim = hidden / heads, llama rope theta 5e5, eps 1e-6; invalid llama3 block ignored
    d, _, _ = cfg({"architectures": ["LlamaForCausalLM"], "hidden_size": 512, "num_attention_heads": 8,
                   "rope_scaling": {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 4.0, "high_freq_factor": 1.0}})
    assert (d["num_kv_heads"], d["head_dim"], d["rope_theta"], d["rope_scaling_kind"]) == (8, 64, 500000.0, 0)
    assert abs(d["rms_eps"] - 1e-6) < 1e-12 and d["num_layers"] == 32 and d["intermediate"] == 11008
    d, _, _ = cfg({"architectures": ["MistralForCausalLM"], "hidden_size": 512, "num_attention_heads": 8, "sliding_window": 4096,
                   "rope_scaling": {"type": "linear", "factor": 2.0}})
    assert (d["rope_theta"], d["sliding_window"], d["rope_scaling_kind"], d["rope_p0"]) == (10000.0, 4096, 1, 2.0)
    d, _, _ = cfg({"architectures": ["MistralForCausalLM"], "hidden_size": 512, "num_attention_heads": 8, "sliding_window": None})
    assert d["sliding_window"] == 0
    d, _, _ = cfg({"architectures": ["Qwen3MoeForCausalLM"], "hidden_size": 2048, "num_attention_heads": 32, "head_dim": 128,
                   "num_key_value_heads": 4, "num_experts": 128, "moe_intermediate_size": 768, "vocab_size": 151936,
                   "num_hidden_layers": 48})
    assert (d["num_experts"], d["top_k"], d["expert_inter"], d["norm_topk_prob"], d["intermediate"]) == (128, 8, 768, 1, 0)
    assert (d["head_dim"], d["has_qk_norm"], d["rope_theta"]) == (128, 1, 1000000.0)
    with pytest.raises(RuntimeError, match="missing num_experts"):
        cfg({"architectures": ["Qwen3MoeForCausalLM"], "moe_intermediate_size": 768})
    with pytest.raises(RuntimeError, match="not supported"):
        cfg({"architectures": ["GPT2LMHeadModel"]})
    # Gemma 3 (gemma3_from_def, llama_family.rs:683-703), nested text_config flattened over the root
    d, arch, _ = cfg({"architectures": ["Gemma3ForConditionalGeneration"], "model_type": "gemma3",
                      "text_config": {"hidden_size": 1152, "num_attention_heads": 4, "num_key_value_heads": 1, "head_dim": 256,
                                      "num_hidden_layers": 26, "intermediate_size": 6912, "vocab_size": 262144,
                                      "hidden_activation": "gelu_pytorch_tanh", "sliding_window": 512,
                                      "query_pre_attn_scalar": 256, "rope_scaling": {"rope_type": "linear", "factor": 8.0}}})
    assert (d["sandwich_norms"], d["sliding_window_pattern"], d["rope_local_theta"], d["activation"]) == (1, 6, 10000.0, 1)
    assert (d["has_qk_norm"], d["sliding_window"], d["rope_theta"], d["rope_scaling_kind"], d["head_dim"]) == (1, 512, 1e6, 1, 256)
    es = np.float32(np.sqrt(1152.0))                        # bf16_round(√hidden): low mantissa bits cleared (llama_family.rs:5951)
    assert np.float32(d["embed_scale"]).view(np.uint32) & 0xFFFF == 0 and abs(d["embed_scale"] - es) < 0.25
    with pytest.raises(RuntimeError, match="no safetensors"):
        pkg.Checkpoint(str(tmp_path / "missing"))


def test_g_idx_validation_cases(pkg, tmp_path):
    """native_safetensors.rs:1288-1324 and its tests :1549-1575: desc_act without g_idx is an error; the trivial order is not
    act-order; anything else is, and is handed on; out-of-range and wrong-length g_idx are rejected; parts must agree."""
    from safetensors.numpy import save_file
    k, n = 256, 16
    rng = np.random.default_rng(0)

    def write(desc_act, g_a=None, g_b=None):
        t = {}
        for stem, g in (("a", g_a), ("b", g_b)):
            t[stem + ".qweight"] = rng.integers(-2**31, 2**31 - 1, size=(k // 8, n), dtype=np.int64).astype(np.int32)
            t[stem + ".scales"] = rng.random((k // 128, n)).astype(np.float16)
            t[stem + ".qzeros"] = np.full((k // 128, n // 8), 0x77777777, np.int32)
            if g is not None:
                t[stem + ".g_idx"] = np.asarray(g, np.int32)
        save_file(t, str(tmp_path / "model.safetensors"))
        json.dump({"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": desc_act, "sym": True},
                  open(tmp_path / "quantize_config.json", "w"))
        return pkg.Checkpoint(str(tmp_path))

    trivial = np.arange(k) // 128
    with pytest.raises(RuntimeError, match="desc_act=true but no g_idx"):
        write(True).read_gptq_fused(["a"])
    assert write(False, trivial).read_gptq_fused(["a"])[3] is None            # trivial order: not act-order
    perm = trivial[::-1].copy()
    got = write(False, perm, perm).read_gptq_fused(["a", "b"])
    assert np.array_equal(got[3], perm) and got[5] == 2 * n                   # nontrivial: detected and handed on
    with pytest.raises(RuntimeError, match="outside expected group range"):
        write(False, np.full(k, 2)).read_gptq_fused(["a"])
    with pytest.raises(RuntimeError, match="g_idx shape"):
        write(False, trivial[:100]).read_gptq_fused(["a"])
    with pytest.raises(RuntimeError, match="g_idx mismatch"):
        write(False, perm, trivial).read_gptq_fused(["a", "b"])
    with pytest.raises(RuntimeError, match="all parts to carry g_idx"):
        write(False, perm, None).read_gptq_fused(["a", "b"])


@pytest.mark.gpu
@pytest.mark.parametrize("moe,arch,kw", [(True, "Qwen3MoeForCausalLM", {}),
                                         (False, "Qwen2ForCausalLM", dict(qk_norm=False, qkv_bias=True)),
                                         (False, "LlamaForCausalLM", dict(qk_norm=False, rope_theta=500000.0, rope_scaling_kind=2,
                                                                          rope_p=(8.0, 1.0, 4.0, 64.0), tied=True))])
def test_model_loaded_from_checkpoint_equals_direct_load(pkg, tmp_path, moe, arch, kw):
    from tests import modelgen
    tm = _tiny(moe, **kw)
    modelgen.write_checkpoint(tm, str(tmp_path), arch, shards=2)
    direct = tm.hip_model(pkg, kv_num_blocks=16, max_seqs=4, max_tokens=64)
    loaded = pkg.HipModel.from_checkpoint(str(tmp_path), kv_num_blocks=16, max_seqs=4, max_tokens=64)
    rng = np.random.default_rng(3)
    prompt = rng.integers(0, tm.cfg["vocab"], size=21).astype(np.uint32)
    t1, l1 = direct.unified_forward([(1, prompt, 0, True)], greedy=True, want_logits=True)
    t2, l2 = loaded.unified_forward([(1, prompt, 0, True)], greedy=True, want_logits=True)
    assert np.array_equal(l1, l2) and np.array_equal(t1, t2)
    om = tm.oracle_model()
    assert modelgen.cosine(om.forward(0, prompt, 0), l2[0]) > 0.999


@pytest.mark.gpu
def test_gemma3_checkpoint_folds_norm_weights(pkg, tmp_path):
    """Gemma-3 checkpoints store RMSNorm weights as w − 1 and expect q_norm × √(head_dim/query_pre_attn_scalar)
    (fold_norm_weight, llama_family.rs:891-967).  A model loaded from such a checkpoint equals a model handed the folded
    arrays directly, and both match the oracle."""
    from tests import modelgen
    kw = dict(activation=1, sandwich=True, sliding_window=8, sliding_window_pattern=2, rope_local_theta=10000.0,
              rope_theta=1e6, rope_scaling_kind=1, rope_p=(8.0, 0.0, 0.0, 0.0))
    stored = modelgen.TinyModel(False, layers=2, seed=71, **kw)           # what the files hold (norms = w − 1 convention)
    qpas = 256.0
    q_scale = np.float32(np.sqrt(stored.cfg["head_dim"] / qpas))
    es = np.float32(np.sqrt(stored.cfg["hidden"])).view(np.uint32)
    es = np.uint32((int(es) + 0x7FFF + ((int(es) >> 16) & 1)) & 0xFFFF0000).view(np.float32)
    eff = modelgen.TinyModel(False, layers=2, seed=71, embed_scale=float(es), **kw)   # what the runner must end up with
    fold = lambda w, s=np.float32(1.0): ((w.astype(np.float32) + np.float32(1.0)) * s).astype(np.float32)
    eff.glob["final_norm"] = fold(stored.glob["final_norm"])
    for Ls, Le in zip(stored.layers, eff.layers):
        for key in Ls["dense"]:
            Le["dense"][key] = fold(Ls["dense"][key], q_scale if key == "q_norm" else np.float32(1.0))
    modelgen.write_checkpoint(stored, str(tmp_path), "Gemma3ForConditionalGeneration",
                              gemma=dict(query_pre_attn_scalar=qpas, nested=True))
    loaded = pkg.HipModel.from_checkpoint(str(tmp_path), kv_num_blocks=16, max_seqs=4, max_tokens=64)
    direct = eff.hip_model(pkg, kv_num_blocks=16, max_seqs=4, max_tokens=64)
    rng = np.random.default_rng(4)
    prompt = rng.integers(0, stored.cfg["vocab"], size=29).astype(np.uint32)
    _, l1 = direct.unified_forward([(1, prompt, 0, True)], greedy=True, want_logits=True)
    _, l2 = loaded.unified_forward([(1, prompt, 0, True)], greedy=True, want_logits=True)
    assert np.array_equal(l1, l2)
    assert modelgen.cosine(eff.oracle_model().forward(0, prompt, 0), l2[0]) > 0.999
