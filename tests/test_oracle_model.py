"""CPU-only checks of the model-level oracle (ferrum_oracle_model.c): internal consistency properties
of the reference's CPU path (prefill ≡ prefill-prefix + decode; chunked prefill ≡ whole prefill) and the
sharding identities the TP lane relies on.  No GPU, no reference weights (none exist offline)."""
import numpy as np
import pytest

from tests import modelgen


@pytest.mark.parametrize("moe", [False, True])
def test_incremental_decode_equals_full_prefill(oracle, moe):
    tm = modelgen.TinyModel(moe, layers=2, seed=1)
    om = tm.oracle_model()
    rng = np.random.default_rng(2)
    toks = rng.integers(0, tm.cfg["vocab"], size=12).astype(np.uint32)
    last, all_logits = om.forward(0, toks, 0, all_logits=True)
    assert np.array_equal(last, all_logits[-1])
    # same sequence fed as 7-token prefill + 5 decode steps in another cache
    om.forward(1, toks[:7], 0)
    for i in range(7, 12):
        step = om.forward(1, toks[i:i + 1], i)
        assert np.max(np.abs(step - all_logits[i])) < 2e-4 * max(1.0, np.max(np.abs(all_logits[i])))
    # chunked prefill (4 + 8) in a third cache
    om.forward(2, toks[:4], 0)
    chunk = om.forward(2, toks[4:], 4)
    assert np.max(np.abs(chunk - last)) < 2e-4 * max(1.0, np.max(np.abs(last)))
    for is_v in (0, 1):
        assert np.allclose(om.read_kv(0, 1, is_v), om.read_kv(1, 1, is_v), atol=1e-4)


def test_forward_rejects_wrong_pos_offset(oracle):
    tm = modelgen.TinyModel(False, layers=1, seed=3)
    om = tm.oracle_model()
    with pytest.raises(RuntimeError):
        om.forward(0, np.array([1, 2], np.uint32), 5)


@pytest.mark.parametrize("moe", [False, True])
def test_worker_threads_do_not_change_a_single_bit(oracle, moe):
    """The oracle's row / head / token loops may run on several threads in the real-dimension GPU parity cases
    (oracle.set_threads): every output element is still computed by one thread in the reference's loop order, so the
    logits, the KV cache and the stand-alone GEMM / attention / MoE ops must be bit-identical to the 1-thread run."""
    tm = modelgen.TinyModel(moe, layers=2, seed=5, hidden=512, nq=8, nkv=2, inter=1024, experts=16, top_k=4, vocab=4096)
    rng = np.random.default_rng(6)
    toks = rng.integers(0, tm.cfg["vocab"], size=70).astype(np.uint32)
    outs = []
    for threads in (1, 5):
        oracle.set_threads(threads)
        assert oracle.get_threads() == threads
        om = tm.oracle_model()
        last, allg = om.forward(0, toks, 0, all_logits=True)
        step = om.forward(0, toks[:1], 70)
        a = rng.standard_normal((37, 2048)).astype(np.float32) if threads == 1 else a
        b = rng.standard_normal((300, 2048)).astype(np.float32) if threads == 1 else b
        outs.append((last, allg, step, om.read_kv(0, 1, 0), om.read_kv(0, 1, 1), oracle.gemm(a, b, 37, 300, 2048)))
    oracle.set_threads(1)
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
