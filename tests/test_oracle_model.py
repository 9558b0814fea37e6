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
