"""The C++ continuous-batching driver (csrc/serve_loop.cc → bin/ferrum_hip_serve) over the C ABI: a closed-loop run with
chunked prefill mixed into decode batches must finish every request, hand every KV block back and be deterministic."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ferrum-infer-rs_amd", "bin", "ferrum_hip_serve")


def _run(*args):
    p = subprocess.run([BIN, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    return json.loads(lines[0]), lines[1:]


@pytest.mark.parametrize("dense", [False, True])
def test_closed_loop_chunked_prefill_mixed_with_decode(dense):
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(BIN)
    # 10 requests, 4 in flight, 40-token prompts, 6 output tokens, 48-token budget: every prompt is chunked (40 > 48 − 3
    # decode tokens once the batch is warm) and prompt chunks ride along with decode tokens of other sequences
    args = ["--layers", 2, "--requests", 10, "--concurrency", 4, "--prompt-len", 40, "--out-len", 6,
            "--max-batched-tokens", 48, "--dump-tokens"] + (["--dense"] if dense else [])
    a, toks_a = _run(*args)
    assert a["output_tokens"] == 10 * 6
    assert a["kv_blocks_free_at_exit"] == a["kv_blocks_total"]
    assert a["mixed_iterations"] >= 10 and a["iterations"] > a["mixed_iterations"]   # ≥ one forward per prompt + pure-decode steps
    assert len(toks_a) == 10 and all(len(l.split(":")[1].split()) == 6 for l in toks_a)
    b, toks_b = _run(*args)
    assert toks_a == toks_b                                   # same schedule, same kernels ⇒ same greedy ids


def test_bench_serve_shape_quick():
    """The reference's bench-serve shape (256-in/128-out, c=32) on 4 of the 48 layers: waves of hipGraph decode between
    whole-batch prefills; throughput is reported, not asserted."""
    r, _ = _run("--layers", 4, "--requests", 64, "--concurrency", 32)
    assert r["output_tokens"] == 64 * 128 and r["kv_blocks_free_at_exit"] == r["kv_blocks_total"]
    assert r["graph_decode_steps"] >= 2 * 127


def test_admission_waits_for_kv_blocks():
    """A KV pool with room for 3 whole requests under concurrency 8: reserve_kv_slots refuses (atomically) until a request
    retires; the loop must still finish everything and hand all blocks back."""
    # 40 + 6 tokens → 3 blocks per request; 10 blocks ⇒ at most 3 requests in flight
    r, _ = _run("--layers", 2, "--requests", 9, "--concurrency", 8, "--prompt-len", 40, "--out-len", 6, "--kv-blocks", 10)
    assert r["output_tokens"] == 9 * 6 and r["kv_blocks_total"] == 10 and r["kv_blocks_free_at_exit"] == 10
