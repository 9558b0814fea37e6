import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.load_library()
from tests import modelgen
tm = modelgen.TinyModel(True, layers=3, seed=5, experts=32, top_k=4, max_seq_len=1024)
rng = np.random.default_rng(0)
prompts = [rng.integers(0, 512, size=n).astype(np.uint32) for n in (40, 3, 17, 64, 1, 9, 33, 25)]
outs = []
for mode in ("graph", "eager"):
    if mode == "eager": os.environ["FERRUM_HIP_NO_GRAPH"] = "1"
    hm = tm.hip_model(pkg, kv_num_blocks=8 * 64 + 8, max_seqs=8, max_tokens=256)
    first, _ = hm.unified_forward([(i, p, 0, True) for i, p in enumerate(prompts)], greedy=True)
    a = hm.decode_steps(list(range(8)), first, 300)
    b = hm.decode_steps(list(range(8)), a[-1], 300)     # second run: graph reuse with a different kv bucket
    outs.append(np.concatenate([a, b]))
print("graph == eager over 600 steps:", np.array_equal(outs[0], outs[1]), outs[0].shape)

# the merged launches (decode chain + one-launch MoE pair) at Qwen3-30B-A3B dims, three layers, c = 32 and c = 3: two fresh
# instances per mode sample the same 96 ids per row, graph ≡ eager — in-launch hand-offs leave no run-to-run freedom
import bench
for c in (32, 3):
    cfg = dict(bench.QWEN3_30B_A3B)
    cfg["vocab"] = 4096
    runs = []
    for mode in ("graph", "graph", "eager"):
        if mode == "eager": os.environ["FERRUM_HIP_NO_GRAPH"] = "1"
        else: os.environ.pop("FERRUM_HIP_NO_GRAPH", None)
        model = bench.build_model(pkg, cfg, c, 64 + 128, c * 64, 1234, layers=3)
        r2 = np.random.default_rng(77)
        pr = [r2.integers(256, 4096, size=64).astype(np.uint32) for _ in range(c)]
        first = bench.prefill(model, pr, 0, c * 64)
        a = model.decode_steps(list(range(c)), np.array(first, np.uint32), 48)
        b = model.decode_steps(list(range(c)), a[-1], 48)
        runs.append(np.concatenate([a, b]))
        del model
    print(f"merged launches, c={c}: repeat identical {np.array_equal(runs[0], runs[1])}, graph == eager {np.array_equal(runs[0], runs[2])}", runs[0].shape)
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])

# KV ranges inside the chain's attention role (ticket merge by the last arriver, in range order): long ragged contexts, the range
# count changing as the kv bucket grows — repeats identical, graph ≡ eager
for c, lens in ((3, (1500, 700, 2300)), (12, tuple(600 + 97 * i for i in range(12)))):
    cfg = dict(bench.QWEN3_30B_A3B)
    cfg["vocab"] = 4096
    runs, forms = [], None
    for mode in ("graph", "graph", "eager"):
        if mode == "eager": os.environ["FERRUM_HIP_NO_GRAPH"] = "1"
        else: os.environ.pop("FERRUM_HIP_NO_GRAPH", None)
        model = bench.build_model(pkg, cfg, c, max(lens) + 160, sum(lens), 4321, layers=3)
        r2 = np.random.default_rng(78)
        pr = [r2.integers(256, 4096, size=n).astype(np.uint32) for n in lens]
        first, _ = model.unified_forward([(i, p, 0, True) for i, p in enumerate(pr)], greedy=True)
        a = model.decode_steps(list(range(c)), np.array(first, np.uint32), 70)
        b = model.decode_steps(list(range(c)), a[-1], 70)
        runs.append(np.concatenate([a, b]))
        del model
    import ctypes as C
    lib = pkg.load_library()
    lib.ferrum_hip_debug_form_name.restype = C.c_char_p
    names = [lib.ferrum_hip_debug_form_name(i).decode() for i in range(lib.ferrum_hip_debug_form_count())]
    arr = (C.c_uint64 * len(names))()
    lib.ferrum_hip_debug_form_hits(arr, len(names))
    assert dict(zip(names, arr)).get("chain_attn_kv_splits", 0) > 0, "the range form did not run"
    print(f"chain KV ranges, c={c}, contexts {min(lens)}..{max(lens)} + 140 steps: repeat identical {np.array_equal(runs[0], runs[1])}, graph == eager {np.array_equal(runs[0], runs[2])}", runs[0].shape)
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
