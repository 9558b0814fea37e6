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
