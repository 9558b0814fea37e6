import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, bench
import __graft_entry__ as ge
pkg=ge.load_package(); pkg.load_library()
cfg=dict(bench.MODELS["qwen3-30b-a3b"])
model=bench.build_model(pkg,cfg,32,512,2048,9271)
rng=np.random.default_rng(1)
prompts=[rng.integers(256,cfg["vocab"],size=256).astype(np.uint32) for _ in range(32)]
first=bench.prefill(model,prompts,0,2048)
model.decode_steps(list(range(32)), first, 4)
for name in ("moe_gate_up","moe_down","attention","qkv","o"):
    us,_=model.time_kernel(name,32,270,reps=3)
    print(name, round(us,2))
