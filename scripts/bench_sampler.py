"""Where does the sampler's time go? (diagnostic) back-to-back launches in a hipGraph, B = 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import ops
dev = "cuda"

def graph_time(fn, n, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3

for V in (2048, 3072):
    lg = (torch.randn(64, V, device=dev) * 3).to(torch.bfloat16).float()
    steps = torch.zeros(64, dtype=torch.int32, device=dev)
    seen = torch.zeros(64, V, dtype=torch.uint8, device=dev)
    for name, kw in (("greedy", dict(greedy=True)), ("sample k=0", dict(greedy=False, temperature=0.9, top_k=0)),
                     ("sample k=50", dict(greedy=False, temperature=0.9, top_k=50)),
                     ("sample k=50 rep", dict(greedy=False, temperature=0.9, top_k=50, rep_penalty=1.05, seen=seen)),
                     ("sample k=50 p=.8", dict(greedy=False, temperature=1.0, top_k=50, top_p=0.8))):
        us = graph_time(lambda: ops.sample(lg, steps=steps, seed=1, **kw), 16)
        print(f"V={V} {name:18s} {us:6.2f} us", flush=True)
