"""Is the decode step's sampler launch slow because its CODE is cold? (diagnostic; run under rocprofv3 --kernel-trace --stats)
The talker row's sampler (V = 3072, top-k 50, T 0.9, repetition penalty) back to back in one graph = hot instruction cache and L2, against
the same launch behind a 1 GiB copy that evicts L2 and the Infinity Cache (the step's position: once per 3.3 ms behind 700 MB of weights).
The kernel trace separates the two by the number of rows: 64 hot, 63 cold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import ops
dev = "cuda"
V = 3072
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
big2 = torch.empty_like(big)
for B, cold in ((64, False), (63, True)):
    lg = (torch.randn(B, V, device=dev) * 3).to(torch.bfloat16).float()
    steps = torch.zeros(B, dtype=torch.int32, device=dev)
    seen = (torch.rand(B, V, device=dev) < 0.05).to(torch.uint8)
    fn = lambda: ops.sample(lg, steps=steps, seed=1, greedy=False, temperature=0.9, top_k=50, rep_penalty=1.05, seen=seen)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            if cold:
                big2.copy_(big)
            fn()
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
print("done")
