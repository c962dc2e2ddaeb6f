"""Sparse-MoE block of the Omni talker at its real shape, decode batch 64 (diagnostic): time and bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import ops
from ht_vllm_omni_amd.engine import frag_shuffle
BF16 = torch.bfloat16
H, E, K, I, Is = 1024, 128, 8, 384, 768
R = 3
ws = []
for r in range(R):
    g = torch.Generator(device="cuda").manual_seed(r)
    rnd = lambda *s, sc: (torch.randn(*s, device="cuda", generator=g) * sc).to(BF16)
    ws.append({"router": rnd(E, H, sc=0.5), "gate_up_f": frag_shuffle(rnd(E, 2 * I, H, sc=0.08)), "down_f": frag_shuffle(rnd(E, H, I, sc=0.08)),
               "shared_gate_up": rnd(2 * Is, H, sc=0.08), "shared_down": rnd(H, Is, sc=0.08), "shared_gate": rnd(1, H, sc=0.3)})
for T in (64, 16, 1):
    x = torch.randn(T, H, device="cuda").to(BF16)
    def fn():
        for r in range(R): ops.moe_block(x, ws[r], K)
    fn(); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph): fn()
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 / R * 1e3
    _, idx, _ = ops.moe_block(x, ws[0], K)
    hit = idx.unique().numel()
    byt = hit * (2 * I * H + H * I) * 2 + (2 * Is * H + H * Is + E * H) * 2
    print(f"T={T:3d}: {us:7.1f} us per block ({7 if True else 0} launches), experts hit {hit}/{E}, {byt/1e6:6.1f} MB -> {byt/us/1e6:5.2f} TB/s", flush=True)
