"""Time of one prefill attention launch (omni_paged_attn_prefill, the MFMA kernel) at the bench's prefill shape: 64 prompts of U{32..160}
tokens (seed 7), 16 q heads / 8 kv heads x 128, fp8 KV, block 16.  usage: python scripts/bench_prefill_attn.py [--kv fp8]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ht_vllm_omni_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--kv", default="fp8")
a = ap.parse_args()
code = {"bf16": L.KV_BF16, "fp8": L.KV_FP8, "int8": L.KV_INT8}[a.kv]
lens = np.random.default_rng(7).integers(32, 161, size=64).tolist()
T, hq, hkv, D, bs = sum(lens), 16, 8, 128, 16
nblk = sum((n + bs - 1) // bs for n in lens) + 1
g = torch.Generator().manual_seed(0)
q = torch.randn(T, hq, D, generator=g).to(torch.bfloat16).cuda()
if a.kv == "bf16":
    cache = torch.randn(2, nblk * bs, hkv, D, generator=g).to(torch.bfloat16).cuda()
else:
    cache = torch.randint(0, 120, (2, nblk * bs, hkv, D), generator=g, dtype=torch.uint8).cuda()
ks = vs = None
if a.kv == "int8":
    ks = (torch.rand(nblk * bs, hkv, generator=g) * 0.02 + 0.01).cuda(); vs = ks.clone()
bt = torch.zeros(64, 16, dtype=torch.int32)
nxt = 1
for r, n in enumerate(lens):
    for b in range((n + bs - 1) // bs):
        bt[r, b] = nxt; nxt += 1
req = torch.cat([torch.full((n,), r, dtype=torch.int32) for r, n in enumerate(lens)]).cuda()
pos = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens]).cuda()
bt = bt.cuda()
f = lambda: ops.paged_attn_prefill(q, cache[0], cache[1], bt, req, pos, q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs, kv_dtype=code,
                                   k_scale=0.05, v_scale=0.05, k_scales=ks, v_scales=vs)
out = f(); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
print(f"prefill attention, T = {T} tokens, kv {a.kv}: {best:.1f} us per launch; checksum {out.float().abs().sum().item():.6e}")
