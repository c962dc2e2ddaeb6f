import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from oracle import talker_oracle as O
BF16 = torch.bfloat16
D, bs, nb, hq, hkv = 128, 16, 300, 16, 8
g = torch.Generator().manual_seed(0)
lens = torch.randint(4, 40, (64,), generator=g).tolist()
pk = O.PagedKV(nb, bs, hkv, D, "fp8", 1.0, 1.0)
k = (torch.randn(nb * bs, hkv, D, generator=g) * 0.5).to(BF16); v = (torch.randn(nb * bs, hkv, D, generator=g) * 0.5).to(BF16)
pk.write(torch.arange(nb * bs), k, v)
bt = torch.zeros(64, 16, dtype=torch.int32); nxt = 1
for r, n in enumerate(lens):
    need = (n + 4 + bs - 1) // bs
    bt[r, :need] = torch.arange(nxt, nxt + need, dtype=torch.int32); nxt += need
req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32)
pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32)
T = req.numel()
q = (torch.randn(T, hq * D, generator=g) * 0.3).to(BF16)
cache = pk.data.view(torch.uint8).cuda()
import ctypes
lib = L.load(); lib.omni_debug_prefill_mfma.argtypes = [ctypes.c_int]; lib.omni_debug_prefill_mfma.restype = None
outs = {}
for on in (0, 1):
    lib.omni_debug_prefill_mfma(on)
    outs[on] = ops.paged_attn_prefill(q.cuda(), cache[0], cache[1], bt.cuda(), req.cuda(), pos.cuda(), q_heads=hq, kv_heads=hkv, head_dim=D,
                             block_size=bs, kv_dtype=L.KV_FP8).cpu().float().view(T, hq, D)
print("valu vs mfma mean abs", (outs[0] - outs[1]).abs().mean().item(), "nonzero frac", ((outs[0] - outs[1]) != 0).float().mean().item())
refs = []
o = 0
for r, n in enumerate(lens):
    kk, vv = pk.gather(bt[r].tolist(), n)
    refs.append(O.attention_rows(q[o:o + n].view(n, hq, D), kk, vv, torch.arange(n), D ** -0.5).float()); o += n
ref_all = torch.cat(refs)
for on in (0, 1):
    dd = (outs[on] - ref_all).abs()
    print("mfma" if on else "valu", "vs oracle: mean abs", dd.mean().item(), "nonzero frac", (dd != 0).float().mean().item())
out = ops.paged_attn_prefill(q.cuda(), cache[0], cache[1], bt.cuda(), req.cuda(), pos.cuda(), q_heads=hq, kv_heads=hkv, head_dim=D,
                             block_size=bs, kv_dtype=L.KV_FP8).cpu().float().view(T, hq, D)
o = 0; worst = []
for r, n in enumerate(lens):
    kk, vv = pk.gather(bt[r].tolist(), n)
    ref = O.attention_rows(q[o:o + n].view(n, hq, D), kk, vv, torch.arange(n), D ** -0.5).float()
    e = (out[o:o + n] - ref).abs().amax(dim=(1, 2))
    for i in range(n): worst.append((e[i].item(), r, i, o + i))
    o += n
worst.sort(reverse=True)
print("T", T, "worst rows (err, req, pos, flat):", [(round(a, 5), b, c, d_, d_ % 16) for a, b, c, d_ in worst[:12]])
print("mean row max err", sum(w[0] for w in worst) / len(worst))
