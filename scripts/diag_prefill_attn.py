"""Error of the MFMA prefill attention and of the per-token VALU decode kernel against an fp64 reference (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from oracle import talker_oracle as O
BF16 = torch.bfloat16
D, bs, nb, hq, hkv = 128, 16, 64, 16, 8
g = torch.Generator().manual_seed(5)
for kv, qscale in (("bf16", 1.0), ("fp8", 1.0), ("bf16", 0.05), ("fp8", 0.05)):
    pk = O.PagedKV(nb, bs, hkv, D, kv, 1.0, 1.0)
    k = (torch.randn(nb * bs, hkv, D, generator=g) * 1.5).to(BF16); v = (torch.randn(nb * bs, hkv, D, generator=g) * 1.5).to(BF16)
    pk.write(torch.arange(nb * bs), k, v)
    n = 150
    bt = torch.arange(1, 1 + (n + bs - 1) // bs, dtype=torch.int32).view(1, -1)
    q = (torch.randn(n, hq * D, generator=g) * qscale).to(BF16)
    store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    cache = store.cuda()
    req = torch.zeros(n, dtype=torch.int32).cuda(); pos = torch.arange(n, dtype=torch.int32).cuda()
    out_m = ops.paged_attn_prefill(q.cuda(), cache[0], cache[1], bt.cuda(), req, pos, q_heads=hq, kv_heads=hkv, head_dim=D,
                                   block_size=bs, kv_dtype=L.KV_CODES[kv]).cpu().float().view(n, hq, D)
    # per-token decode kernel: row t with seq_len t+1
    btn = bt.repeat(n, 1).contiguous().cuda()
    out_d = torch.cat([ops.paged_attn_decode(q[i:i + 50].cuda(), cache[0], cache[1], btn[i:i + 50].contiguous(),
                                             (pos[i:i + 50] + 1).contiguous(), q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs,
                                             kv_dtype=L.KV_CODES[kv], max_seq_len=256, split=False) for i in range(0, n, 50)]).cpu().float().view(n, hq, D)
    kk, vv = pk.gather(bt[0].tolist(), n)
    qd = q.view(n, hq, D).double(); kd = kk.double().repeat_interleave(hq // hkv, 1); vd = vv.double().repeat_interleave(hq // hkv, 1)
    s = torch.einsum("thd,shd->hts", qd, kd) * D ** -0.5
    s = s.masked_fill(torch.arange(n)[None, :] > torch.arange(n)[:, None], float("-inf"))
    ref = torch.einsum("hts,shd->thd", torch.softmax(s, -1), vd)
    orc = O.attention_rows(q.view(n, hq, D), kk, vv, torch.arange(n), D ** -0.5).double()
    for name, o in (("mfma", out_m), ("valu-decode", out_d), ("oracle-fp32", orc)):
        e = (o.double() - ref).abs()
        print(kv, qscale, name, "mean abs err %.3e max %.3e" % (e.mean().item(), e.max().item()),
              "vs oracle mean %.3e" % (o.double() - orc).abs().mean().item())
