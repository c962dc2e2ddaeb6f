"""The fused decode attention op alone in a loop on fixed inputs, first by itself and then beside the looping Code2Wav process of
tests/test_gpu_colocation.py: launches whose output differs from the first launch's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.multiprocessing as mp
from ht_vllm_omni_amd import ops, _lib as L
from tests.test_gpu_colocation import _code2wav_loop


def make(kv, ctx):
    g = torch.Generator().manual_seed(5)
    B, hq, hkv, D, bs = 64, 16, 8, 128, 16
    nb = B * (ctx // bs + 2) + 1
    cache = torch.randint(0, 120, (2, nb, bs, hkv, D), generator=g, dtype=torch.uint8).cuda()
    per = ctx // bs + 2
    bt = (torch.arange(B * per, dtype=torch.int32).view(B, per) + 1).cuda()
    seq = torch.full((B,), ctx, dtype=torch.int32).cuda()
    qkv = (torch.randn(B, (hq + 2 * hkv) * D, generator=g) * 2).to(torch.bfloat16).cuda()
    qn = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
    kn = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
    cos_sin = ops.rope_table(1024, D, 1e6).cuda()
    pos = (seq - 1).contiguous()
    return lambda: ops.attn_decode_fused(qkv, qn, kn, pos, cos_sin, cache[0], cache[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D,
                                         block_size=bs, kv_dtype=L.KV_CODES[kv], eps=1e-6, k_scale=0.5, v_scale=1.0, max_seq_len=1024,
                                         split=False)[0]


STATS = {}


def count(run, n, ref=None):
    ref = run().clone() if ref is None else ref
    bad = 0
    heads = torch.zeros(16, dtype=torch.long)
    rows = torch.zeros(ref.shape[0], dtype=torch.long)
    worst = 0.0
    for _ in range(n):
        out = run()
        if not torch.equal(out, ref):
            bad += 1
            d = (out != ref).view(ref.shape[0], 16, 128).cpu()
            heads += d.any(-1).sum(0)
            rows += d.any(-1).any(-1).long()
            worst = max(worst, float((out.float() - ref.float()).abs().max()))
    torch.cuda.synchronize()
    STATS["last"] = dict(heads=heads.tolist(), rows_hit=int((rows > 0).sum()), worst_abs=worst)
    return bad


def main():
    n = int(os.environ.get("N", 3000))
    runs = {c: make("fp8", c) for c in (24, 356)}
    refs = {}
    for c, r in runs.items():
        refs[c] = r().clone()
        print(f"ctx {c}: alone, launches differing from the first: {count(r, n, refs[c])} / {n}")
    ctx = mp.get_context("spawn")
    ready, stop, cnt = ctx.Event(), ctx.Event(), ctx.Value("i", 0)
    child = ctx.Process(target=_code2wav_loop, args=(ready, stop, cnt))
    child.start()
    try:
        assert ready.wait(300)
        for c, r in runs.items():
            print(f"ctx {c}: beside the code2wav process, launches differing from the SOLO output: {count(r, n, refs[c])} / {n}  {STATS['last']}")
    finally:
        stop.set()
        child.join(120)
    print("code2wav windows meanwhile:", cnt.value)


if __name__ == "__main__":
    main()
