"""Does the decode attention read a register or an LDS word before writing it?  The op (unfused instantiation: the same batch loop) runs
repeatedly on fixed inputs; before every launch a poison kernel fills every VGPR / AGPR of every SIMD and the whole LDS of every CU
with NaN patterns (scripts/probes/poison.hip).  Outputs must keep the bits of the unpoisoned run.
    python scripts/probes/attn_poison_probe.py            (GPU box; builds the poison library with hipcc)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ht_vllm_omni_amd import ops, _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    so = "/tmp/libpoison.so"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "poison.hip")], check=True,
                   capture_output=True)
    P = ctypes.CDLL(so)
    P.poison_launch.argtypes = [ctypes.c_void_p, ctypes.c_int]
    g = torch.Generator().manual_seed(5)
    B, hq, hkv, D, bs, ctx = 64, 16, 8, 128, 16, 356
    nb = B * (ctx // bs + 2) + 1
    for kv in ("fp8", "bf16"):
        if kv == "fp8":
            cache = torch.randint(0, 120, (2, nb, bs, hkv, D), generator=g, dtype=torch.uint8).cuda()
        else:
            cache = (torch.randn(2, nb, bs, hkv, D, generator=g) * 0.5).to(torch.bfloat16).cuda()
        per = ctx // bs + 2
        bt = (torch.arange(B * per, dtype=torch.int32).view(B, per) + 1).cuda()
        seq = torch.full((B,), ctx, dtype=torch.int32).cuda()
        q = torch.randn(B, hq * D, generator=g).to(torch.bfloat16).cuda()
        run = lambda: ops.paged_attn_decode(q, cache[0], cache[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs,
                                            kv_dtype=L.KV_CODES[kv], k_scale=0.5 if kv == "fp8" else 1.0, v_scale=1.0, max_seq_len=1024,
                                            split=False)
        if os.environ.get("FUSED", "1") == "1":        # the decode step's instantiation: q / k-norm + RoPE + KV write fused in
            qkv = (torch.randn(B, (hq + 2 * hkv) * D, generator=g) * 2).to(torch.bfloat16).cuda()
            qn = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
            kn = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
            cos_sin = ops.rope_table(1024, D, 1e6).cuda()
            pos = (seq - 1).contiguous()
            run = lambda: ops.attn_decode_fused(qkv, qn, kn, pos, cos_sin, cache[0], cache[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D,
                                                block_size=bs, kv_dtype=L.KV_CODES[kv], eps=1e-6, k_scale=0.5 if kv == "fp8" else 1.0,
                                                v_scale=1.0, max_seq_len=1024, split=False)[0]
        ref = run().clone()
        torch.cuda.synchronize()
        plain = sum(int(not torch.equal(run(), ref)) for _ in range(200))
        poisoned = 0
        worst = 0
        for _ in range(200):
            assert P.poison_launch(ctypes.c_void_p(L.current_stream()), 256) == 0
            out = run()
            if not torch.equal(out, ref):
                poisoned += 1
                worst = max(worst, int((out != ref).sum()))
        torch.cuda.synchronize()
        print(f"kv {kv}: runs differing from the first -- plain {plain} / 200, behind the poison kernel {poisoned} / 200 (most elements off in one run: {worst}; "
              f"NaNs in the last: {int(torch.isnan(out.float()).sum())})")


if __name__ == "__main__":
    main()
