import torch, sys
E, cap, H, N = 128, 512, 1024, 768
A = torch.randn(E, cap, H, device="cuda").to(torch.bfloat16)
W = torch.randn(E, N, H, device="cuda").to(torch.bfloat16)
for name, fn in (("bmm contiguous B", lambda: torch.bmm(A, W.transpose(1, 2).contiguous())),
                 ("einsum", lambda: torch.einsum("ech,enh->ecn", A, W)),
                 ("bmm transposed view", lambda: torch.bmm(A, W.transpose(1, 2)))):
    try:
        o = fn(); torch.cuda.synchronize(); print("ok", name, o.shape, flush=True)
    except Exception as e:
        print("fail", name, repr(e)[:200], flush=True)
