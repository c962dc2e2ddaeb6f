"""omni_gemm_tile vs hipBLASLt vs the correctly rounded fp64 result on the engine's own prefill operands (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ht_vllm_omni_amd import ops, _lib as L
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import frag_shuffle, gu8_shuffle
from ht_vllm_omni_amd.weights import make_weights
BF16 = torch.bfloat16
for model, T in (("tiny", 126), ("tts-1.7b", 300)):
    d = get_dims(model)
    w = make_weights(d, seed=12, std=0.06 if model == "tiny" else 0.02, norm_noise=0.1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, d.hidden, generator=g).to(BF16).cuda()
    for name in ("wqkv", "wo", "wgu", "wdown"):
        W = w["l0." + name].cuda()
        K = W.shape[1]
        a = torch.randn(T, K, generator=g).to(BF16).cuda()
        ref = (a.double() @ W.double().T)
        if name == "wgu":
            I = W.shape[0] // 2
            t = ops.gemm_tile(a, gu8_shuffle(W), act=L.TILE_ACT_SILU_MUL_GU8)
            b = ops.silu_mul(F.linear(a, W))
            ga, up = ref[:, :I].float().to(BF16).double(), ref[:, I:].float().to(BF16).double()
            r = (ga / (1 + torch.exp(-ga)) * up).float().to(BF16)
        else:
            t = ops.gemm_tile(a, frag_shuffle(W))
            b = F.linear(a, W)
            r = ref.float().to(BF16)
        print(f"{model:9s} {name:6s} N={W.shape[0]:6d} K={K:5d}: tile==rounded {float((t == r).float().mean()):.5f}  blas==rounded {float((b == r).float().mean()):.5f}  tile==blas {float((t == b).float().mean()):.5f}  "
              f"max|tile-r| {float((t.float() - r.float()).abs().max()):.4g} max|blas-r| {float((b.float() - r.float()).abs().max()):.4g}")
