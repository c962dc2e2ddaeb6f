"""Is the skinny GEMM limited by power-of-2 row strides (channel camping)? (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from scripts.bench_ops import graph_time, BF16, dev

def run(N, K, xpad, R=8):
    Ws = [torch.randn(N, K, device=dev, dtype=BF16) * 0.02 for _ in range(R)]
    xb = torch.randn(64, K + xpad, device=dev, dtype=BF16)
    x = xb[:, :K]
    def fn():
        for i in range(R):
            M, Kk = x.shape
            out = torch.empty(M, N, dtype=BF16, device=dev)
            L.check(L.load().omni_gemm_bf16(x.data_ptr(), x.stride(0), Ws[i].data_ptr(), None, out.data_ptr(), M, N, Kk, L.EPI_BF16, None, L.current_stream()))
    us = graph_time(fn) / R
    print(f"N={N} K={K} xpad={xpad}: {us:6.2f} us  {N*K*2/us/1e6:5.2f} TB/s", flush=True)

for K in (2048, 2080, 2176):
    for xpad in (0, 128):
        run(4096, K, xpad)
run(2048, 6144, 0); run(2048, 6144 + 32, 0); run(2048, 6144 + 32, 128)
