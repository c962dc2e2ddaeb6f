"""Small-prompt prefill: omni_gemm_tile vs hipBLASLt per call (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd import ops
d = get_dims("tts-1.7b")
w = make_weights(d, seed=1, std=0.02, device="cuda")
eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=256, max_batch=1, prefill_gemm="both")
T = 100
x = (torch.randn(T, d.hidden, device="cuda") * 0.05).to(torch.bfloat16)
pos = torch.arange(T, dtype=torch.int32, device="cuda")
req = torch.zeros(T, dtype=torch.int32, device="cuda")
eng.block_table[0, :8] = torch.arange(1, 9, dtype=torch.int32)
slots = (16 + torch.arange(T)).cuda()
for g in ("tile", "blas", "tile", "blas"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.prefill(x, pos, req, slots, use_blas=True, gemm=g)
    torch.cuda.synchronize(); print(g, "prefill ms", (time.perf_counter() - t0) * 1e3)
lw = eng.layer_w[0]
a = x
for name, fn in (("qkv tile", lambda: ops.gemm_tile(a, lw["wqkv_f"])), ("qkv blas", lambda: torch.nn.functional.linear(a, lw["wqkv"]))):
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(name, i, (time.perf_counter() - t0) * 1e3, "ms")
