"""Does a weight matrix that sits in the Infinity Cache (MALL) stream faster than one from HBM? (diagnostic)
Rotates R weight buffers per shape: R * bytes ~ 130 MB (fits the 256 MB MALL, far exceeds the 8 x 4 MB L2s) vs ~1 GB."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from ht_vllm_omni_amd.engine import frag_shuffle
dev, BF16 = "cuda", torch.bfloat16

def graph_time(fn, reps=8):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def run(name, N, K, epi, total_mb):
    rows = 2 * N if epi == L.EPI_SILU_MUL else N
    wb = rows * K * 2
    R = max(1, int(total_mb * 1e6 // wb))
    Ws = [frag_shuffle(torch.randn(rows, K, device=dev, dtype=BF16) * 0.02) for _ in range(R)]
    x = frag_shuffle(torch.randn(64, K, device=dev, dtype=BF16))
    out = torch.empty(64, N, device=dev, dtype=BF16 if epi != L.EPI_F32_BF16RND else torch.float32)
    lib = L.load()
    lay = L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG
    n_launch = max(R, 16)
    def fn():
        for i in range(n_launch):
            L.check(lib.omni_gemm_bf16_ex(L.ptr(x), K, L.ptr(Ws[i % R]), None, L.ptr(out), 64, N, K, epi, None, lay, L.current_stream()), "gemm")
    us = graph_time(fn) / n_launch
    print(f"{name:12s} {wb/1e6:5.1f} MB x {R:3d} = {R*wb/1e6:6.0f} MB rotating: {us:6.2f} us  {wb/us/1e6:5.2f} TB/s", flush=True)

for tot in (1000, 130, 24):
    run("bb qkv", 4096, 2048, L.EPI_BF16, tot)
    run("bb gate_up", 6144, 2048, L.EPI_SILU_MUL, tot)
    run("bb down", 2048, 6144, L.EPI_BF16, tot)
    run("bb o", 2048, 2048, L.EPI_BF16, tot)
