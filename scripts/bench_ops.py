"""Per-op micro-benchmarks inside hipGraphs with rotating (cache-defeating) buffers (diagnostic)."""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import ctypes as C, os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
lib = L.load()
lib.omni_debug_set.argtypes = [C.c_int, C.c_int, C.c_int]; lib.omni_debug_set.restype = None
dev = "cuda"
BF16 = torch.bfloat16

def graph_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us per graph

def bench_gemm(name, N, K, epi, M=64, hot=False):
    rows = 2 * N if epi == L.EPI_SILU_MUL else N
    wbytes = rows * K * 2
    R = 1 if hot else max(2, min(64, int(1.2e9 // wbytes)))
    Ws = [torch.randn(rows, K, device=dev, dtype=BF16) * 0.02 for _ in range(R)]
    x = torch.randn(M, K, device=dev, dtype=BF16)
    n_launch = R if not hot else 20
    frag = os.environ.get("OMNI_BENCH_FRAG", "1") == "1"
    lay = 0
    if frag:
        from ht_vllm_omni_amd.engine import frag_shuffle
        Ws = [frag_shuffle(w_) for w_ in Ws]
        xp = torch.zeros((M + 15) // 16 * 16, K, device=dev, dtype=BF16); xp[:M] = x
        x = frag_shuffle(xp)
        lay = L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG
    def fn():
        for i in range(n_launch):
            ops.gemm(x, Ws[i % R], epilogue=epi, layout=lay, M=M)
    res = []
    for nt, wgs in ((5, 256), (5, 128), (5, 512), (4, 256)):   # nt bit0 = non-temporal W, bit1 = static-K, bit2 = silu NT=2
        lib.omni_debug_set(nt, 0, wgs)
        us = graph_time(fn) / n_launch
        res.append(f"nt={nt} wgs={wgs}: {us:6.2f} us {wbytes / us / 1e6:5.2f} TB/s")
    lib.omni_debug_set(1, 0, 256)
    print(f"{name:22s} M={M} N={N:5d} K={K:5d} {wbytes/1e6:6.1f} MB {'hot' if hot else 'cold'} | " + " | ".join(res), flush=True)

def bench_norm(rows, hidden):
    R = 32
    xs = [torch.randn(rows, hidden, device=dev, dtype=BF16) for _ in range(R)]
    ds = [torch.randn(rows, hidden, device=dev, dtype=BF16) for _ in range(R)]
    w = torch.ones(hidden, device=dev, dtype=BF16)
    def fn():
        for i in range(R): ops.rmsnorm(None, w, 1e-6, delta=ds[i], residual=xs[i])
    print(f"rmsnorm+resid {rows}x{hidden}: {graph_time(fn)/R:6.2f} us", flush=True)

def bench_attn(name, B, hq, hkv, ctx, kv, layers, fused):
    D, bs = 128, 16
    nblk = B * ((ctx + bs) // bs + 1) + 1
    store = torch.uint8 if kv == "fp8" else BF16
    caches = [torch.randint(0, 100, (2, nblk, bs, hkv, D), device=dev, dtype=torch.uint8).view(store) if kv == "fp8"
              else torch.randn(2, nblk, bs, hkv, D, device=dev, dtype=BF16) for _ in range(layers)]
    per = (ctx + bs) // bs + 1
    bt = (torch.arange(B * per, dtype=torch.int32).view(B, per) + 1).to(dev)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    pos = seq - 1
    q = torch.randn(B, hq * D, device=dev, dtype=BF16)
    qkv = torch.randn(B, (hq + 2 * hkv) * D, device=dev, dtype=BF16)
    nw = torch.ones(D, device=dev, dtype=BF16)
    cs = ops.rope_table(4096, D, 1e6).to(dev)
    code = L.KV_CODES[kv]
    def fn():
        for c in caches:
            if fused:
                ops.attn_decode_fused(qkv, nw, nw, pos, cs, c[0], c[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D,
                                      block_size=bs, kv_dtype=code, eps=1e-6, max_seq_len=4096)
            else:
                ops.paged_attn_decode(q, c[0], c[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs,
                                      kv_dtype=code, max_seq_len=4096)
    us = graph_time(fn) / layers
    kvb = B * ctx * hkv * D * 2 * (1 if kv == "fp8" else 2)
    print(f"{name:26s} B={B} ctx={ctx} {kv} fused={fused}: {us:6.2f} us  {kvb/1e6:6.1f} MB {kvb/us/1e6:5.2f} TB/s", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "norm", "attn"]
    if "gemm" in which:
        bench_gemm("bb qkv", 4096, 2048, L.EPI_BF16)
        bench_gemm("bb o_proj", 2048, 2048, L.EPI_BF16)
        bench_gemm("bb gate_up+silu", 6144, 2048, L.EPI_SILU_MUL)
        bench_gemm("bb down", 2048, 6144, L.EPI_BF16)
        bench_gemm("lm_head", 3072, 2048, L.EPI_F32_BF16RND)
        bench_gemm("cp qkv", 4096, 1024, L.EPI_BF16)
        bench_gemm("cp o_proj", 1024, 2048, L.EPI_BF16)
        bench_gemm("cp gate_up+silu", 3072, 1024, L.EPI_SILU_MUL)
        bench_gemm("cp down", 1024, 3072, L.EPI_BF16)
        bench_gemm("cp qkv hot", 4096, 1024, L.EPI_BF16, hot=True)
        bench_gemm("cp down hot", 1024, 3072, L.EPI_BF16, hot=True)
    if "norm" in which:
        bench_norm(64, 2048); bench_norm(64, 1024)
    if "attn" in which:
        for fused in (False, True):
            bench_attn("backbone", 64, 16, 8, 352, "fp8", 28, fused)
            bench_attn("backbone bf16", 64, 16, 8, 352, "bf16", 14, fused)
            bench_attn("backbone long", 64, 16, 8, 2048, "fp8", 6, fused)
            bench_attn("code predictor", 64, 16, 8, 9, "bf16", 5, fused)
