"""Tile sweep of the skinny GEMM (plain and PRO_XNORM) and of the residual epilogue, cold rotating weights (diagnostic)."""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from ht_vllm_omni_amd.engine import frag_shuffle
lib = L.load()
lib.omni_debug_tile.argtypes = [C.c_int, C.c_int]; lib.omni_debug_tile.restype = None
dev, BF16 = "cuda", torch.bfloat16

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
    return e0.elapsed_time(e1) / reps * 1e3

def sweep(name, N, K, epi, M=64, modes=("plain", "xnorm")):
    rows = 2 * N if epi in L.SILU_EPIS else N
    wbytes = rows * K * 2
    R = max(2, min(48, int(1.0e9 // wbytes)))
    D = int(os.environ.get('BT_D', 0)) or R      # distinct weight copies (1 = L2-hot)
    Ws = [frag_shuffle(torch.randn(rows, K, device=dev, dtype=BF16) * 0.02) for _ in range(D)]
    Ws = [Ws[i % D] for i in range(R)]
    Mp = (M + 15) // 16 * 16
    x = frag_shuffle(torch.randn(Mp, K, device=dev, dtype=BF16))
    nw = torch.ones(K, device=dev, dtype=BF16)
    part = torch.rand(K // 16, 64, device=dev) + 1.0
    lay = L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG
    out = []
    for mode in modes:
        for nt in (1, 2, 3, 4):
            if epi in L.SILU_EPIS and nt == 1: continue
            if nt == 3 and epi != L.EPI_SILU_MUL_GU8: continue
            if os.environ.get('BT_POLICY') and nt != 1 and not (nt == 2 and epi in L.SILU_EPIS): continue
            for mt in (1, 2, 4):
                if os.environ.get('BT_POLICY'):
                    if mt != 1: continue
                    nt_, mt_ = 0, 0
                else:
                    nt_, mt_ = nt, mt
                lib.omni_debug_tile(nt_, mt_)
                try:
                    if mode == "plain":
                        def fn():
                            for i in range(R): ops.gemm(x, Ws[i], epilogue=epi, layout=lay, M=M)
                    else:
                        def fn():
                            for i in range(R): ops.gemm_xnorm(x, part, K // 16, nw, Ws[i], 1e-6, M=M, epilogue=epi)
                    us = graph_time(fn) / R
                    out.append(f"{mode[0]}{nt}x{mt}:{us:5.1f}")
                except L.OmniError as e:
                    out.append(f"{mode[0]}{nt}x{mt}: n/a")
    lib.omni_debug_tile(0, 0)
    print(f"{name:14s} N={N:5d} K={K:5d} {wbytes/1e6:5.1f}MB | " + " ".join(out), flush=True)

def sweep_resid(name, N, K, M=64):
    wbytes = N * K * 2
    R = max(2, min(48, int(1.0e9 // wbytes)))
    D = int(os.environ.get('BT_D', 0)) or R
    Ws = [frag_shuffle(torch.randn(N, K, device=dev, dtype=BF16) * 0.02) for _ in range(D)]
    Ws = [Ws[i % D] for i in range(R)]
    x = frag_shuffle(torch.randn(64, K, device=dev, dtype=BF16))
    r = frag_shuffle(torch.randn(64, N, device=dev, dtype=BF16))
    part = torch.zeros(N // 16, 64, device=dev)
    out = []
    for mt in (1, 2, 4):
        if os.environ.get('BT_POLICY'):
            if mt != 1: continue
            lib.omni_debug_tile(0, 0)
        else:
            lib.omni_debug_tile(1, mt)
        def fn():
            for i in range(R): ops.gemm_resid(x, Ws[i], r, part, M=M)
        out.append(f"resid 1x{mt}:{graph_time(fn)/R:5.1f}")
        def fn2():
            for i in range(R): ops.gemm(x, Ws[i], layout=L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG, M=M)
        out.append(f"plain 1x{mt}:{graph_time(fn2)/R:5.1f}")
    lib.omni_debug_tile(0, 0)
    print(f"{name:14s} N={N:5d} K={K:5d} {wbytes/1e6:5.1f}MB | " + " ".join(out), flush=True)

if __name__ == "__main__":
    if os.environ.get("BT_ONLY") == "xnorm":
        sweep("cp qkv", 4096, 1024, L.EPI_BF16, modes=("xnorm",)); sweep("bb qkv", 4096, 2048, L.EPI_BF16, modes=("xnorm",))
        sweep("cp head", 2048, 1024, L.EPI_F32_BF16RND, modes=("xnorm",)); sweep("lm_head", 3072, 2048, L.EPI_F32_BF16RND, modes=("xnorm",))
        sweep("cp gate_up gu8", 3072, 1024, L.EPI_SILU_MUL_GU8, modes=("xnorm",)); sweep("bb gate_up gu8", 6144, 2048, L.EPI_SILU_MUL_GU8, modes=("xnorm",))
        sys.exit(0)
    if os.environ.get("BT_ONLY") == "resid":
        sweep_resid("cp o", 1024, 2048); sweep_resid("cp down", 1024, 3072)
        sweep_resid("bb o", 2048, 2048); sweep_resid("bb down", 2048, 6144)
        sys.exit(0)
    sweep("cp qkv", 4096, 1024, L.EPI_BF16)
    sweep("cp gate_up", 3072, 1024, L.EPI_SILU_MUL)
    sweep("cp gate_up gu8", 3072, 1024, L.EPI_SILU_MUL_GU8)
    sweep("cp head", 2048, 1024, L.EPI_F32_BF16RND, modes=("xnorm",))
    sweep("bb qkv", 4096, 2048, L.EPI_BF16)
    sweep("bb gate_up", 6144, 2048, L.EPI_SILU_MUL)
    sweep("bb gate_up gu8", 6144, 2048, L.EPI_SILU_MUL_GU8)
    sweep("lm_head", 3072, 2048, L.EPI_F32_BF16RND, modes=("xnorm",))
    sweep_resid("cp o", 1024, 2048)
    sweep_resid("cp down", 1024, 3072)
    sweep_resid("bb o", 2048, 2048)
    sweep_resid("bb down", 2048, 6144)
    R = 32
    for rows, hidden in ((64, 2048), (64, 1024)):
        xs = [torch.randn(rows, hidden, device=dev, dtype=BF16) for _ in range(R)]
        ds = [torch.randn(rows, hidden, device=dev, dtype=BF16) for _ in range(R)]
        w = torch.ones(hidden, device=dev, dtype=BF16)
        def fn():
            for i in range(R): ops.rmsnorm(None, w, 1e-6, delta=ds[i], residual=xs[i])
        print(f"rmsnorm+resid {rows}x{hidden}: {graph_time(fn)/R:6.2f} us", flush=True)
