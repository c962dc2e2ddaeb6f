import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L, ops
from scripts.bench_ops import graph_time, BF16, dev
lib = L.load(); #lib.omni_debug_ablate.argtypes = [C.c_int]; lib.omni_debug_ablate.restype = None
def run(N, K, epi, R=8):
    rows = 2 * N if epi == L.EPI_SILU_MUL else N
    Ws = [torch.randn(rows, K, device=dev, dtype=BF16) * 0.02 for _ in range(R)]
    x = torch.randn(64, K, device=dev, dtype=BF16)
    def fn():
        for i in range(R): ops.gemm(x, Ws[i], epilogue=epi)
    for bits, name in ((0, "full"), (1, "no x loads"), (2, "no W loads"), (3, "no loads"), (4, "no mfma"), (8, "no combine"), (9, "no x, no combine"), (15, "nothing")):
        #lib.omni_debug_ablate(bits)
        print(f"N={N} K={K} epi={epi} {name:18s}: {graph_time(fn)/R:6.2f} us", flush=True)
    #lib.omni_debug_ablate(0)
run(4096, 2048, L.EPI_BF16)
run(6144, 2048, L.EPI_SILU_MUL)
run(2048, 6144, L.EPI_BF16)
