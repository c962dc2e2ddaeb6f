import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_ops import *
for M in (64, 32, 16, 1):
    bench_gemm("bb qkv", 4096, 2048, L.EPI_BF16, M=M)
    bench_gemm("bb gate_up", 6144, 2048, L.EPI_SILU_MUL, M=M)
