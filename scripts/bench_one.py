"""One GEMM shape at M=64 and M=1 for PMC passes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_ops import *
bench_gemm("bb qkv", 4096, 2048, L.EPI_BF16, M=64)
bench_gemm("bb qkv", 4096, 2048, L.EPI_BF16, M=1)
