"""Where does the ~4.5 us per-kernel floor of the captured step come from? (diagnostic)"""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
f = lib.omni_debug_mix; f.restype = C.c_int; f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
small = torch.zeros(1 << 20, device="cuda")
big = torch.zeros(1 << 28, dtype=torch.uint8, device="cuda")   # 256 MiB
def run(pattern, nbytes, reps=50):
    def go(): L.check(f(pattern, small.data_ptr(), big.data_ptr(), nbytes, reps, L.current_stream()))
    go(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): go()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps * 1e3
same = run(0, 0); distinct = run(1, 0)
print(f"4 x same trivial kernel      : {same/4:5.2f} us/kernel")
print(f"4 x distinct trivial kernels : {distinct/4:5.2f} us/kernel")
for mb in (16, 64, 256):
    n = mb << 20
    s_only = run(6, n)
    both = run(3, n)
    print(f"stream {mb:3d} MiB alone {s_only:7.2f} us ({n/s_only/1e6:5.2f} TB/s); + 4 distinct trivial kernels: {both:7.2f} us -> {(both - s_only)/4:5.2f} us per small kernel")
f2 = lib.omni_debug_cfgmix; f2.restype = C.c_int; f2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
def run2(mode, reps=50):
    def go(): L.check(f2(mode, small.data_ptr(), reps, L.current_stream()))
    go(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): go()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps / 4 * 1e3
print(f"4 x B(256thr,no LDS)               : {run2(0):5.2f} us/kernel")
print(f"4 x A(512thr,64KB LDS,256 WGs)     : {run2(1):5.2f} us/kernel")
print(f"A64K, B, A32K, C(1 wave) mixed     : {run2(2):5.2f} us/kernel")
print(f"A with 4 different LDS/grid configs: {run2(3):5.2f} us/kernel")
