"""Launch / dependency floors under hipGraph replay (diagnostic)."""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
f = lib.omni_debug_launch
f.restype = C.c_int; f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = "cuda"
buf = torch.zeros(1 << 22, device=dev)
idx = torch.randint(0, 1 << 20, (1 << 22,), device=dev, dtype=torch.int32)
out = torch.zeros(1 << 22, device=dev, dtype=torch.int32)

def timeit(name, mode, blocks, threads, arg=0, reps=500):
    def go():
        L.check(f(mode, blocks, threads, buf.data_ptr() if mode == 1 else idx.data_ptr(), out.data_ptr(), arg, reps, L.current_stream()))
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(); torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / reps * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        go()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / reps / 5 * 1e6
    print(f"{name:40s} eager {eager:6.2f} us/kernel   graph {gr:6.2f} us/kernel")

timeit("empty 256x512", 0, 256, 512)
timeit("lds 0KB 256x512", 3, 256, 512, 0)
timeit("lds 32KB 256x512", 3, 256, 512, 32768)
timeit("lds 64KB 256x512", 3, 256, 512, 65536)
timeit("lds 64KB 384x512", 3, 384, 512, 65536)
timeit("lds 64KB 512x512", 3, 512, 512, 65536)
timeit("vgpr32 256x512 n=1", 4, 256, 512, 1)
timeit("vgpr120 256x512 n=1", 5, 256, 512, 1)
timeit("vgpr120 384x512 n=1", 5, 384, 512, 1)
timeit("vgpr120 256x512 n=20", 5, 256, 512, 20)
