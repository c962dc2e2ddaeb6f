"""Launch / dependency floors under hipGraph replay (diagnostic)."""
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

timeit("empty 1x64", 0, 1, 64)
timeit("empty 256x256", 0, 256, 256)
timeit("empty 1024x512", 0, 1024, 512)
timeit("touch 1x64", 1, 1, 64)
timeit("touch 256x256", 1, 256, 256)
timeit("chase depth1 16x256", 2, 16, 256, 1)
timeit("chase depth2 16x256", 2, 16, 256, 2)
timeit("chase depth4 16x256", 2, 16, 256, 4)
timeit("chase depth8 16x256", 2, 16, 256, 8)
