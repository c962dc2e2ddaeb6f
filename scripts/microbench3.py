"""Do parallel hipGraph branches (two capture streams) overlap on this stack? (diagnostic)"""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
f = lib.omni_debug_launch; f.restype = C.c_int
f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
idx = torch.randint(0, 1 << 20, (1 << 22,), device="cuda", dtype=torch.int32)
out = torch.zeros(1 << 22, device="cuda", dtype=torch.int32)
out2 = torch.zeros(1 << 22, device="cuda", dtype=torch.int32)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def chain(stream, o, reps, blocks, depth):
    L.check(f(2, blocks, 256, idx.data_ptr(), o.data_ptr(), depth, reps, stream.cuda_stream))

def timed(two, graph, reps=200, blocks=16, depth=8):
    def go():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        s1.wait_event(ev); chain(s1, out, reps, blocks, depth)
        e1 = torch.cuda.Event(); e1.record(s1); cur.wait_event(e1)
        if two:
            s2.wait_event(ev); chain(s2, out2, reps, blocks, depth)
            e2 = torch.cuda.Event(); e2.record(s2); cur.wait_event(e2)
    go(); torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): go()
        run = g.replay
    else:
        run = go
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e6
for blocks in (16, 256):
    for graph in (False, True):
        a = timed(False, graph, blocks=blocks); b = timed(True, graph, blocks=blocks)
        print(f"blocks={blocks:4d} graph={graph}: one chain {a:8.1f} us, two chains {b:8.1f} us  (x{b/a:.2f})")
