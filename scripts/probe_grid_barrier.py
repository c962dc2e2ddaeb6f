"""Kernel boundary vs in-kernel grid barrier for a chain of dependent steps (diagnostic).
A hipGraph of N trivial dependent launches (omni_debug_chain) against ONE persistent launch doing N x { cross-workgroup hand-off +
grid barrier } (omni_debug_grid_barrier_chain), 256 workgroups x 512 threads each."""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
ch = lib.omni_debug_chain; ch.restype = C.c_int; ch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
gb = lib.omni_debug_grid_barrier_chain; gb.restype = C.c_int
gb.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
blocks, iters = 256, 400
a = torch.zeros(blocks * 512, device="cuda"); b = torch.zeros_like(a)
ctr = torch.zeros(256, dtype=torch.int32, device="cuda"); err = torch.zeros(1, dtype=torch.int32, device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / iters


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    L.check(ch(0, a.data_ptr(), b.data_ptr(), blocks, iters, st.cuda_stream)); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        L.check(ch(0, a.data_ptr(), b.data_ptr(), blocks, iters, st.cuda_stream))
    print(f"kernel boundary (hipGraph of {iters} dependent launches): {timed(g.replay):.2f} us per step")
    for mode, name in ((0, "fences, one counter"), (1, "fences, counter + s_sleep"), (2, "fences, per-XCD counters"), (3, "sc1 data, one counter"), (4, "sc1 data, per-XCD counters")):
        for nb in (256, 64):
            a.zero_(); b.zero_()
            fn = lambda: L.check(gb(mode, a.data_ptr(), b.data_ptr(), ctr.data_ptr(), err.data_ptr(), nb, iters, st.cuda_stream))
            t = timed(fn)
            # value check: after `iters` steps every element went through x -> x / 2 + 1 iters times from 0
            ref = 0.0
            for _ in range(iters): ref = ref * 0.5 + 1.0
            res = (b if iters & 1 else a)[: nb * 512]
            ok = bool(((res - ref).abs() < 1e-5).all())
            print(f"grid barrier, {nb:3d} workgroups, {name:28s}: {t:.2f} us per step   err={int(err.item())} values_ok={ok}")
