"""(1) XCC id of every block over consecutive launches (is block b mod 8 -> XCD stable across launches of a stream / graph?)
(2) a dependent chain of small kernels: by-value struct kernarg vs preloaded scalar kernargs."""
import ctypes as C, os, sys
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
vp, ci = C.c_void_p, C.c_int
lib.omni_debug_xcc_probe.argtypes = [vp, vp, ci, ci, ci, ci, vp]; lib.omni_debug_xcc_probe.restype = ci
lib.omni_debug_chain.argtypes = [ci, vp, vp, ci, ci, vp]; lib.omni_debug_chain.restype = ci
st = lambda: torch.cuda.current_stream().cuda_stream
scratch = torch.zeros(1 << 20, device="cuda")
for gx, gy, odd in ((64, 4, 0), (64, 4, 250), (128, 2, 0), (256, 1, 16), (64, 4, 3)):
    reps = 6
    out = torch.full((reps, gx * gy), -1, dtype=torch.int32, device="cuda")
    for mode in ("eager", "graph"):
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                L.check(lib.omni_debug_xcc_probe(out.data_ptr(), scratch.data_ptr(), gx, gy, odd, reps, st()))
            g.replay()
        else:
            L.check(lib.omni_debug_xcc_probe(out.data_ptr(), scratch.data_ptr(), gx, gy, odd, reps, st()))
        torch.cuda.synchronize()
        o = out.cpu()
        rr = [(o[r] - torch.arange(gx * gy)) % 8 for r in range(reps)]
        consistent = [bool((x == x[0]).all()) for x in rr]          # round-robin inside one launch?
        offs = [int(x[0]) for x in rr]
        print(f"grid ({gx},{gy}) odd={odd} {mode}: round-robin within launch {consistent}; offset (xcc - block) mod 8 per launch {offs}")
a = torch.zeros(256 * 512, device="cuda"); b = torch.zeros_like(a)
for blocks in (256, 64):
    for mode in (0, 1):
        reps = 400
        g = torch.cuda.CUDAGraph()
        L.check(lib.omni_debug_chain(mode, a.data_ptr(), b.data_ptr(), blocks, 4, st())); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            L.check(lib.omni_debug_chain(mode, a.data_ptr(), b.data_ptr(), blocks, reps, st()))
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"chain blocks={blocks} {'scalar+preload' if mode else 'struct kernarg'}: {e0.elapsed_time(e1) / 5 / reps * 1e3:.3f} us per kernel")
