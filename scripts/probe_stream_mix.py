"""What bounds the streaming phases of the backbone chain: per-CU rates of a unique (HBM) stream W, a shared (L2-hit) stream X, and
their mixes, as a function of the loads in flight per wave (csrc/debug.hip dbg_stream_mix_kernel; debug library).
usage: OMNI_TALKER_DEBUG=1 python scripts/probe_stream_mix.py"""
import ctypes as C, os, sys
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
lib = L.load()
f = lib.omni_debug_stream_mix
f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
f.restype = C.c_int
WG = 256
w_wg = 4 << 20                                   # 4 MB of unique bytes per workgroup and launch: 1 GB in all
W = torch.randint(0, 255, (WG * w_wg,), dtype=torch.uint8, device="cuda")
X = torch.randint(0, 255, (512 << 10,), dtype=torch.uint8, device="cuda")
out = torch.zeros(WG, dtype=torch.int32, device="cuda")
st = L.current_stream()

def run(mode, nw, nx, depth, sc1=1, x_bytes=256 << 10, reps=5):
    f(W.data_ptr(), w_wg, X.data_ptr(), x_bytes, nw, nx, mode, sc1, depth, out.data_ptr(), 1, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f(W.data_ptr(), w_wg, X.data_ptr(), x_bytes, nw, nx, mode, sc1, depth, out.data_ptr(), reps, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    waves_w = {0: 8, 1: 0, 2: 8, 3: 4}[mode]; waves_x = {0: 0, 1: 8, 2: 8, 3: 4}[mode]
    wb, xb = waves_w * nw * 1024, waves_x * nx * 1024            # bytes per workgroup
    return us, wb / us / 1e3, xb / us / 1e3                      # us, GB/s per CU of W, of X

print("mode: 0 W only (unique, HBM) | 1 X only (256 KB shared, L2) | 2 W and X alternating in every wave | 3 waves 0-3 W, waves 4-7 X")
print(f"{'mode':>4} {'depth':>5} {'sc1':>3} {'KB/wave W':>9} {'KB/wave X':>9} {'us':>8} {'W GB/s/CU':>10} {'X GB/s/CU':>10} {'W TB/s chip':>11} {'W+X TB/s':>9}")
for depth in (4, 16):
    for mode, nw, nx, sc1 in ((0, 512, 0, 1), (1, 0, 2048, 1), (1, 0, 2048, 0), (1, 0, 2048, 3), (2, 512, 512, 1), (2, 512, 1024, 1), (2, 512, 1024, 3), (3, 1024, 1024, 1), (3, 1024, 2048, 1),
                              (3, 1024, 2048, 3)):      # sc1 column: 1 = sc1 loads, 0 = sc0 loads, 3 = sc1 + every workgroup starts its walk over X elsewhere
        us, wg_, xg_ = run(mode, nw, nx, depth, sc1)
        print(f"{mode:4d} {depth:5d} {sc1:3d} {nw:9d} {nx:9d} {us:8.1f} {wg_:10.1f} {xg_:10.1f} {wg_ * WG / 1e3:11.2f} {(wg_ + xg_) * WG / 1e3:9.2f}")
