"""Stage timeline of the backbone segment launches (csrc/bb_chain.hip): o_proj -> gate_up -> down_proj -> next qkv, wave 0 of every
workgroup stamps the 100 MHz counter at 8 points per stage (debug library).  Prints the LAST layer-launch of a W3 decode step.
usage: OMNI_TALKER_DEBUG=1 python scripts/bb_timeline.py"""
import ctypes as C, os, sys, types
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = types.SimpleNamespace(allreduce="rccl", model="tts-1.7b", kv="fp8", batch=64, num_blocks=8192, device_weights=True, parallel="tp", sub_batches=1,
                             tp_force=False, prefill_gemm="tile", warmup=0, steps=64, ttfa_steps=0, ctx_extra=0, target_ctx=352)
torch.cuda.set_device(0)
d, w, eng = bench.build_engine(args, 0, 1)
lib = eng.lib
_mode = os.environ.get("BB_STAMPS", "eng")
_f = lib.omni_debug_bb_engine; _f.argtypes = [C.c_int]; _f.restype = None      # BB_STAMPS=bb: the plain chain, eng: the loader / consumer engine
_f(int(_mode == "eng"))
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
bench.setup_requests(d, eng, args)
B = 64
for _ in range(4):
    eng.decode_step(B)
torch.cuda.synchronize()
NST, NW = 8, 256
buf = torch.zeros(40 * NST * NW, dtype=torch.int64, device="cuda")
STAMPS = getattr(lib, "omni_debug_" + os.environ.get("BB_STAMPS", "eng") + "_stamps"); STAMPS.argtypes = [C.c_void_p]; STAMPS.restype = None
STAMPS(buf.data_ptr())
eng.decode_step(B)
torch.cuda.synchronize()
STAMPS(None)
t = buf.view(40, NST, NW)[:4].cpu().double() * 0.01
names = ["o", "gate_up", "down", "qkv(next)"]
seg = ["W issue", "flag wait", "slabs->rstd", "x+MFMA", "barrier", "epilogue", "drain+flag"]
print("layer 26's segment launch (the last one with a qkv stage is layer 26; this is the step's final launch: 3 stages), us, medians over workgroups")
print(f"{'stage':10s} " + " ".join(f"{s:>11s}" for s in seg) + f" {'total':>8s} {'span':>8s}")
for s in range(3):
    x = t[s]
    d_ = [(x[k + 1] - x[k]).median().item() for k in range(7)]
    print(f"{names[s]:10s} " + " ".join(f"{v:11.2f}" for v in d_) + f" {sum(d_):8.2f} {(x[7].max() - x[0].min()).item():8.2f}")
print(f"launch span {(t[2][7].max() - t[0][0].min()).item():.1f} us")

if os.environ.get("BB_STAMPS", "eng") == "pp":
    t0 = t[0][0].min()
    print("absolute times since the first workgroup entered (us), median over workgroups [max]:")
    lab = ["entry", "W issued", "flags seen", "rstd", "MFMA done", "combine bar", "stores issued", "flag out"]
    print(f"{'stage':10s} " + " ".join(f"{s:>15s}" for s in lab))
    for s in range(3):
        print(f"{names[s]:10s} " + " ".join(f"{(t[s][k] - t0).median().item():8.2f}[{(t[s][k] - t0).max().item():5.1f}]" for k in range(8)))
if os.environ.get("BB_STAMPS", "eng") == "eng":
    L_ = buf.view(40, NST, NW)[32].cpu().double()
    ts = L_[:5] * 0.01
    print("loader wave 8 (medians, us): issue o %.2f | gate_up %.2f | down %.2f | (qkv %.2f); polls spent waiting for FIFO space: median %d, max %d" % (
        (ts[1] - ts[0]).median(), (ts[2] - ts[1]).median(), (ts[3] - ts[2]).median(), (ts[4] - ts[3]).median(), int(L_[5].median()), int(L_[5].max())))
    print("loader start -> compute-wave o-stage entry (us): %.2f; loader end vs down-stage end: %.2f" % (
        (t[0][0] - ts[0]).median(), (t[2][7] - ts[3]).median()))
