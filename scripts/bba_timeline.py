"""Stage timeline of the one-launch backbone (csrc/bb_all.hip): wave 0 of every workgroup stamps the 100 MHz counter at 8 points of
each of a layer's five stages (qkv -> attention -> o_proj -> gate_up -> down_proj); debug library.
usage: OMNI_TALKER_DEBUG=1 python scripts/bba_timeline.py [layer]"""
import ctypes as C, os, sys, types
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 13
args = types.SimpleNamespace(allreduce="rccl", model="tts-1.7b", kv="fp8", batch=64, num_blocks=8192, device_weights=True, parallel="tp", sub_batches=1,
                             tp_force=False, prefill_gemm="tile", warmup=0, steps=64, ttfa_steps=0, ctx_extra=0, target_ctx=352)
torch.cuda.set_device(0)
d, w, eng = bench.build_engine(args, 0, 1)
lib = eng.lib
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
bench.setup_requests(d, eng, args)
B = 64
for _ in range(4):
    eng.decode_step(B)
torch.cuda.synchronize()
NST, NW = 8, 256
buf = torch.zeros(8 * NST * NW, dtype=torch.int64, device="cuda")
STAMPS = lib.omni_debug_bb_all_stamps; STAMPS.argtypes = [C.c_void_p, C.c_int]; STAMPS.restype = None
STAMPS(buf.data_ptr(), layer)
eng.decode_step(B)
torch.cuda.synchronize()
STAMPS(None, -1)
t = buf.view(8, NST, NW).cpu().double() * 0.01
order = [(0, "qkv"), (1, "attention"), (2, "o"), (3, "gate_up"), (4, "down")]
seg = ["W issue", "flag wait", "slabs->rstd", "x+MFMA", "barrier", "epilogue", "drain+flag"]
print(f"layer {layer} of the one-launch backbone, us, medians over workgroups (attention columns: batch-0 issue | flag wait | loop | -> records | barrier | merge+stores | drain+flag)")
print(f"{'stage':10s} " + " ".join(f"{s:>11s}" for s in seg) + f" {'total':>8s} {'span':>8s}")
t0 = t[0][0].min()
for s, name in order:
    x = t[s]
    if x.max() == 0:
        continue
    d_ = [(x[k + 1] - x[k]).median().item() for k in range(7)]
    print(f"{name:10s} " + " ".join(f"{v:11.2f}" for v in d_) + f" {sum(d_):8.2f} {(x[7].max() - x[0].min()).item():8.2f}")
lab = ["entry", "W issued", "flags seen", "rstd", "MFMA done", "combine bar", "stores issued", "flag out"]
print("absolute times since the first workgroup entered the layer's qkv stage (us), median over workgroups [min..max]:")
print(f"{'stage':10s} " + " ".join(f"{s:>19s}" for s in lab))
for s, name in order:
    if t[s].max() == 0:
        continue
    print(f"{name:10s} " + " ".join(f"{(t[s][k] - t0).median().item():7.2f}[{(t[s][k] - t0).min().item():5.1f}..{(t[s][k] - t0).max().item():5.1f}]" for k in range(8)))
