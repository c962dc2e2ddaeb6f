"""Same-process A/B of run-time policy knobs (libomni_talker_debug.so) on the W3 decode step: one engine, one request set,
one captured hipGraph per knob setting, timed round-robin (boxes differ by ~1 %, graphs in one process do not).
usage: OMNI_TALKER_DEBUG=1 python scripts/ab_knobs.py [--steps 48] [--rounds 3] knob=val[,knob=val] ...   e.g.  cp_prefetch=0 cp_prefetch=1"""
import argparse, ctypes as C, os, sys, types
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--model", default="tts-1.7b")
ap.add_argument("--kv", default="fp8")
ap.add_argument("--ctx", type=int, default=352)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("settings", nargs="+")
a = ap.parse_args()
args = types.SimpleNamespace(allreduce="rccl", model=a.model, kv=a.kv, batch=a.batch, num_blocks=8192, device_weights=True, parallel="tp", sub_batches=1,
                             tp_force=False, prefill_gemm="tile", warmup=0, steps=a.steps * a.rounds * len(a.settings) + 64, ttfa_steps=0, ctx_extra=0, target_ctx=a.ctx)
torch.cuda.set_device(0)
d, w, eng = bench.build_engine(args, 0, 1)
lib = eng.lib
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
lens, _ = bench.setup_requests(d, eng, args)
B = a.batch

def apply(setting):
    for kv in setting.split(","):
        k, v = kv.split("=")
        fn = getattr(lib, "omni_debug_" + k)
        vals = [int(x) for x in v.split(":")]
        fn.argtypes = [C.c_int] * len(vals); fn.restype = None
        fn(*vals)

eng.decode_step(B); torch.cuda.synchronize()
adv = max(0, a.ctx - int(eng.seq_lens[:B].float().mean().item()) - (a.steps + 1) * a.rounds * len(a.settings) // 2)
for _ in range(adv):
    eng.decode_step(B)
graphs, bb_graphs = {}, {}
for s in a.settings:
    apply(s)
    eng.decode_step(B); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.decode_step(B)
    graphs[s] = g
    eng.backbone_step(B); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        eng.backbone_step(B)
    bb_graphs[s] = g2
res = {s: [] for s in a.settings}
bb = {}
for s in a.settings:
    g2 = bb_graphs[s]
    g2.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(32):
        g2.replay()
    e1.record(); torch.cuda.synchronize()
    bb[s] = e0.elapsed_time(e1) / 32
for r in range(a.rounds):
    for s in a.settings:
        g = graphs[s]
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res[s].append(e0.elapsed_time(e1) / a.steps)
print("mean ctx at end", float(eng.seq_lens[:B].float().mean().item()))
for s in a.settings:
    print(f"{s:40s} ms/step " + " ".join(f"{x:.4f}" for x in res[s]) + f"   min {min(res[s]):.4f}   backbone-only {bb[s]:.4f}  cp = {min(res[s]) - bb[s]:.4f}")
