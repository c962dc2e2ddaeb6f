"""Round 6: the code predictor's passes INSIDE the real decode step (W3: 28-layer backbone in front, sampling on, one hipGraph) against the
same passes with the predictor phase replayed alone (its 161 MB of weights warm in the Infinity Cache): wave 0 of every workgroup stamps
the 100 MHz counter at 8 points of every stage of every pass (debug library, one stamp block per pass).  Prints per pass its span on the
chip in both settings, and the per-stage medians of passes 2, 3, 8, 15 and of the pair kernel.
usage: python scripts/step_timeline.py [--batch 64] [--ctx 352]"""
import argparse, ctypes as C, os, sys, types
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--ctx", type=int, default=352)
ap.add_argument("--kv", default="fp8")
ap.add_argument("--model", default="tts-1.7b")
ap.add_argument("--knobs", default="", help="debug knobs applied before capture, e.g. chain_defer=1,bb_deep=6")
ap.add_argument("--stages", action="store_true", help="per-stage medians of passes 2, 3, 8, 15")
a = ap.parse_args()
args = types.SimpleNamespace(allreduce="rccl", model=a.model, kv=a.kv, batch=a.batch, num_blocks=8192, device_weights=True, parallel="tp", sub_batches=1,
                             tp_force=False, prefill_gemm="tile", warmup=0, steps=512, ttfa_steps=0, ctx_extra=0, target_ctx=a.ctx)
torch.cuda.set_device(0)
d, w, eng = bench.build_engine(args, 0, 1)
lib = eng.lib
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
lens, _ = bench.setup_requests(d, eng, args)
B = a.batch
for kv in [k for k in a.knobs.split(",") if k]:
    k, v = kv.split("=")
    fn = getattr(lib, "omni_debug_" + k)
    vals = [int(x) for x in v.split(":")]
    fn.argtypes = [C.c_int] * len(vals); fn.restype = None
    fn(*vals)
eng.decode_step(B); torch.cuda.synchronize()
adv = max(0, a.ctx - int(eng.seq_lens[:B].float().mean().item()) - 40)
for _ in range(adv):
    eng.decode_step(B)
torch.cuda.synchronize()
NST, NW, NP = 8, 256, 16
buf = torch.zeros(NP * 40 * NST * NW, dtype=torch.int64, device="cuda")
lib.omni_debug_chain_stamps.argtypes = [C.c_void_p]; lib.omni_debug_chain_stamps.restype = None
lib.omni_debug_chain_stamps(buf.data_ptr())          # the pointer is a kernel argument: captured with the graphs below

def capture(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g

g_step = capture(lambda: eng.decode_step(B))
g_cp = capture(lambda: eng.step_part(B, 1))
lib.omni_debug_chain_stamps(None)
SLOT = list(range(25)) + [30, 26]
names = ["qkv", "attn", "o", "gate_up", "down"]
def nm(s):
    return f"L{s // 5} {names[s % 5]:8s}" if s < 25 else ("head       " if s == 25 else "sampler    ")

def run(g, reps=8):
    out = []
    for _ in range(reps):
        buf.zero_()
        g.replay(); g.replay()
        torch.cuda.synchronize()
        raw = buf.view(NP, 40, NST, NW).cpu()
        out.append(raw[:, SLOT].double() * 0.01)      # us
        run.smp = raw[:, 27]                          # the pick's own stamps (sampler_body.cuh SMP_STAMP): [pass][8][wg]
        run.smp_in = raw[:, 26]
    return out

def spans(t):
    """per pass: first stage entry (any workgroup) -> last flag of the sampler; pair kernel = block 0"""
    res = {}
    for p in [0] + list(range(2, 16)):
        x = t[p]
        ent = x[0, 0][x[0, 0] > 0]
        end = x[26, 7][x[26, 7] > 0]
        if len(ent) and len(end):
            res[p] = (end.max() - ent.min()).item()
    return res

def timed(g, n=64):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

print(f"B={B} kv={a.kv} mean ctx {float(eng.seq_lens[:B].float().mean()):.1f} knobs '{a.knobs}'")
print(f"graph times (stamps on): step {timed(g_step):.4f} ms, predictor phase alone {timed(g_cp):.4f} ms")
ts, tc = run(g_step), run(g_cp)
import statistics as S
print("pass   in-step span us (median of 8 replays)   alone (warm) span us    diff")
tot_s = tot_c = 0.0
for p in [0] + list(range(2, 16)):
    vs = S.median([spans(t).get(p, float('nan')) for t in ts]); vc = S.median([spans(t).get(p, float('nan')) for t in tc])
    tot_s += vs; tot_c += vc
    print(f"{'pair' if p == 0 else p:>4}   {vs:10.2f}                                {vc:10.2f}          {vs - vc:+6.2f}")
print(f" sum   {tot_s:10.2f}                                {tot_c:10.2f}          {tot_s - tot_c:+6.2f}")
if a.stages:
    seg = ["W issue", "flag wait", "slabs->rstd", "x+MFMA", "barrier", "epilogue", "drain+flag"]
    for label, tt in (("in-step", ts[-1]), ("alone", tc[-1])):
        for p in (0, 2, 3, 8, 15):
            t = tt[p]
            print(f"--- {label}, pass {'pair' if p == 0 else p}: medians over the workgroups that stamped, us")
            print(f"{'stage':11s} " + " ".join(f"{s:>11s}" for s in seg) + f" {'total':>8s} {'span':>8s}")
            for s in range(27):
                m = (t[s][7] > 0) & (t[s][0] > 0)
                x = t[s][:, m]
                if x.shape[1] == 0:
                    continue
                if s == 26:
                    own = x[4] > 0
                    x = x[:, own]
                    d_ = [(x[1] - x[0]).median().item(), (x[2] - x[1]).median().item(), 0.0, (x[4] - x[2]).median().item(), 0.0,
                          (x[6] - x[4]).median().item(), (x[7] - x[6]).median().item()]
                elif s % 5 == 1 and s < 25:        # attention stage: stamps 0, 1, 2, 6, 7 only
                    d_ = [(x[1] - x[0]).median().item(), (x[2] - x[1]).median().item(), 0.0, (x[6] - x[2]).median().item(), 0.0, 0.0,
                          (x[7] - x[6]).median().item()]
                else:
                    d_ = [(x[k + 1] - x[k]).median().item() for k in range(7)]
                span = (x[7].max() - x[0].min()).item()
                print(nm(s) + " " + " ".join(f"{v:11.2f}" for v in d_) + f" {sum(d_):8.2f} {span:8.2f}")

    # the sampler stage's pick, from inside (slot 27): flags seen -> pick entered (logit loads issued) -> A bound (loads arrived, local maxima,
    # per-wave quota rank) -> B compaction -> C rank / kth -> D Gumbel scores -> block argmax; candidates ranked
    run(g_step, 1)
    sm, si = run.smp.double(), run.smp_in.double()
    print("--- sampler pick inside (in-step), medians over the 64 sampling workgroups, us: load issue | A bound | B compact | C rank+kth | D gumbel | argmax | candidates n")
    for p in (0, 2, 3, 8, 15):
        own = sm[p][5] > 0
        if own.sum() == 0:
            continue
        x = sm[p][:, own] * 0.01
        f = si[p][2][own] * 0.01
        seg = [(x[0] - f).median().item()] + [(x[k + 1] - x[k]).median().item() for k in range(5)]
        print(f"pass {'pair' if p == 0 else p:>4}: " + " ".join(f"{v:6.2f}" for v in seg) + f"   n median {sm[p][6][own].median().item():.0f} max {sm[p][6][own].max().item():.0f}")
