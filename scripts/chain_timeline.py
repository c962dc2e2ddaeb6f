"""Where a stage of the persistent code-predictor chain spends its time: wave 0 of every workgroup stamps the 100 MHz
counter at 8 points of each stage (csrc/cp_chain.hip CH_STAMP, debug library only).  Prints, per stage of the LAST pass of a
code-predictor run, the median over workgroups of each segment and the stage's span on the chip (first entry -> last flag).
usage: OMNI_TALKER_DEBUG=1 python scripts/chain_timeline.py [--batch 64]"""
import argparse, ctypes as C, os, sys
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
d = get_dims("tts-1.7b").with_(layers=1, max_model_len=256)
w = make_weights(d, seed=1, std=0.02)
eng = TalkerEngine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=64)
lib = eng.lib
NS, NST, NW = 27, 8, 256
SLOT = list(range(25)) + [30, 26]            # stamp slots: 25 layer stages, head GEMM, sampler
buf = torch.zeros(16 * 40 * NST * NW, dtype=torch.int64, device="cuda")      # one block per pass (round 6); the last pass is block 15
lib.omni_debug_chain_stamps.argtypes = [C.c_void_p]; lib.omni_debug_chain_stamps.restype = None
B = a.batch
g = torch.Generator().manual_seed(0)
code0 = torch.randint(1, d.codebook, (B,), generator=g).to(torch.int32).cuda()
e0 = w["embed"][code0.cpu().long()].cuda()
lh = torch.randn(B, d.hidden, generator=g).to(torch.bfloat16).cuda()
for _ in range(3):
    eng.code_predictor(code0, e0, lh, greedy=True)
torch.cuda.synchronize()
lib.omni_debug_chain_stamps(buf.data_ptr())
acc = None
for _ in range(a.reps):
    eng.code_predictor(code0, e0, lh, greedy=True)
    torch.cuda.synchronize()
    t = buf.view(16, 40, NST, NW)[15][SLOT].cpu().double() * 0.01                 # us
    acc = t if acc is None else acc
    # keep the last repetition (every launch overwrites); medians over workgroups are stable
lib.omni_debug_chain_stamps(None)
t = acc
names = ["qkv", "attn", "o", "gate_up", "down"]
def nm(s):
    return f"L{s // 5} {names[s % 5]:9s}" if s < 25 else ("head        " if s == 25 else "sampler     ")
seg = ["W issue", "flag wait", "slabs->rstd", "x+MFMA", "barrier", "epilogue", "drain+flag"]
print(f"B={B}; medians over 256 workgroups, us.  span = first workgroup entering -> last flag published; gap = this stage's median flag time -> next stage's median 'flags seen'")
print(f"{'stage':12s} " + " ".join(f"{s:>11s}" for s in seg) + f" {'total':>8s} {'span':>8s} {'gap':>6s}")
tot = 0.0
grp = os.environ.get("GROUP")        # restrict the statistics to one 64-workgroup row group (partly filled batches)
def live(x):                          # workgroups that stamped this stage (a workgroup without rows in a stage skips it)
    m = (x[7] > 0) & (x[0] > 0)
    if grp is not None:
        g_ = torch.arange(NW) // 64 == int(grp)
        m = m & g_
    return m
for s in range(NS):
    x = t[s][:, live(t[s])]
    if x.shape[1] == 0:
        print(nm(s) + "  (no workgroup of this selection has rows here)")
        continue
    d_ = [(x[k + 1] - x[k]).median().item() for k in range(7)]
    span = (x[7].max() - x[0].min()).item()
    nx_ = t[s + 1][:, live(t[s + 1])] if s + 1 < NS else None
    gap = (nx_[2].median() - x[7].median()).item() if nx_ is not None and nx_.shape[1] else float("nan")
    if s == 26:      # sampler: only the workgroups that own a row
        own = (x[4] > 0)
        x = x[:, own]
        d_ = [(x[1] - x[0]).median().item(), (x[2] - x[1]).median().item(), 0.0, (x[4] - x[2]).median().item(), 0.0,
              (x[6] - x[4]).median().item(), (x[7] - x[6]).median().item()]
    print(nm(s) + " " + " ".join(f"{v:11.2f}" for v in d_) + f" {sum(d_):8.2f} {span:8.2f} {gap:6.2f}")
    tot += sum(d_)
t0_, t1_ = t[0][0][t[0][0] > 0].min().item(), t[NS - 1][7].max().item()
print(f"whole pass: {t1_ - t0_:.1f} us for {NS} stages = {(t1_ - t0_) / NS:.2f} us per stage")
