"""Round 6 diagnostic: where the persistent chains and the launch path part (code predictor alone): per group, rows / logits that differ.
usage: python scripts/diag_chain_vs_launch.py [model] [B]"""
import ctypes as C, os, sys
os.environ["OMNI_TALKER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd import _lib as L
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.weights import make_weights
model = sys.argv[1] if len(sys.argv) > 1 else "tts-0.6b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = get_dims(model).with_(layers=1, max_model_len=256)
w = make_weights(d, seed=33, std=0.02)
g = torch.Generator().manual_seed(B)
code0 = torch.randint(1, d.codebook, (B,), generator=g)
e0, lh = w["embed"][code0], torch.randn(B, d.hidden, generator=g).to(torch.bfloat16)
lib = L.load()
def knob(name, *v):
    f = getattr(lib, "omni_debug_" + name); f.argtypes = [C.c_int] * len(v); f.restype = None; f(*v)
def run(cp_chain, pair):
    knob("cp_chain", cp_chain); knob("chain_pair", pair)
    eng = TalkerEngine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=64)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    torch.cuda.synchronize()
    return codes.cpu(), lg.cpu()
ref = run(0, 0)
for name, cfg in (("per-pass chain", (1, 0)), ("all-pass chain, launch pair", (2, 0)), ("all-pass chain + pair kernel", (2, 1))):
    c, lg = run(*cfg)
    dif = (lg != ref[1])
    print(f"{name}: codes equal {bool((c == ref[0]).all())}; groups with differing logits {dif.any(-1).any(0).nonzero().flatten().tolist()}; "
          f"rows {dif.any(-1).any(1).nonzero().flatten().tolist()}; max |diff| {float((lg - ref[1]).abs().max()):.4g}; differing logits {int(dif.sum())}")
    if dif.any():
        gi = int(dif.any(-1).any(0).nonzero()[0])
        rows = dif[:, gi].any(-1).nonzero().flatten().tolist()
        print(f"   first differing group {gi + 1}: rows {rows}, logits differing there {int(dif[:, gi].sum())}")
