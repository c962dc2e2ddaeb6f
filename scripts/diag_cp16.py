import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd.engine import TalkerEngine
from oracle import talker_oracle as O
BF16 = torch.bfloat16
d = get_dims("tts-1.7b").with_(layers=1, cp_layers=2, max_model_len=256)
w = make_weights(d, seed=11, std=0.02)
B = 9
g = torch.Generator().manual_seed(B)
code0 = torch.randint(1, d.codebook, (B,), generator=g)
e0 = w["embed"][code0]
lh = torch.randn(B, d.hidden, generator=g).to(BF16)
orc = O.TalkerOracle(d, w)
ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
res = {}
for name, kw in (("default", {}), ("separate-norms", dict(fused_norm=False)), ("row-major", dict(frag_layout=False))):
    eng = TalkerEngine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16, **kw)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    lg = lg.cpu(); res[name] = lg
    same = (codes.cpu() == ref_codes)
    dd = (lg - ref_lg).abs()
    print(name, "codes equal", same.float().mean().item(), "mean diff", dd.mean().item(), "max", dd.max().item(),
          "per-group mean", [round(dd[:, q].mean().item(), 5) for q in (0, 1, 7, 14)])
print("default vs separate", (res["default"] - res["separate-norms"]).abs().mean().item())
