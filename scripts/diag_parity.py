"""Print end-to-end error statistics GPU vs oracle (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from tests import test_gpu_engine as T
from oracle import talker_oracle as O

def stats(name, g, o):
    g, o = g.float().nan_to_num(neginf=0), o.float().nan_to_num(neginf=0)
    d = (g - o).abs()
    rel = d / o.abs().clamp_min(2 ** -6)
    print(f"  {name:16s} max|d|={d.max():.4g} mean|d|={d.mean():.3g} max rel={rel.max():.3g} exact={(d == 0).float().mean():.3f} |ref|max={o.abs().max():.3g}")

T.assert_bf16_close = lambda *a, **k: None
for kv in ("bf16", "fp8", "int8"):
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.06, norm_noise=0.1)
    rec = T._scenario(d, w, kv, prompt_lens=[5, 17, 33, 16], n_steps=6)
    print("tiny", kv)
    stats("prefill_logits", *rec["prefill_logits"])
    for i, st in enumerate(rec["steps"]):
        print(f" step {i}: slots eq {torch.equal(*st['slots'])} codes eq {torch.equal(*st['codes'])} ids eq {torch.equal(st['ids'][0].long(), st['ids'][1].long())}")
        stats("logits", *st["logits"]); stats("hidden", *st["hidden"])
d = get_dims("tts-1.7b").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
w = make_weights(d, seed=8, std=0.02)
g = torch.Generator().manual_seed(0)
lens = torch.randint(4, 40, (64,), generator=g).tolist()
rec = T._scenario(d, w, "fp8", prompt_lens=lens, n_steps=2, num_blocks=300)
print("real dims 1 layer")
stats("prefill_logits", *rec["prefill_logits"])
for i, st in enumerate(rec["steps"]):
    print(f" step {i}: slots eq {torch.equal(*st['slots'])} codes eq {torch.equal(*st['codes'])} ids eq {torch.equal(st['ids'][0].long(), st['ids'][1].long())}")
    stats("logits", *st["logits"]); stats("hidden", *st["hidden"])
# code predictor logits
d = get_dims("tiny"); w = make_weights(d, seed=3, std=0.08, norm_noise=0.1)
eng = T._engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16); orc = O.TalkerOracle(d, w)
g = torch.Generator().manual_seed(16); B = 16
code0 = torch.randint(1, d.codebook, (B,), generator=g); e0 = w["embed"][code0]; lh = torch.randn(B, d.hidden, generator=g).to(torch.bfloat16)
codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
rc, rl = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
print("cp codes eq", torch.equal(codes.cpu(), rc)); stats("cp logits", lg.cpu(), rl)
