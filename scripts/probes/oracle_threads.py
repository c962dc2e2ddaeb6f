"""How long do the oracle pieces of the full-depth GPU parity tests take on the GPU box's host, by torch thread count?  (They took
120 s there and 25 s on an 8-core container: tests/conftest.py caps the threads.)  usage: python scripts/probes/oracle_threads.py [threads]"""
import os, time, torch, numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
print("threads", torch.get_num_threads(), "cpus", os.cpu_count())
from oracle import talker_oracle as O
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
BF16=torch.bfloat16
d = get_dims("tts-1.7b").with_(max_model_len=512)
w = make_weights(d, seed=1234, std=0.02)
B, bs, nb = 64, 16, 2 * 64 + 2
g = torch.Generator().manual_seed(3)
lens = torch.randint(6, 22, (B,), generator=g).tolist()
prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
bts = [[1 + 2 * r, 2 + 2 * r] for r in range(B)]
x = torch.cat(prompts, 0)
pos = torch.cat([torch.arange(n) for n in lens])
req = [r for r, n in enumerate(lens) for _ in range(n)]
last = torch.tensor(np.cumsum(lens) - 1)
t0=time.time()
orc = O.TalkerOracle(d, w, kv_dtype="fp8", num_blocks=nb, block_size=bs)
print("ctor", time.time()-t0); t0=time.time()
h = orc.backbone(x, pos, req, bts, lens)
print("backbone prefill bf16", time.time()-t0, x.shape); t0=time.time()
ids0 = torch.randint(0, 2048, (B,))
ref_codes, ref_lg = orc.code_predictor(ids0, w["embed"][ids0], h[last], do_sample=False, return_logits=True)
print("code predictor 64 rows", time.time()-t0); t0=time.time()
xt = torch.randn(B, d.hidden).to(BF16)
hd = orc.backbone(xt, torch.tensor(lens), list(range(B)), bts, [n + 1 for n in lens])
print("backbone decode bf16", time.time()-t0); t0=time.time()
keep=O.BF16; O.BF16=torch.float32
o2 = O.TalkerOracle(d, w, kv_dtype="fp8", num_blocks=nb, block_size=bs)
o2.backbone(x.float(), pos, req, bts, lens)
print("backbone prefill fp32", time.time()-t0); t0=time.time()
O.BF16=keep
