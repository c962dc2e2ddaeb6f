"""Stage-by-stage run of the batched MoE prefill MLP at a W3-sized prompt (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd import ops
import torch.nn.functional as F
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6438
d = get_dims("omni-talker").with_(layers=1, cp_layers=1, num_code_groups=3)
w = make_weights(d, seed=1, std=0.02)
eng = TalkerEngine(d, w, kv_dtype="int8", num_blocks=1024, block_size=16, max_batch=64)
lw = eng.layer_w[0]
a = (torch.randn(T, d.hidden, device="cuda") * 0.5).to(torch.bfloat16)
def stage(name, fn):
    out = fn(); torch.cuda.synchronize(); print("ok", name, flush=True); return out
logits = stage("router", lambda: F.linear(a, lw["moe_router"]))
idx, wts = stage("route", lambda: ops.moe_route(logits, d.moe_top_k, d.moe_norm_topk))
print(int(idx.min()), int(idx.max()))
out = stage("moe_mlp_blas", lambda: eng._moe_mlp_blas(a, lw))
print(out.float().abs().mean().item())
