"""A/B of the end-to-end prefill deviation from the oracle: MFMA prefill attention vs the per-token VALU path."""
import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ht_vllm_omni_amd import _lib as L
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.sched import BlockPool
from oracle import talker_oracle as O
BF16 = torch.bfloat16
lib = L.load()
lib.omni_debug_prefill_mfma.argtypes = [ctypes.c_int]; lib.omni_debug_prefill_mfma.restype = None
d = get_dims("tts-1.7b").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
w = make_weights(d, seed=8, std=0.02)
for seed in (0, 1, 2):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(4, 40, (64,), generator=g).tolist()
    bs, nbk = 16, 300
    orc = O.TalkerOracle(d, w, kv_dtype="fp8", num_blocks=nbk, block_size=bs)
    pool = BlockPool(nbk, bs)
    prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
    bts = []
    for r, n in enumerate(lens):
        pool.allocate(f"r{r}", n + 4); bts.append(pool.block_ids(f"r{r}"))
    states = [O.OracleState() for _ in lens] if hasattr(O, "OracleState") else None
    x = torch.cat(prompts); pos = torch.cat([torch.arange(n) for n in lens]); req = [r for r, n in enumerate(lens) for _ in range(n)]
    o_h = orc.backbone(x, pos, req, bts, lens)
    last = torch.tensor(np.cumsum(lens) - 1)
    o_lg = orc.compute_logits(o_h[last])
    slots = torch.tensor([bts[r][int(p) // bs] * bs + int(p) % bs for r, p in zip(req, pos.tolist())])
    for on in (1, 0):
        lib.omni_debug_prefill_mfma(on)
        eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=nbk, block_size=bs, max_batch=64)
        bt = torch.zeros(64, eng.bt_stride, dtype=torch.int32)
        for r, ids in enumerate(bts): bt[r, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        eng.block_table.copy_(bt)
        for blas in (True, False):
            hid = eng.prefill(x.cuda(), pos.to(torch.int32).cuda(), torch.tensor(req, dtype=torch.int32).cuda(), slots.cuda(), use_blas=blas)
            lg = eng.compute_logits(hid[last.cuda()]).cpu()
            fin = torch.isfinite(o_lg)
            print(f"seed {seed} mfma={on} blas={blas}: hidden mean diff {(hid.cpu().float() - o_h.float()).abs().mean().item():.5f}  logits mean diff {(lg[fin] - o_lg[fin]).abs().mean().item():.5f}")
