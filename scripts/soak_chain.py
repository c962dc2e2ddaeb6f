"""Soak of the persistent chains: two engines fed the same requests replay a captured decode step N times each (flags, epochs and
flag copies advance on the device across replays); every step's codes must agree between the engines, the error word must stay 0.
usage: python scripts/soak_chain.py [--steps 1500] [--batch 64 48 17]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--batch", type=int, nargs="+", default=[64, 48, 17])
a = ap.parse_args()
d = get_dims("tts-1.7b").with_(layers=4, max_model_len=4096)
w = make_weights(d, seed=4, std=0.02)
for B in a.batch:
    outs = []
    for _ in range(2):
        eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=64 * 130 + 2, max_batch=64)
        g = torch.Generator().manual_seed(9)
        eng.input_ids[:B] = torch.randint(1, d.codebook, (B,), generator=g).to(torch.int32).cuda()
        eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(torch.bfloat16).cuda()
        eng.text_step[:B] = (torch.randn(B, d.hidden, generator=g) * 0.02).to(torch.bfloat16).cuda()
        eng.positions[:B] = 17
        eng.seq_lens[:B] = 18
        nb = (18 + a.steps + 40) // 16 + 1
        for b in range(B):
            eng.block_table[b, :nb] = torch.arange(1 + nb * b, 1 + nb * (b + 1), dtype=torch.int32)
        eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
        eng.decode_step(B); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eng.decode_step(B)
        hist = torch.empty(a.steps, B, d.num_code_groups, dtype=torch.int64, device="cuda")
        for s in range(a.steps):
            gr.replay()
            hist[s] = eng.audio_codes[:B]
        torch.cuda.synchronize()
        assert eng.chain_error() == 0, f"B={B}: chain error word {eng.chain_error():#x}"
        assert eng.persistent_chains
        outs.append(hist.cpu())
        del eng, gr
    same = torch.equal(outs[0], outs[1])
    print(f"B={B}: {a.steps} replayed steps x 2 engines, codes identical: {same}, error word 0, distinct frames {len(set(map(tuple, outs[0][:, 0].tolist())))}")
    assert same
