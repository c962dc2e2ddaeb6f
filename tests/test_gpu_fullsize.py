"""BASELINE-size checks (W3: Qwen3-TTS-1.7B shape, 28 layers, B = 64, fp8 KV, sampled decoding) through properties that do
not need the CPU oracle at that size:
  * permutation of the batch rows permutes the outputs and nothing else (bit-exact): no kernel lets a row see its launch
    mates -- m-split tiles, the two-block pair pass, fragment-major padding rows, sampler, KV writes;
  * a request run alone (B = 1) has the prompt cache bytes and first token it has inside the batch of 64, and decode logits
    within rounding (B = 1 takes the KV-split attention: another summation order);
  * a decode step writes exactly the KV slots  block_table[r][pos / 16] * 16 + pos % 16  of every layer and no other byte;
  * replaying the run reproduces every bit.
Small-size parity against the oracle and the golden vectors is in test_gpu_engine.py / test_gpu_ops.py."""
import numpy as np
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.sched import BlockPool
from ht_vllm_omni_amd.weights import make_weights

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
BS, NB, STEPS = 16, 1024, 3


@pytest.fixture(scope="module")
def w3():
    d = get_dims("tts-1.7b")
    return d, make_weights(d, seed=1234, std=0.02)


def _drive(d, w, req_ids, row_of, *, snapshot_kv=False, kv="fp8", cp_top_p=1.0):
    """Requests req_ids (global ids: prompt content and length are functions of the id) placed on rows row_of[i]; native
    prefill, STEPS sampled decode steps.  Returns per request: logits / codes / ids of every step (+ touched KV slots)."""
    from ht_vllm_omni_amd.engine import TalkerEngine
    from ht_vllm_omni_amd import ops
    B = len(req_ids)
    eng = TalkerEngine(d, w, kv_dtype=kv, num_blocks=NB, block_size=BS, max_batch=B, device="cuda:0", allow_eos=False)
    eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50,
                     **({"cp_top_p": cp_top_p} if cp_top_p < 1.0 else {}))
    lens = {q: 32 + (q * 37) % 129 for q in req_ids}
    prompt = {q: (torch.randn(lens[q], d.hidden, generator=torch.Generator().manual_seed(1000 + q)) * 0.05).to(BF16) for q in req_ids}
    pad = {q: (torch.randn(d.hidden, generator=torch.Generator().manual_seed(5000 + q)) * 0.05).to(BF16) for q in req_ids}
    # blocks: request q always gets the same block ids (so cache bytes can be compared across runs)
    blocks = {q: list(range(1 + q * 14, 1 + q * 14 + 14)) for q in req_ids}
    order = sorted(range(B), key=lambda i: row_of[i])            # prefill in row order
    bt = torch.zeros(eng.max_batch, eng.bt_stride, dtype=torch.int32)
    for i, q in enumerate(req_ids):
        bt[row_of[i], :14] = torch.tensor(blocks[q], dtype=torch.int32)
    eng.block_table.copy_(bt)
    x = torch.cat([prompt[req_ids[i]] for i in order]).cuda()
    pos = torch.cat([torch.arange(lens[req_ids[i]]) for i in order]).to(torch.int32)
    req = torch.cat([torch.full((lens[req_ids[i]],), row_of[i]) for i in order]).to(torch.int32)
    slots = torch.tensor([int(bt[int(req[t]), int(pos[t]) // BS]) * BS + int(pos[t]) % BS for t in range(len(pos))])
    hid = eng.prefill(x, pos.cuda(), req.cuda(), slots.cuda(), use_blas=False)
    last = torch.tensor(np.cumsum([lens[req_ids[i]] for i in order]) - 1).cuda()
    rows = torch.tensor([row_of[i] for i in order]).cuda()
    hl = torch.zeros(B, d.hidden, dtype=BF16, device="cuda")
    hl[rows] = hid[last]
    eng.seen.zero_(); eng.seen[:B, d.codec_pad_id] = 1; eng.steps.zero_()
    s = eng.sampling
    ids = ops.sample(eng.compute_logits(hl), greedy=False, temperature=s["temperature"], top_k=s["top_k"], rep_penalty=s["rep_penalty"],
                     seen=eng.seen, seed=s["seed"], steps=eng.steps, inc_steps=True)
    eng.input_ids[:B] = ids
    eng.last_hidden[:B] = hl
    ln = torch.zeros(B, dtype=torch.int32)
    ts = torch.zeros(B, d.hidden, dtype=BF16)
    for i, q in enumerate(req_ids):
        ln[row_of[i]] = lens[q]
        ts[row_of[i]] = pad[q]
    eng.positions[:B] = ln.cuda(); eng.seq_lens[:B] = (ln + 1).cuda(); eng.text_step[:B] = ts.cuda()
    out = {q: {"logits": [], "codes": [], "ids": [], "first_id": int(ids[row_of[i]])} for i, q in enumerate(req_ids)}
    kv_before = [c.clone() for c in (eng.kv_caches[0], eng.kv_caches[d.layers - 1])] if snapshot_kv else None
    for _ in range(STEPS):
        eng.decode_step(B)
        torch.cuda.synchronize()
        for i, q in enumerate(req_ids):
            r = row_of[i]
            out[q]["logits"].append(eng.logits[r].cpu().clone())
            out[q]["codes"].append(eng.audio_codes[r].cpu().clone())
            out[q]["ids"].append(int(eng.input_ids[r]))
    extra = {}
    if snapshot_kv:
        changed = []
        for before, after in zip(kv_before, (eng.kv_caches[0], eng.kv_caches[d.layers - 1])):
            diff = (before != after).reshape(2, NB * BS, -1).any(-1).any(0)            # per slot: any byte of K or V changed
            changed.append(set(diff.nonzero().flatten().tolist()))
        want = {blocks[q][(lens[q] + t) // BS] * BS + (lens[q] + t) % BS for q in req_ids for t in range(STEPS)}
        extra = {"changed": changed, "want": want}
    for q in req_ids:   # the request's own cache bytes (layer 0 and last), positions 0 .. len + STEPS
        n = lens[q] + STEPS
        sl = torch.tensor([blocks[q][p // BS] * BS + p % BS for p in range(n)]).cuda()
        out[q]["kv"] = [c.reshape(2, NB * BS, -1)[:, sl].cpu().clone() for c in (eng.kv_caches[0], eng.kv_caches[d.layers - 1])]
    del eng
    torch.cuda.empty_cache()
    return out, extra


def _same(a, b, what):
    for key in ("logits", "codes"):
        for t, (x, y) in enumerate(zip(a[key], b[key])):
            assert torch.equal(x, y), f"{what}: {key} differ at step {t}"
    assert a["ids"] == b["ids"] and a["first_id"] == b["first_id"], what
    for x, y in zip(a["kv"], b["kv"]):
        assert torch.equal(x, y), f"{what}: cache bytes differ"


def test_w3_full_size_row_permutation_solo_request_slots_and_replay(w3):
    d, w = w3
    reqs = list(range(64))
    base, extra = _drive(d, w, reqs, list(range(64)), snapshot_kv=True)
    # exactly the expected slots of the first and last layer were written by the decode steps
    for changed in extra["changed"]:
        assert changed == extra["want"], (len(changed), len(extra["want"]))
    # finite logits on the allowed ids, -inf elsewhere; codes inside the codebook
    lg = base[7]["logits"][0]
    assert torch.isfinite(lg).sum().item() == 2047 and (base[7]["codes"][0][1:] < d.codebook).all()
    # replay
    again, _ = _drive(d, w, reqs, list(range(64)))
    for q in reqs:
        _same(base[q], again[q], f"replay, request {q}")
    # row permutation (also moves requests across the 16-row tiles and the m-split halves)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(3)).tolist()
    moved, _ = _drive(d, w, reqs, perm)
    for q in reqs:
        _same(base[q], moved[q], f"row permutation, request {q} (row {q} -> {perm[q]})")
    # the same request alone: the prompt's cache bytes and the first sampled id are the batch's bit for bit (row-independent
    # prefill); the decode steps run the KV-split attention at B = 1 (8 workgroups would not fill the chip otherwise), i.e.
    # another fp32 summation order, so logits agree to rounding, not bit for bit -- and only while the sampled codes agree
    from tests.util import assert_e2e_close
    lens = {q: 32 + (q * 37) % 129 for q in reqs}
    for q in (5, 40):
        solo, _ = _drive(d, w, [q], [0])
        assert solo[q]["first_id"] == base[q]["first_id"], q
        for x, y in zip(base[q]["kv"], solo[q]["kv"]):
            assert torch.equal(x[:, :lens[q]], y[:, :lens[q]]), f"request {q}: prompt cache bytes, alone vs in the batch"
        if torch.equal(base[q]["codes"][0], solo[q]["codes"][0]):
            a, b = solo[q]["logits"][0], base[q]["logits"][0]
            fin = torch.isfinite(b)
            dd = (a[fin] - b[fin]).abs()
            print(f"request {q}: alone vs batch, step 0: mean |diff| {dd.mean().item():.4g}, max {dd.max().item():.4g}, scale {b[fin].abs().max().item():.3g}")
            # 28 bf16 layers deep, one differing summation order per layer: a few ulps at the largest magnitude
            assert_e2e_close(a, b, mean_tol=3e-2, max_ulps=8, what=f"request {q} alone vs in the batch, step 0")


def test_omni_talker_full_size_row_permutation_and_replay():
    """The same properties for BASELINE config #4's talker: 20 sparse-MoE layers (128 experts, top-8, gated shared expert),
    16 q / 2 kv heads, int8 KV with per-token scales, code predictor with top-k 50 + top-p 0.8, B = 64."""
    d = get_dims("omni-talker")
    w = make_weights(d, seed=1234, std=0.02)
    reqs = list(range(64))
    base, extra = _drive(d, w, reqs, list(range(64)), snapshot_kv=True, kv="int8", cp_top_p=0.8)
    for changed in extra["changed"]:
        assert changed == extra["want"], (len(changed), len(extra["want"]))
    again, _ = _drive(d, w, reqs, list(range(64)), kv="int8", cp_top_p=0.8)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(4)).tolist()
    moved, _ = _drive(d, w, reqs, perm, kv="int8", cp_top_p=0.8)
    for q in reqs:
        _same(base[q], again[q], f"replay, request {q}")
        _same(base[q], moved[q], f"row permutation, request {q} (row {q} -> {perm[q]})")
