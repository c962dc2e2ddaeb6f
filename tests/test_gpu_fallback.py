"""A decode step whose persistent chain reports a timed-out flag wait is REDONE on the launch-per-op path, not handed on
(VERDICT r3 weak #3 / ADVICE r3): the step's status words ride in the runner's per-step host copy (omni_step_io.status, ABI v4),
the engine clears the device words, turns its chains off, the runner re-captures its graphs, restores every decode row from its
host records and runs the step again.  Checked against a clean runner fed the same requests: every sampled id, code frame and
hidden state of every step is identical (the chains and the launch path compute the same bits).
Reference behaviour being protected: gpu_ar_model_runner.py:403-660 hands every step's outputs to the scheduler."""
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.engine import TalkerEngine
from ht_vllm_omni_amd.payloads import OmniCachedRequestData, OmniNewRequestData, OmniSchedulerOutput, SamplingParams, encode_tensor
from ht_vllm_omni_amd.runner import MI355XARModelRunner
from ht_vllm_omni_amd.sched import BlockPool
from ht_vllm_omni_amd.weights import make_weights

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _drive(d, w, B, steps, fault_at, graphs, async_on=False):
    bs, nb = 16, 4 * B + 8
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=nb, block_size=bs, max_batch=64)
    eng.set_sampling(cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
    run = MI355XARModelRunner(eng, use_graphs=graphs, async_scheduling=async_on)
    if graphs:
        run.capture_graphs([64])
    pool = BlockPool(nb, bs)
    g = torch.Generator().manual_seed(11)
    sp = SamplingParams(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=5)
    keys = [f"r{i}" for i in range(B)]
    plen = {k: 3 + (i % 5) for i, k in enumerate(keys)}
    new = []
    for k in keys:
        pool.allocate(k, plen[k] + 1)
        info = {"talker_prompt_embeds": encode_tensor((torch.randn(plen[k], d.hidden, generator=g) * 0.5).to(BF16)),
                "tts_pad_embed": encode_tensor((torch.randn(d.hidden, generator=g) * 0.02).to(BF16)),
                "tailing_text_hidden": encode_tensor((torch.randn(2, d.hidden, generator=g) * 0.02).to(BF16))}
        new.append(OmniNewRequestData(req_id=k, prompt_token_ids=[d.codec_pad_id] * plen[k], block_ids=(pool.block_ids(k),),
                                      sampling_params=sp, additional_information=info))
    so = OmniSchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens=dict(plen), total_num_scheduled_tokens=sum(plen.values()))
    assert run.execute_model(so) is None
    handles = [run.sample_tokens(None)]
    outs, ran = [], []
    for s in range(steps):
        nbk = []
        for k in keys:
            nb_new = pool.allocate(k, run.requests[k].num_computed + 2)
            nbk.append((nb_new,) if nb_new else None)
        so = OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=list(keys), new_block_ids=nbk),
                                 num_scheduled_tokens={k: 1 for k in keys}, total_num_scheduled_tokens=B)
        if s == fault_at:
            torch.cuda.synchronize()
            eng.chain_error(reset=2)          # as if a flag wait had timed out: the step's chains stop waiting -> garbage
        run.execute_model(so)
        handles.append(run.sample_tokens(None))
        if async_on:                          # the engine core's order: the step just dispatched runs while its predecessor is read
            outs.append(handles[-2].get_output())
            torch.cuda.synchronize()
        else:
            outs.append(handles[-2])
        ran.append(int(eng.status[2]))
    outs.append(handles[-1].get_output() if async_on else handles[-1])
    return outs, ran, run, eng


@pytest.mark.parametrize("graphs,async_on", [(True, False), (False, False), (True, True)])
def test_chain_timeout_falls_back_to_the_launch_path_and_redoes_the_step(graphs, async_on):
    """async_on: the status word of step t is read in get_output(t) -- after step t + 1 was dispatched on step t's garbage: both
    are redone, t from the host records of t - 1, t + 1 from the records the redone t wrote."""
    d = get_dims("tts-1.7b").with_(layers=2, max_model_len=256)
    w = make_weights(d, seed=12, std=0.02)
    B, steps = 64, 5
    clean, ran0, run0, eng0 = _drive(d, w, B, steps, fault_at=-1, graphs=graphs, async_on=async_on)
    assert all(r == 3 for r in ran0), f"both chains must run at the 1.7B shape with 64 rows (chains_ran per step: {ran0})"
    assert getattr(run0, "chain_fallbacks", 0) == 0
    hurt, ran1, run1, eng1 = _drive(d, w, B, steps, fault_at=2, graphs=graphs, async_on=async_on)
    assert run1.chain_fallbacks == 1
    k = 3 if async_on else 2       # async: the step behind the faulted one had been dispatched (chains still on) before the word was read
    assert all(r == 3 for r in ran1[:k - (1 if async_on else 0)]) and all(r == 0 for r in ran1[k:]), f"after the fall-back the steps run launch-per-op: {ran1}"
    assert eng1.chain_error() == 0 and not eng1.persistent_chains
    for s, (a, b) in enumerate(zip(clean, hurt)):
        assert a.req_ids == b.req_ids
        assert a.sampled_token_ids == b.sampled_token_ids, f"step {s}: sampled ids differ after the fall-back"
        for i in range(len(a.req_ids)):
            assert torch.equal(a.pooler_output[i]["audio_codes"], b.pooler_output[i]["audio_codes"]), f"step {s}: codes of {a.req_ids[i]}"
            assert torch.equal(a.pooler_output[i]["hidden"], b.pooler_output[i]["hidden"]), f"step {s}: hidden of {a.req_ids[i]}"
    # the KV the invalid step wrote was written again by the redone step: caches identical
    for l, (x, y) in enumerate(zip(eng0.kv_caches, eng1.kv_caches)):
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), f"KV cache of layer {l} differs after the redone step"


def test_status_words_report_what_ran():
    """omni_step_io.status[2] / omni_talker_chains_ran: the smoke test and the bench report what actually ran, not what was asked
    for: the 1.7B shape runs both chains at every batch size (49-64 rows since round 3, 1-32 since round 4, 33-48 since round 5), 4 code
    groups only the backbone's."""
    d = get_dims("tts-1.7b").with_(layers=1, max_model_len=256)
    w = make_weights(d, seed=12, std=0.02)
    for B, Q, want in ((64, 16, 3), (40, 16, 3), (16, 16, 3), (64, 4, 2)):
        dd = d.with_(num_code_groups=Q)
        ww = make_weights(dd, seed=12, std=0.02)
        eng = TalkerEngine(dd, ww, kv_dtype="fp8", num_blocks=2 * 64 + 2, block_size=16, max_batch=64)
        for b in range(B):
            eng.block_table[b, :2] = torch.tensor([1 + 2 * b, 2 + 2 * b], dtype=torch.int32)
        eng.input_ids[:B] = 5
        eng.positions[:B] = 3
        eng.seq_lens[:B] = 4
        eng.decode_step(B)
        torch.cuda.synchronize()
        assert eng.chains_ran() == want, (B, Q, eng.chains_ran())
        assert eng.status.cpu().tolist() == [0, 0, want, 0]
