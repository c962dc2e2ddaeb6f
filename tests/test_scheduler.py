"""Row a12: the AR-stage scheduler (vLLM V1 admission / block allocation / stop check as stated in SURVEY Appendix A,
plus the Omni KV hand-off protocol of V/core/sched/omni_ar_scheduler.py) driving the runner -- host logic, no GPU."""
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.connectors import InProcConnector, OmniKVTransferManager
from ht_vllm_omni_amd.payloads import SamplingParams, encode_tensor
from ht_vllm_omni_amd.runner import MI355XARModelRunner
from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, RequestStatus, TalkerStageEngine
from ht_vllm_omni_amd.stage_input_processors import CodecChunkStreamer
from tests.fakes import FakeEngine

BF16 = torch.bfloat16


def _request(d, rid, n_prompt, *, max_tokens=8, stop=(), tail=1, seed=0):
    g = torch.Generator().manual_seed(seed + n_prompt)
    info = {"talker_prompt_embeds": encode_tensor(torch.randn(n_prompt, d.hidden, generator=g).to(BF16)),
            "tailing_text_hidden": encode_tensor(torch.randn(tail, d.hidden, generator=g).to(BF16)),
            "tts_pad_embed": encode_tensor(torch.zeros(d.hidden).to(BF16))}
    sp = SamplingParams(temperature=0.0, max_tokens=max_tokens, stop_token_ids=tuple(stop))
    return Request(request_id=rid, num_prompt_tokens=n_prompt, prompt_token_ids=[d.codec_pad_id] * n_prompt,
                   sampling_params=sp, additional_information=info)


class _Worker:
    def __init__(self, runner):
        self.model_runner = runner

    def execute_model(self, so):
        return self.model_runner.execute_model(so)

    def sample_tokens(self, g):
        return self.model_runner.sample_tokens(g)


def test_admission_budget_chunked_prefill_and_block_order():
    d = get_dims("tiny")
    s = MI355XARScheduler(num_blocks=32, block_size=16, max_num_seqs=2, max_num_batched_tokens=24, max_model_len=512)
    for rid, n in (("a", 5), ("b", 40), ("c", 3)):
        s.add_request(_request(d, rid, n))
    so = s.schedule()
    # FCFS under a 24-token budget: a whole (5), b chunked to the remaining 19; c waits (max_num_seqs = 2)
    assert so.num_scheduled_tokens == {"a": 5, "b": 19} and so.total_num_scheduled_tokens == 24
    assert [r.req_id for r in so.scheduled_new_reqs] == ["a", "b"]
    # block 0 is the null block; ids in queue order; b holds ceil(19 / 16) = 2 blocks so far
    assert so.scheduled_new_reqs[0].block_ids == ([1],) and so.scheduled_new_reqs[1].block_ids == ([2, 3],)
    assert s.requests["b"].num_computed_tokens == 19 and len(s.waiting) == 1
    with pytest.raises(ValueError):
        s.add_request(_request(d, "a", 4))
    with pytest.raises(ValueError):
        s.add_request(Request("z", num_prompt_tokens=0))


def test_preemption_when_the_pool_runs_dry():
    d = get_dims("tiny")
    s = MI355XARScheduler(num_blocks=4, block_size=16, max_num_seqs=4, max_num_batched_tokens=64, max_model_len=512)   # 3 usable blocks
    s.add_request(_request(d, "a", 16, max_tokens=40))
    s.add_request(_request(d, "b", 30, max_tokens=40))
    so = s.schedule()
    assert so.num_scheduled_tokens == {"a": 16, "b": 30}
    for r in ("a", "b"):
        s.requests[r].output_token_ids.append(7)       # both sampled their first token
    so = s.schedule()                                   # a needs a 2nd block for token 17: none free -> b (newest) is preempted
    assert so.preempted_req_ids == {"b"} and so.num_scheduled_tokens == {"a": 1}
    assert s.requests["b"].status == RequestStatus.PREEMPTED and s.requests["b"].num_computed_tokens == 0
    assert s.waiting[0].request_id == "b" and not so.scheduled_new_reqs     # no admission in a preempting step
    assert s.pool.block_ids("a") == [1, 3]        # b's blocks went back in reverse order: 3 is at the head


def test_engine_loop_stop_conditions_kv_handoff_and_chunk_stream():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    conn = InProcConnector()
    runner = MI355XARModelRunner(eng, kv_transfer=OmniKVTransferManager(conn), use_graphs=False)
    chunk_conn = InProcConnector()
    streamer = CodecChunkStreamer(codec_chunk_frames=4, codec_left_context_frames=2, max_num_seqs=4,
                                  num_quantizers=d.num_code_groups, connector=chunk_conn)
    sched = MI355XARScheduler(num_blocks=32, block_size=16, max_num_seqs=4, max_num_batched_tokens=64, max_model_len=512,
                              need_send_cache=True, chunk_streamer=streamer)
    core = TalkerStageEngine(_Worker(runner), sched)
    core.add_request(_request(d, "a", 5, max_tokens=6))
    core.add_request(_request(d, "b", 20, max_tokens=3))
    outs = core.step()                                   # prefill of both, first tokens sampled
    assert {o.request_id: len(o.new_token_ids) for o in outs} == {"a": 1, "b": 1}
    tokens = {o.request_id: list(o.new_token_ids) for o in outs}
    finished = {}
    for _ in range(20):
        for o in core.step():
            tokens[o.request_id] += o.new_token_ids
            if o.finished:
                finished[o.request_id] = o
        if not sched.has_unfinished_requests() and not sched.waiting_for_transfer_free and not sched.requests_needing_kv_transfer:
            break
    assert len(tokens["a"]) == 6 and len(tokens["b"]) == 3
    assert finished["a"].finish_reason == "length" and finished["b"].finish_reason == "length"
    # finished requests shipped their KV (block list truncated to the sequence), were acked, and gave their blocks back
    assert finished["b"].kv_transfer_params["kv_metadata"]["seq_len"] == 20 + 2     # the last sampled token was never computed
    kv_b, _ = conn.get("0", "1", "omni_0_to_1_kv_cache_b")
    assert kv_b["metadata"]["seq_len"] == finished["b"].kv_transfer_params["kv_metadata"]["seq_len"]
    assert len(finished["b"].kv_transfer_params["past_key_values"]) == (kv_b["metadata"]["seq_len"] + 15) // 16
    assert not sched.requests and sched.pool.num_free == 31 and not sched.active_kv_transfers
    assert runner.rows == []
    # the codec frames of every decode step went through the chunk connector under {req}_{stage}_{chunk}
    assert "a_0_0" in chunk_conn.store and "a" not in streamer.code_prompt_token_ids
    first, _ = chunk_conn.get("0", "1", "a_0_0")
    assert len(first["code_predictor_codes"]) % d.num_code_groups == 0 and first["left_context_size"] == 0


def test_stop_token_and_eos_and_abort():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    runner = MI355XARModelRunner(eng, use_graphs=False)
    sched = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512)
    core = TalkerStageEngine(_Worker(runner), sched)
    probe = _request(d, "p", 6, max_tokens=5)
    core.add_request(probe)
    toks = core.run()["p"]                                # deterministic fake engine: learn its token stream
    assert len(toks) == 5 and probe.status == RequestStatus.FINISHED_LENGTH_CAPPED
    r2 = _request(d, "q", 6, max_tokens=50, stop=(toks[2],))
    core.add_request(r2)
    toks2 = core.run()["q"]
    assert toks2 == toks[:3] and r2.status == RequestStatus.FINISHED_STOPPED and r2.stop_reason == toks[2]
    r3 = _request(d, "e", 6, max_tokens=50)
    r3.eos_token_id = toks[1]
    core.add_request(r3)
    assert core.run()["e"] == toks[:2] and r3.get_finished_reason() == "stop" and r3.stop_reason is None
    r4 = _request(d, "x", 6, max_tokens=50)
    core.add_request(r4)
    core.step()
    sched.abort_request("x")
    assert r4.get_finished_reason() == "abort" and not sched.has_unfinished_requests()
    core.step()                                           # the runner drops the row on the next step
    assert runner.rows == [] and sched.pool.num_free == 15


def test_kv_transfer_criteria_trigger_once_without_stopping():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    conn = InProcConnector()
    runner = MI355XARModelRunner(eng, kv_transfer=OmniKVTransferManager(conn), use_graphs=False)
    sched = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512,
                              kv_transfer_criteria={"type": "prefill_finished"})
    core = TalkerStageEngine(_Worker(runner), sched)
    req = _request(d, "a", 18, max_tokens=4)
    core.add_request(req)
    core.step()                                           # prefill done -> transfer marked, request keeps running
    assert "a" in sched.transfer_triggered_requests and sched.requests_needing_kv_transfer["a"] == {"seq_len": 18, "block_ids": [1, 2]}
    core.step()                                           # delivered to the runner exactly once, acked in the same step
    assert not sched.requests_needing_kv_transfer and not sched.active_kv_transfers
    kv, _ = conn.get("0", "1", "omni_0_to_1_kv_cache_a")
    assert kv["metadata"]["seq_len"] == 18
    toks = core.run()
    assert req.is_finished() and not sched.requests and sched.pool.num_free == 15     # finished later: no second transfer
    assert conn.get("0", "1", "omni_0_to_1_kv_cache_a") is None


def _run_engine(num_blocks, n_req=6, max_tokens=14):
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4, num_blocks=num_blocks)
    run = MI355XARModelRunner(eng, use_graphs=False)
    s = MI355XARScheduler(num_blocks=num_blocks, block_size=16, max_num_seqs=4, max_num_batched_tokens=48, max_model_len=512)
    core = TalkerStageEngine(_Worker(run), s)
    for i in range(n_req):
        core.add_request(_request(d, f"r{i}", 9 + 5 * i, max_tokens=max_tokens, tail=3, seed=i))
    streams, codes, preempted = {}, {}, 0
    for _ in range(400):
        so = s.schedule()
        preempted += len(so.preempted_req_ids)
        if so.total_num_scheduled_tokens == 0 and not so.finished_req_ids:
            if not s.has_unfinished_requests():
                break
            continue
        first = run.execute_model(so)
        out = first if first is not None else run.sample_tokens(None)
        for o in s.update_from_output(so, out):
            streams.setdefault(o.request_id, []).extend(o.new_token_ids)
            if o.pooling_output is not None and o.new_token_ids and o.pooling_output["audio_codes"].shape[0] == 1 \
                    and int(o.pooling_output["audio_codes"].abs().sum()) > 0:
                codes.setdefault(o.request_id, []).append(o.pooling_output["audio_codes"][0].tolist())
    return streams, codes, preempted, run


def test_preempted_request_resumes_where_it_stopped():
    """a request preempted in its decode phase is RECOMPUTED -- prompt plus the inputs of the decode
    steps it already took, rebuilt from its emitted codes and text queue -- and continues: same token stream and per-step
    audio codes as the run that never preempts (the old runner restarted it at the end of the prompt: replayed tokens,
    RNG keys and text steps from 0)."""
    ref_streams, ref_codes, p0, _ = _run_engine(num_blocks=64)
    streams, codes, p1, run = _run_engine(num_blocks=8)            # 7 usable blocks of 16 for up to 4 x (34 + 14) tokens
    assert p0 == 0 and p1 >= 1, (p0, p1)
    assert streams == ref_streams
    assert codes == ref_codes
    assert all(len(v) == 14 for v in streams.values()) and not run.preempted and not run.requests


def test_resume_restores_row_state_and_rebuilds_inputs():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=2)
    run = MI355XARModelRunner(eng, use_graphs=False)
    from ht_vllm_omni_amd.payloads import OmniCachedRequestData, OmniNewRequestData, OmniSchedulerOutput
    req = _request(d, "a", 6, tail=2)
    nr = OmniNewRequestData(req_id="a", prompt_token_ids=req.prompt_token_ids, block_ids=([1, 2],), sampling_params=req.sampling_params,
                            additional_information=req.additional_information)
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[nr], num_scheduled_tokens={"a": 6}, total_num_scheduled_tokens=6))
    toks = run.sample_tokens(None).sampled_token_ids[0]
    xs = []
    for _ in range(3):
        run.execute_model(OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["a"], new_block_ids=[None]),
                                              num_scheduled_tokens={"a": 1}, total_num_scheduled_tokens=1))
        xs.append(eng.text_step[0].clone())
        toks += run.sample_tokens(None).sampled_token_ids[0]
    st = run.requests["a"]
    state = (int(eng.positions[0]), int(eng.seq_lens[0]), int(eng.steps[0]), int(eng.input_ids[0]), eng.seen[0].clone(), eng.last_hidden[0].clone())
    assert len(toks) == 4 and len(st.codes_hist) == 3 and state[0] == 9
    # preempt, then bring it back through vLLM's own route: scheduled_cached_reqs flagged resumed_from_preemption
    run.execute_model(OmniSchedulerOutput(preempted_req_ids={"a"}))
    assert run.rows == [] and "a" in run.preempted and "a" not in run.requests
    eng.calls.clear()
    so = OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["a"], resumed_from_preemption=[True],
                                                                         new_block_ids=[([5, 6],)], num_computed_tokens=[0]),
                             num_scheduled_tokens={"a": 6 + 4}, total_num_scheduled_tokens=10)
    run.execute_model(so)
    kind, n, pos, _, slots = eng.calls[0]
    assert kind == "prefill" and n == 9 and pos == list(range(9)) and slots[:2] == [80, 81]      # prompt + 3 rebuilt inputs, new blocks
    assert eng.calls[-1][0] == "decode" and not [c for c in eng.calls if c[0] == "sample_rows"]  # nothing sampled at the seam
    out = run.sample_tokens(None)
    assert len(out.sampled_token_ids[0]) == 1 and run.text_queue_pos("a") == 4
    # the rebuilt inputs: embed[c0] + sum of the group embeddings (fp32, in order) -> bf16, + the text step of that step
    rebuilt = run.requests["a"].prompt_embeds[6:9]
    for j in range(3):
        acc = eng.embed[toks[j]].float()
        for g in range(1, d.num_code_groups):
            acc = acc + eng.cp_embed[g - 1][st.codes_hist[j][g]].float()
        want = (acc.to(BF16).float() + xs[j].float()).to(BF16)
        assert torch.equal(rebuilt[j], want), j
    assert int(eng.positions[0]) == state[0] + 1 and int(eng.steps[0]) == state[2] + 1


# ---------------------------------------------------------------------------------------------------------------------------
# Known answers read off the reference (its update_from_output needs vLLM and cannot be imported here):
# each case scripts the runner's output by hand and states the reference lines whose behaviour it pins.
# (tests/core/sched/test_generation_scheduler_restore.py, the only file under the reference's tests/core/sched, pins the
# try/finally queue restore of OmniGenerationScheduler's chunk adapter -- the generation stage, not this AR path: no counterpart.)
def _scripted(sched, tokens, acks=()):
    """One engine-core step with a hand-written runner output: `tokens` = {req_id: [sampled ids]} for the rows that sampled."""
    from ht_vllm_omni_amd.payloads import OmniModelRunnerOutput
    so = sched.schedule()
    ids = list(so.num_scheduled_tokens)
    out = OmniModelRunnerOutput(req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)},
                                sampled_token_ids=[list(tokens.get(r, [])) for r in ids], kv_extracted_req_ids=list(acks) or None)
    return so, sched.update_from_output(so, out)


def _plain_request(rid, n_prompt, *, max_tokens=8, stop=()):
    return Request(request_id=rid, num_prompt_tokens=n_prompt, prompt_token_ids=list(range(1, n_prompt + 1)),
                   sampling_params=SamplingParams(temperature=0.0, max_tokens=max_tokens, stop_token_ids=tuple(stop)))


def test_ref_pin_prefill_finished_marks_once_with_truncated_blocks():
    """omni_ar_scheduler.py:104-112 (prefill_finished: mark with num_computed_tokens, do not stop), :97-99 (once),
    :549-588 (_mark_request_for_kv_transfer: block ids cut to ceil(seq_len / block_size), no second marking)."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_num_batched_tokens=24, max_model_len=512,
                          kv_transfer_criteria={"type": "prefill_finished"})
    s.add_request(_plain_request("a", 40))
    _scripted(s, {})                                      # chunk 1 of the prompt (24 of 40): nothing computed past the prompt yet
    assert "a" not in s.transfer_triggered_requests and not s.requests_needing_kv_transfer
    _, outs = _scripted(s, {"a": [7]})                    # chunk 2 ends the prompt and samples
    assert s.requests["a"].num_computed_tokens == 40 and not s.requests["a"].is_finished()
    assert s.requests_needing_kv_transfer == {"a": {"seq_len": 40, "block_ids": s.pool.block_ids("a")[:3]}}   # ceil(40 / 16)
    assert outs[0].kv_transfer_params is None             # a non-stop trigger hands nothing to the client (:293-294 returns False)
    first = dict(s.requests_needing_kv_transfer["a"])
    _scripted(s, {"a": [8]})                              # delivered to the runner in this step's scheduler output, not re-marked
    assert not s.requests_needing_kv_transfer and s.active_kv_transfers == {"a": first} or "a" in s.active_kv_transfers
    _scripted(s, {"a": [9]})
    assert not s.requests_needing_kv_transfer             # once semantics: later steps never mark again


def test_ref_pin_special_token_snapshot_length_excludes_tokens_after_the_sentinel():
    """omni_ar_scheduler.py:114-133: snapshot_len = num_computed_tokens - (len(new_token_ids) - (idx + 1)), first occurrence."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512,
                          kv_transfer_criteria={"type": "special_token", "token_id": 99})
    s.add_request(_plain_request("a", 20, max_tokens=16))
    _scripted(s, {"a": [5]})
    assert not s.requests_needing_kv_transfer             # no sentinel yet
    req = s.requests["a"]
    _scripted(s, {"a": [6]})
    computed_before = req.num_computed_tokens
    # a multi-token step (spec-decode shaped): sentinel second of four, and again fourth -> the FIRST one counts
    so = s.schedule()
    from ht_vllm_omni_amd.payloads import OmniModelRunnerOutput
    out = OmniModelRunnerOutput(req_ids=["a"], req_id_to_index={"a": 0}, sampled_token_ids=[[11, 99, 12, 99]])
    s.update_from_output(so, out)
    assert req.num_computed_tokens == computed_before + 1
    assert s.requests_needing_kv_transfer["a"]["seq_len"] == req.num_computed_tokens - (4 - (1 + 1))
    assert "a" in s.transfer_triggered_requests and not req.is_finished()


def test_ref_pin_finish_after_trigger_waits_for_the_ack_then_frees():
    """omni_ar_scheduler.py:497-506 (finished while its earlier-triggered transfer is still active: hold the blocks, no
    second marking, no kv_transfer_params), :455-479 (kv_extracted_req_ids ack: leave active set; if waiting, free now)."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512,
                          kv_transfer_criteria={"type": "prefill_finished"})
    s.add_request(_plain_request("a", 18, max_tokens=2))
    _scripted(s, {"a": [5]})                              # prefill done: triggered + marked
    so, outs = _scripted(s, {"a": [6]})                   # marked set handed over in this schedule(); max_tokens reached -> finished
    assert "a" in so.finished_requests_needing_kv_transfer and "a" in s.active_kv_transfers
    assert outs[0].finish_reason is not None and outs[0].kv_transfer_params is None
    assert "a" in s.waiting_for_transfer_free and "a" in s.requests and s.pool.num_free == 16 - 1 - 2   # blocks held (block 0 reserved)
    assert not s.requests_needing_kv_transfer             # not marked a second time
    _scripted(s, {}, acks=["a"])                          # the runner reports the extraction
    assert "a" not in s.active_kv_transfers and "a" not in s.waiting_for_transfer_free
    assert "a" not in s.requests and "a" not in s.transfer_triggered_requests and s.pool.num_free == 15


def test_ref_pin_finish_after_acked_trigger_frees_immediately():
    """omni_ar_scheduler.py:507-511: triggered earlier and already extracted -> the stop frees at once, nothing to send."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512,
                          kv_transfer_criteria={"type": "prefill_finished"})
    s.add_request(_plain_request("a", 18, max_tokens=3))
    _scripted(s, {"a": [5]})
    _scripted(s, {"a": [6]}, acks=["a"])                  # handed over and acked in the same step
    assert not s.active_kv_transfers and "a" in s.transfer_triggered_requests
    _, outs = _scripted(s, {"a": [7]})                    # third token = max_tokens
    assert outs[0].finish_reason is not None and outs[0].kv_transfer_params is None
    assert "a" not in s.requests and not s.waiting_for_transfer_free and s.pool.num_free == 15


def test_ref_pin_finish_without_trigger_marks_and_returns_kv_transfer_params():
    """omni_ar_scheduler.py:512-543: an untriggered finished request is marked with num_computed_tokens, waits for the ack, and
    the client gets {"past_key_values": block_ids, "kv_metadata": {"seq_len", "block_ids"}}; :397-401 a request that is waiting
    keeps its trigger bookkeeping until the ack."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512, need_send_cache=True)     # no criteria
    s.add_request(_plain_request("a", 30, max_tokens=3))
    _scripted(s, {"a": [5]}); _scripted(s, {"a": [6]})
    _, outs = _scripted(s, {"a": [7]})
    seq_len = s.requests["a"].num_computed_tokens         # 30 prompt + 2 decoded-and-computed; the last sample was never run
    assert seq_len == 32
    blocks = s.pool.block_ids("a")[:2]                    # ceil(32 / 16)
    assert outs[0].kv_transfer_params == {"past_key_values": blocks, "kv_metadata": {"seq_len": 32, "block_ids": blocks}}
    assert "a" in s.waiting_for_transfer_free and s.requests_needing_kv_transfer["a"]["seq_len"] == 32
    so, _ = _scripted(s, {})                              # delivered once
    assert so.finished_requests_needing_kv_transfer == {"a": {"seq_len": 32, "block_ids": blocks}}
    so, _ = _scripted(s, {}, acks=["a"])
    assert not so.finished_requests_needing_kv_transfer and "a" not in s.requests and s.pool.num_free == 15


def test_ref_pin_waiting_request_is_not_retriggered():
    """omni_ar_scheduler.py:93-95: a request already waiting for its transfer to finish is skipped by the trigger."""
    s = MI355XARScheduler(num_blocks=16, block_size=16, max_num_seqs=4, max_model_len=512,
                          kv_transfer_criteria={"type": "special_token", "token_id": 99})
    s.add_request(_plain_request("a", 18, max_tokens=8))
    _scripted(s, {"a": [5]})
    req = s.requests["a"]
    s.waiting_for_transfer_free.add("a")
    assert s._process_kv_transfer_trigger(req, [99]) is False and not s.requests_needing_kv_transfer
    s.waiting_for_transfer_free.discard("a")
    assert s._process_kv_transfer_trigger(req, [99]) is False and "a" in s.requests_needing_kv_transfer   # never stops (:133)


def test_ref_pin_mtp_rows_land_on_their_requests():
    """tests/worker/test_omni_gpu_model_runner.py:119-152 (reference): talker_mtp output row b belongs to request b of the
    batch -- embeds back at the request's first flattened position, codes row b into that request's info under
    `code_predictor_codes`.  Here the batch is row-major on the device; the same statement is that decode row r's codes reach
    request r's pooler payload and history, whatever the admission order and however rows are recycled."""
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    runner = MI355XARModelRunner(eng, use_graphs=False)
    sched = MI355XARScheduler(num_blocks=32, block_size=16, max_num_seqs=4, max_model_len=512)
    core = TalkerStageEngine(_Worker(runner), sched)
    for i, n in enumerate((9, 14, 11)):
        core.add_request(_request(d, f"r{i}", n, max_tokens=5, seed=i))
    rows_seen, last_tok = {}, {}
    for _ in range(12):
        for o in core.step():
            if o.pooling_output is not None and o.new_token_ids and o.pooling_output["audio_codes"].shape[0] == 1:
                st = runner.requests.get(o.request_id)
                frame = o.pooling_output["audio_codes"][0].tolist()
                if st is not None and o.request_id in runner.rows and frame[0] != 0:
                    row = runner.rows.index(o.request_id)
                    assert frame == eng.audio_codes[row].tolist() == st.codes_hist[-1]       # row r -> request r, and its history
                    assert frame[0] == last_tok[o.request_id]      # layer-0 code = the token this step consumed (the previous sample)
                    rows_seen.setdefault(o.request_id, set()).add(row)
            if o.new_token_ids:
                last_tok[o.request_id] = o.new_token_ids[-1]
    assert len(rows_seen) == 3 and len({next(iter(v)) for v in rows_seen.values()}) == 3          # three requests, three rows


def test_cached_resume_route_with_a_batch_mate_keeps_the_text_table_in_step():
    """b is preempted while a keeps decoding, then comes back through scheduled_cached_reqs
    [resumed_from_preemption] -- a row is appended without any new / finished request in the step.  The per-step text rows are
    one gather from a table built for the batch as it WAS: the resume must rebuild it (runner._resume -> _tt_flush), or b's
    text_step row is stale and its queue position stops advancing."""
    from ht_vllm_omni_amd.payloads import OmniCachedRequestData, OmniNewRequestData, OmniSchedulerOutput
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, use_graphs=False)
    reqs = {rid: _request(d, rid, n, tail=6, seed=s) for rid, n, s in (("a", 5, 1), ("b", 7, 2))}
    new = [OmniNewRequestData(req_id=rid, prompt_token_ids=r.prompt_token_ids, block_ids=([1 + 2 * i, 2 + 2 * i],), sampling_params=r.sampling_params,
                              additional_information=r.additional_information) for i, (rid, r) in enumerate(reqs.items())]
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens={"a": 5, "b": 7}, total_num_scheduled_tokens=12))
    run.sample_tokens(None)

    def step(ids):
        run.execute_model(OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=ids, new_block_ids=[None] * len(ids)),
                                              num_scheduled_tokens={i: 1 for i in ids}, total_num_scheduled_tokens=len(ids)))
        rows = {rid: eng.text_step[run.rows.index(rid)].clone() for rid in ids}
        run.sample_tokens(None)
        return rows
    step(["a", "b"]); step(["a", "b"])
    assert run.text_queue_pos("a") == 2 and run.text_queue_pos("b") == 2
    run.execute_model(OmniSchedulerOutput(preempted_req_ids={"b"}, scheduled_cached_reqs=OmniCachedRequestData(req_ids=["a"], new_block_ids=[None]),
                                          num_scheduled_tokens={"a": 1}, total_num_scheduled_tokens=1))
    run.sample_tokens(None)
    assert run.rows == ["a"] and run.text_queue_pos("a") == 3
    # b returns by the cached route with fresh blocks: prompt + its 2 decode inputs recomputed, then it decodes in the same step
    so = OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["a", "b"], resumed_from_preemption=[False, True],
                                                                         new_block_ids=[None, ([7, 8],)], num_computed_tokens=[0, 0]),
                             num_scheduled_tokens={"a": 1, "b": 7 + 2 + 1}, total_num_scheduled_tokens=11)
    run.execute_model(so)
    tb = reqs["b"].additional_information
    from ht_vllm_omni_amd.payloads import decode_additional_information
    tail_b = decode_additional_information(tb)["tailing_text_hidden"].to(BF16)
    assert sorted(run.rows) == ["a", "b"]
    assert torch.equal(eng.text_step[run.rows.index("b")].cpu(), tail_b[2]), "the resumed row must read ITS queue entry of this step"
    run.sample_tokens(None)
    assert run.text_queue_pos("b") == 3 and run.text_queue_pos("a") == 4
    rows = step(["a", "b"])
    assert torch.equal(rows["b"].cpu(), tail_b[3]) and run.text_queue_pos("b") == 4
