"""Async scheduling of the talker stage (stage_configs/qwen3_tts.yaml:16 `async_scheduling: true`; the reference's
`AsyncGPUModelRunnerOutput` hand-over, gpu_ar_model_runner.py:641-660, and vLLM's batch queue of depth 2 around it): step t + 1 is
scheduled and dispatched BEFORE step t's outputs are read.  Host logic on the CPU stand-in engine: the request streams, code
frames, KV hand-off lengths and stop behaviour are those of the synchronous loop; a stop is seen one step late and the surplus
frame dropped; a length cap is not overrun; a chain time-out found in get_output() redoes that step AND the one dispatched on
its garbage."""
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.connectors import InProcConnector, OmniKVTransferManager
from ht_vllm_omni_amd.payloads import SamplingParams, encode_tensor
from ht_vllm_omni_amd.runner import AsyncStepOutput, MI355XARModelRunner
from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, TalkerStageEngine
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
        self.handles = []

    def execute_model(self, so):
        return self.model_runner.execute_model(so)

    def sample_tokens(self, g):
        out = self.model_runner.sample_tokens(g)
        self.handles.append(out)
        return out


def _run(async_on, *, num_blocks=64, n_req=6, max_tokens=14, stop=(), fault_at=(), kv=False, max_batch=4, late=()):
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=max_batch, num_blocks=num_blocks)
    conn = InProcConnector()
    run = MI355XARModelRunner(eng, use_graphs=False, async_scheduling=async_on, kv_transfer=OmniKVTransferManager(conn) if kv else None)
    s = MI355XARScheduler(num_blocks=num_blocks, block_size=16, max_num_seqs=max_batch, max_num_batched_tokens=48, max_model_len=512,
                          async_scheduling=async_on, need_send_cache=kv)
    wk = _Worker(run)
    core = TalkerStageEngine(wk, s)
    for i in range(n_req):
        core.add_request(_request(d, f"r{i}", 9 + 5 * i, max_tokens=max_tokens, tail=3, seed=i, stop=stop))
    late = dict(late)
    streams, codes, hidden, finish, kvp = {}, {}, {}, {}, {}
    decodes = 0
    for it in range(600):
        if it in late:
            core.add_request(_request(d, late[it], 7, max_tokens=max_tokens, tail=2, seed=77, stop=stop))
        if not core.has_work():
            break
        n0 = sum(1 for c in eng.calls if c[0] == "decode")
        if n0 in fault_at and eng.persistent_chains:
            eng.fault_next = True
        for o in core.step():
            streams.setdefault(o.request_id, []).extend(o.new_token_ids)
            if o.pooling_output is not None and o.new_token_ids and o.pooling_output["audio_codes"].shape[0] == 1 \
                    and int(o.pooling_output["audio_codes"].abs().sum()) > 0:
                codes.setdefault(o.request_id, []).append(o.pooling_output["audio_codes"][0].tolist())
                hidden.setdefault(o.request_id, []).append(float(o.pooling_output["hidden"].float().sum()))
            if o.finished:
                finish[o.request_id] = (o.finish_reason, o.stop_reason)
                if o.kv_transfer_params:
                    kvp[o.request_id] = o.kv_transfer_params["kv_metadata"]["seq_len"]
        decodes = max(decodes, n0)
    assert not core.has_work(), "engine loop did not drain"
    return dict(streams=streams, codes=codes, hidden=hidden, finish=finish, kv=kvp, run=run, eng=eng, sched=s, wk=wk, conn=conn)


def test_async_loop_matches_the_synchronous_loop():
    a, b = _run(False), _run(True)
    assert a["streams"] == b["streams"] and a["codes"] == b["codes"] and a["hidden"] == b["hidden"] and a["finish"] == b["finish"]
    assert all(len(v) == 14 for v in b["streams"].values())
    # the handles really were AsyncModelRunnerOutput-shaped, and a step was dispatched before its predecessor was read
    assert all(isinstance(h, AsyncStepOutput) for h in b["wk"].handles if h is not None and not hasattr(h, "req_ids"))
    assert any(isinstance(h, AsyncStepOutput) for h in b["wk"].handles)
    assert not b["run"].requests and not b["run"].preempted and not b["run"]._inflight
    assert b["sched"].pool.num_free == a["sched"].pool.num_free


def test_async_length_cap_is_not_overrun_and_a_stop_costs_one_surplus_step():
    # length cap: known ahead -> the async loop runs exactly the decode steps the synchronous one does
    a, b = _run(False, n_req=3, max_tokens=6), _run(True, n_req=3, max_tokens=6)
    na, nb = (sum(1 for c in r["eng"].calls if c[0] == "decode") for r in (a, b))
    assert a["streams"] == b["streams"] and nb <= na + 1, (na, nb)
    # stop token: find one the streams really emit, rerun with it as a stop id: same truncated streams in both modes, the surplus
    # frame (sampled behind the stop, one step late) never reaches the outputs
    tok = a["streams"]["r1"][3]
    a2, b2 = _run(False, n_req=3, max_tokens=12, stop=(tok,)), _run(True, n_req=3, max_tokens=12, stop=(tok,))
    assert a2["streams"] == b2["streams"] and a2["finish"] == b2["finish"] and a2["codes"] == b2["codes"]
    assert a2["finish"]["r1"] == ("stop", tok) and a2["streams"]["r1"][-1] == tok
    assert not b2["run"].requests and b2["sched"].pool.num_free == a2["sched"].pool.num_free


def test_async_preemption_recomputes_and_joins_the_same_streams():
    ref = _run(False, num_blocks=64)
    a, b = _run(False, num_blocks=8), _run(True, num_blocks=8)
    assert a["streams"] == ref["streams"] and b["streams"] == ref["streams"]
    assert b["codes"] == ref["codes"]
    assert not b["run"].preempted and not b["run"].requests


def test_async_kv_handoff_ships_the_same_lengths():
    a, b = _run(False, kv=True, n_req=3, max_tokens=5), _run(True, kv=True, n_req=3, max_tokens=5)
    assert a["kv"] == b["kv"] and len(a["kv"]) == 3
    assert a["streams"] == b["streams"]
    for rid in a["kv"]:
        ka, _ = a["conn"].get("0", "1", f"omni_0_to_1_kv_cache_{rid}")
        kb, _ = b["conn"].get("0", "1", f"omni_0_to_1_kv_cache_{rid}")
        assert ka["layer_blocks"]["key_cache"][0].shape == kb["layer_blocks"]["key_cache"][0].shape
    assert b["sched"].pool.num_free == a["sched"].pool.num_free and not b["sched"].waiting_for_transfer_free


@pytest.mark.parametrize("fault_at", [(3,), (1,), (6,)])
def test_async_chain_timeout_redoes_the_step_and_the_one_dispatched_behind_it(fault_at):
    """The status word of step t is read in get_output(t), when step t + 1 has already been dispatched on step t's garbage: both
    are redone (t from the host records of t - 1, t + 1 from the records the redone t just wrote), the streams are those of a
    clean run -- with late arrivals and finishing requests changing the batch between the two steps."""
    late = ((4, "late0"), (7, "late1"))
    clean = _run(True, n_req=3, max_tokens=9, late=late)
    hurt = _run(True, n_req=3, max_tokens=9, late=late, fault_at=fault_at)
    assert hurt["run"].chain_fallbacks == 1 and clean["run"].chain_fallbacks == 0
    assert ("recover",) in hurt["eng"].calls and not hurt["eng"].persistent_chains
    assert clean["streams"] == hurt["streams"] and clean["codes"] == hurt["codes"] and clean["hidden"] == hurt["hidden"]
    sync = _run(False, n_req=3, max_tokens=9, late=late, fault_at=fault_at)
    assert sync["streams"] == clean["streams"] and sync["run"].chain_fallbacks == 1


def test_get_output_is_idempotent_and_ordered():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, use_graphs=False, async_scheduling=True)
    s = MI355XARScheduler(num_blocks=64, block_size=16, max_num_seqs=4, max_num_batched_tokens=48, max_model_len=512, async_scheduling=True)
    s.add_request(_request(d, "a", 6, max_tokens=5))
    so1 = s.schedule(); run.execute_model(so1); h1 = run.sample_tokens(None)
    so2 = s.schedule(); run.execute_model(so2); h2 = run.sample_tokens(None)
    assert isinstance(h1, AsyncStepOutput) and so2.num_scheduled_tokens == {"a": 1}       # scheduled on a placeholder
    o2 = h2.get_output()                # asking for the younger step first finishes the older one before it
    o1 = h1.get_output()
    assert h1.get_output() is o1 and len(o1.sampled_token_ids[0]) == 1 and len(o2.sampled_token_ids[0]) == 1
    assert run.requests["a"].output_ids == o1.sampled_token_ids[0] + o2.sampled_token_ids[0]
    s.update_from_output(so1, o1); s.update_from_output(so2, o2)
    assert s.requests["a"].num_output_placeholders == 0 and s.requests["a"].output_token_ids == run.requests["a"].output_ids


def test_async_abort_while_a_step_is_in_flight():
    """A request aborted between dispatch and get_output: the scheduler frees it at once, the step in flight still carries its row (its
    output is dropped on arrival: omni_ar_scheduler.py:240-245 `request is None or request.is_finished()`), the runner drops the row with the
    next scheduler output, the other requests' streams are untouched."""
    d = get_dims("tiny")

    def run(abort_at):
        eng = FakeEngine(d, max_batch=4)
        runr = MI355XARModelRunner(eng, use_graphs=False, async_scheduling=True)
        s = MI355XARScheduler(num_blocks=64, block_size=16, max_num_seqs=4, max_num_batched_tokens=48, max_model_len=512, async_scheduling=True)
        core = TalkerStageEngine(_Worker(runr), s)
        for i in range(3):
            core.add_request(_request(d, f"r{i}", 6 + i, max_tokens=10, tail=2, seed=i))
        streams = {}
        for it in range(60):
            if it == abort_at:
                assert core.inflight, "a step must be in flight when the abort arrives"
                s.abort_request("r1")
            if not core.has_work():
                break
            for o in core.step():
                streams.setdefault(o.request_id, []).extend(o.new_token_ids)
        assert not core.has_work() and not runr.requests and not runr._inflight
        return streams, s

    ref, _ = run(-1)
    got, s = run(4)
    assert got["r0"] == ref["r0"] and got["r2"] == ref["r2"]
    assert len(got["r1"]) < len(ref["r1"]) and got["r1"] == ref["r1"][:len(got["r1"])]
    assert s.pool.num_free == 63 and not s.requests


def test_async_stop_token_of_a_preempted_request_arrives_late():
    """The step that samples a request's STOP token is in flight when the scheduler preempts that request (pool dry): the token arrives
    while the request sits in the waiting queue; it is finished there (vLLM's stopped_preempted_reqs), never recomputed, and the
    synchronous loop's streams are reproduced."""
    ref = _run(False, num_blocks=8, n_req=6, max_tokens=14)
    tok = ref["streams"]["r3"][6]
    a, b = _run(False, num_blocks=8, n_req=6, max_tokens=14, stop=(tok,)), _run(True, num_blocks=8, n_req=6, max_tokens=14, stop=(tok,))
    assert a["streams"] == b["streams"] and a["finish"] == b["finish"]
    assert not b["run"].requests and not b["run"].preempted and b["sched"].pool.num_free == a["sched"].pool.num_free


def test_a_dropped_row_gives_its_placeholder_back():
    """runner._redo_step drops a decode row whose request was preempted before the redo (its token of the redone step never existed): the
    runner output then lacks a request the step had scheduled.  The scheduler must return that request's placeholder, or the preempted
    request would never be admitted again."""
    from ht_vllm_omni_amd.payloads import OmniModelRunnerOutput
    from ht_vllm_omni_amd.scheduler import RequestStatus
    d = get_dims("tiny")
    s = MI355XARScheduler(num_blocks=64, block_size=16, max_num_seqs=4, max_num_batched_tokens=48, max_model_len=512, async_scheduling=True)
    s.add_request(_request(d, "a", 6, max_tokens=8))
    so1 = s.schedule()
    s.update_from_output(so1, OmniModelRunnerOutput(req_ids=["a"], req_id_to_index={"a": 0}, sampled_token_ids=[[5]]))
    so2 = s.schedule()                                   # a decode step for a, in flight
    req = s.requests["a"]
    assert so2.num_scheduled_tokens == {"a": 1} and req.num_output_placeholders == 1
    s.running.remove(req); s.pool.free_request("a"); req.status = RequestStatus.PREEMPTED; req.num_computed_tokens = 0; s.waiting.appendleft(req)
    s.update_from_output(so2, OmniModelRunnerOutput(req_ids=[], req_id_to_index={}, sampled_token_ids=[]))      # the runner dropped the row
    assert req.num_output_placeholders == 0
    so3 = s.schedule()
    assert [r.req_id for r in so3.scheduled_new_reqs] == ["a"] and so3.num_scheduled_tokens["a"] == 7      # recomputed: prompt + its one token


def _sched_only_loop(async_on, *, prompt=20, budget=8, criteria=None, steps=12, abort_at=None, need_send_cache=False):
    """The scheduler alone under the engine loop's call order (depth-2 queue when async): every scheduled decode row samples token 7."""
    from ht_vllm_omni_amd.payloads import OmniModelRunnerOutput
    s = MI355XARScheduler(num_blocks=64, block_size=16, max_num_seqs=4, max_num_batched_tokens=budget, max_model_len=512,
                          async_scheduling=async_on, kv_transfer_criteria=criteria, need_send_cache=need_send_cache)
    s.add_request(Request(request_id="a", num_prompt_tokens=prompt, prompt_token_ids=[1] * prompt,
                          sampling_params=SamplingParams(temperature=0.0, max_tokens=6)))
    shipped, inflight = {}, []

    def runner_output(so):
        ids, toks = [], []
        for rid, n in so.num_scheduled_tokens.items():
            r = s.requests.get(rid)
            ids.append(rid)
            # the step samples iff it computes the request's last known token (prompt end or a decode row)
            samples = r is not None and so_computed[id(so)][rid] + n >= r.num_prompt_tokens
            toks.append([7] if samples else [])
        return OmniModelRunnerOutput(req_ids=ids, req_id_to_index={r: i for i, r in enumerate(ids)}, sampled_token_ids=toks,
                                     pooler_output=None, kv_extracted_req_ids=list(so.finished_requests_needing_kv_transfer))

    so_computed = {}
    for it in range(steps):
        if abort_at is not None and it == abort_at:
            shipped["seen_at_abort"] = len(s.requests["a"].output_token_ids)
            s.abort_request("a")
        before = {rid: r.num_computed_tokens for rid, r in s.requests.items()}
        so = s.schedule()
        so_computed[id(so)] = before
        for rid, meta in so.finished_requests_needing_kv_transfer.items():
            shipped.setdefault(rid, []).append((it, meta["seq_len"]))
        if so.total_num_scheduled_tokens or so.finished_requests_needing_kv_transfer:
            inflight.append(so)
        if async_on and len(inflight) < 2 and so.total_num_scheduled_tokens:
            continue
        if inflight:
            so0 = inflight.pop(0)
            s.update_from_output(so0, runner_output(so0))
    return shipped, s


def test_async_prefill_finished_handoff_of_a_chunked_prompt_ships_the_whole_prompt():
    """with async scheduling and a prompt chunked by the token budget (P = 20, budget 8), the `prefill_finished` KV hand-off must
    fire when the final chunk has SETTLED and ship seq_len = P, as the synchronous loop does -- not one step early with P - 1."""
    crit = {"type": "prefill_finished"}
    a, _ = _sched_only_loop(False, criteria=crit)
    b, _ = _sched_only_loop(True, criteria=crit)
    assert [n for _, n in a["a"]] == [20], a
    assert [n for _, n in b["a"]] == [20], b


def test_async_abort_ships_no_positions_of_dropped_tokens():
    """an aborted request that must ship its KV (need_send_cache) ships the tokens that SETTLED -- the placeholders of the steps in
    flight are taken back from num_computed_tokens as on the stop path."""
    a, sa = _sched_only_loop(False, prompt=9, budget=48, need_send_cache=True, abort_at=4, steps=8)
    b, sb = _sched_only_loop(True, prompt=9, budget=48, need_send_cache=True, abort_at=4, steps=8)
    # a decode step computes one position and samples one token: settled positions = P + tokens seen - 1, in both modes (before the fix
    # the async scheduler shipped up to two positions more: those of the dropped tokens)
    for r in (a, b):
        assert r["seen_at_abort"] >= 1 and [n for _, n in r["a"]] == [9 + r["seen_at_abort"] - 1], r
    assert b["seen_at_abort"] <= a["seen_at_abort"]
