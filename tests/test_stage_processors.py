"""Talker -> Code2Wav hand-off against known answers minted from the reference's
stage_input_processors/qwen3_tts.py (tests/golden/make_fixtures.py cw) and the cases the reference's own
test_qwen3_tts_async_chunk.py holds.  Host logic: runs without a GPU."""
import json
import os
from collections import defaultdict
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ht_vllm_omni_amd import stage_input_processors as SP


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "chunk_windows.json")))


def _tm(chunk, left, max_num_seqs=8, store=list):
    return SimpleNamespace(code_prompt_token_ids=defaultdict(store), scheduler_max_num_seqs=max_num_seqs,
                           put_req_chunk=defaultdict(int), request_payload={},
                           connector=SimpleNamespace(config={"extra": {"codec_chunk_frames": chunk,
                                                                       "codec_left_context_frames": left}}))


def _req(rid, finished, ic=None, **extra):
    ai = None
    if ic is not None or extra:
        entries = {k: SimpleNamespace(list_data=[v]) for k, v in extra.items()}
        if ic is not None:
            entries["initial_codec_chunk_frames"] = SimpleNamespace(list_data=[ic])
        ai = SimpleNamespace(entries=entries)
    return SimpleNamespace(external_req_id=rid, is_finished=lambda: finished, additional_information=ai)


@pytest.mark.parametrize("store", ["lists", "frame-buffers"])
def test_streaming_window_sweep_matches_reference(gold, store):
    Q = gold["Q"]
    bad = 0
    for chunk, left, ic, others, n, fin, res in gold["sweep"]:
        t = _tm(chunk, left, store=list if store == "lists" else (lambda: SP.FrameBuffer(Q)))
        for o in range(others):
            t.code_prompt_token_ids[f"other-{o}"].append([1] * Q)
        for f in range(n):
            t.code_prompt_token_ids["r"].append([f % 7 + 1, 2, 3, 4])
        pl = SP.talker2code2wav_async_chunk(t, {"audio_codes": torch.zeros((0,))}, _req("r", fin, ic), is_finished=fin)
        got = None if pl is None else [pl.get("left_context_size"), len(pl["code_predictor_codes"]) // Q, bool(pl["finished"])]
        if got != res:
            bad += 1
            assert bad < 3, f"chunk={chunk} left={left} ic={ic} others={others} n={n} finished={fin}: {got} != {res}"
    assert bad == 0


def test_full_payloads_match_reference(gold):
    for case in gold["full"]:
        t = _tm(case["chunk"], case["left"], max_num_seqs=4)
        for fr in case["frames"]:
            t.code_prompt_token_ids["r"].append(fr)
        po = {"audio_codes": torch.zeros((0,))}
        if case["ref"] is not None:
            po["ref_code"] = torch.tensor(case["ref"], dtype=torch.long)
        r = _req("r", case["finished"], case["ic"], speaker=" Vivian ", language="English")
        pl = SP.talker2code2wav_async_chunk(t, po, r, is_finished=case["finished"])
        assert pl == case["payload"]


def test_ic_ladder_and_max_ic(gold):
    for a, m, mi, want in gold["ladder"]:
        assert SP.compute_dynamic_initial_chunk_size(a, m, mi) == want
    for c, want in gold["max_ic"]:
        assert SP.max_ic_for_chunk_size(c) == want


def test_window_plan_is_the_rule_the_processor_applies(gold):
    for chunk, left, ic, others, n, fin, res in gold["sweep"]:
        if ic is None or ic < 0:
            continue
        plan = SP.window_plan(n, fin, chunk, left, ic)
        if res is None or n == 0:
            assert plan is None or n == 0
        else:
            end, lc = plan
            assert [lc, end] == res[:2]


def test_eof_marker_hold_and_errors():
    t = _tm(25, 25)
    assert SP.talker2code2wav_async_chunk(t, {"audio_codes": torch.zeros((0,))}, _req("r", False)) is None
    assert SP.talker2code2wav_async_chunk(t, None, _req("r", False)) is None
    p = SP.talker2code2wav_async_chunk(t, None, _req("r", True), is_finished=True)
    assert p == {"code_predictor_codes": [], "finished": True}
    with pytest.raises(ValueError):
        SP.talker2code2wav_async_chunk(_tm(0, 25), {"audio_codes": torch.tensor([[1, 2, 3, 4]])}, _req("r", False))
    with pytest.raises(ValueError):
        SP.talker2code2wav_async_chunk(t, {"audio_codes": torch.zeros(1, 2, 3)}, _req("r", False))
    # an all-zero frame (padding / EOS step) is not a codec frame
    SP.talker2code2wav_async_chunk(t, {"audio_codes": torch.zeros(1, 4, dtype=torch.long)}, _req("z", False))
    assert len(t.code_prompt_token_ids["z"]) == 0


def test_non_streaming_processor_known_answers(gold):
    for case in gold["nonasync"]:
        out = SimpleNamespace(multimodal_output={"audio_codes": torch.tensor(case["audio_codes"]),
                                                 "ref_code": torch.tensor(case["ref_code"])},
                              token_ids=list(range(case["n_token_ids"])))
        stage = SimpleNamespace(engine_outputs=[SimpleNamespace(outputs=[out], finished=True)])
        prompts = SP.talker2code2wav(stage_list=[stage], engine_input_source=[0])
        assert len(prompts) == 1
        assert prompts[0]["additional_information"] == case["additional_information"]
        if "prompt_token_ids" in case:
            assert prompts[0]["prompt_token_ids"] == case["prompt_token_ids"]
        else:
            assert len(prompts[0]["prompt_token_ids"]) == case["prompt_token_ids_len"]
    with pytest.raises(ValueError):
        SP.talker2code2wav([], [])
    with pytest.raises(IndexError):
        SP.talker2code2wav([], [0])
    with pytest.raises(RuntimeError):
        SP.talker2code2wav([SimpleNamespace(engine_outputs=None)], [0])
    unfinished = SimpleNamespace(engine_outputs=[SimpleNamespace(outputs=[], finished=False)])
    assert SP.talker2code2wav([unfinished], [0]) == []


def test_batch_streamer_equals_per_request_calls():
    """One host copy of the step's [B, Q] codes feeding every request == the reference's per-request calls."""
    rng = np.random.default_rng(0)
    B, Q, steps = 5, 16, 70
    streamer = SP.CodecChunkStreamer(codec_chunk_frames=25, codec_left_context_frames=25, max_num_seqs=8, num_quantizers=Q)
    ref_tm = _tm(25, 25, max_num_seqs=8)
    reqs_done_at = [70, 33, 51, 12, 64]
    got, want = [], []
    for s in range(steps):
        codes = rng.integers(1, 2048, size=(B, Q))
        codes[2] = 0 if s % 9 == 4 else codes[2]                 # an all-zero row now and then: skipped, not appended
        live = [b for b in range(B) if s < reqs_done_at[b]]
        reqs = [_req(f"r{b}", s == reqs_done_at[b] - 1) for b in live]
        fin = [s == reqs_done_at[b] - 1 for b in live]
        got += streamer.on_step(reqs, codes[live], fin)
        for b, r, f in zip(live, reqs, fin):
            pl = SP.talker2code2wav_async_chunk(ref_tm, {"audio_codes": torch.from_numpy(codes[b:b + 1])}, r, is_finished=f)
            if pl is not None:
                want.append((f"r{b}", pl))
    assert got == want and len(got) > 10
    assert got[-1][1]["finished"] is True
    streamer.cleanup("r0")
    assert "r0" not in streamer.code_prompt_token_ids


@pytest.mark.parametrize("kind", ["inproc", "shm"])
def test_streamer_sends_chunks_through_the_connector(kind):
    """Talker side puts `{req}_{stage}_{chunk}` payloads, the Code2Wav side gets them back intact (SHM hop)."""
    from ht_vllm_omni_amd.connectors import OmniConnectorFactory
    conn = OmniConnectorFactory.create_connector("InProcConnector" if kind == "inproc" else "SharedMemoryConnector", {})
    st = SP.CodecChunkStreamer(codec_chunk_frames=25, codec_left_context_frames=25, max_num_seqs=1, num_quantizers=16, connector=conn)
    rng = np.random.default_rng(1)
    frames = rng.integers(1, 2048, size=(40, 16))
    keys, metas = [], {}
    orig_put = conn.put

    def put(from_stage, to_stage, put_key, data):
        r = orig_put(from_stage, to_stage, put_key, data)
        metas[put_key] = r[2]
        return r
    conn.put = put
    for s in range(40):
        keys += st.send_step([_req("rq", s == 39)], frames[s:s + 1], [s == 39], stage_id=0)
    # max_num_seqs = 1, one active request -> IC = 16: chunks at 16 frames, then 16 + 25 = 41 > 40 -> flushed at finish
    assert keys == ["rq_0_0", "rq_0_1"]
    got0, _ = conn.get("0", "1", "rq_0_0", metadata=metas["rq_0_0"])
    got1, _ = conn.get("0", "1", "rq_0_1", metadata=metas["rq_0_1"])
    assert got0["left_context_size"] == 0 and got0["finished"] is False
    assert got0["code_predictor_codes"] == frames[:16].T.reshape(-1).tolist()
    assert got1["finished"] is True and got1["left_context_size"] == 16
    assert got1["code_predictor_codes"] == frames[40 - (16 + 24):40].T.reshape(-1).tolist()
    assert "rq" not in st.code_prompt_token_ids and "rq" not in st.put_req_chunk
    conn.close()


# ------------------------------------------------------------------ Qwen3-Omni hand-offs
def _same(a, b, path=""):
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        assert isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor), path
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.cpu(), b.cpu()), path
    elif isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


@pytest.fixture(scope="module")
def omni_gold(golden_dir):
    return torch.load(os.path.join(golden_dir, "omni_stage_processors.pt"), weights_only=True)


def test_omni_talker_prompt_length(omni_gold):
    from ht_vllm_omni_amd import stage_input_processors_omni as SO
    for case in omni_gold["length"]:
        assert SO._compute_talker_prompt_ids_length(case["info"]) == case["out"]


def test_omni_thinker2talker_streaming_chunks(omni_gold):
    """Chunked thinker prefill parks the first piece and concatenates; decode chunks carry override keys."""
    from ht_vllm_omni_amd import stage_input_processors_omni as SO
    tm = SimpleNamespace(put_req_chunk=defaultdict(int), request_payload={})
    for step in omni_gold["t2t_chunks"]:
        ai = SimpleNamespace(entries={"speaker": SimpleNamespace(list_data=[" Ethan "]), "language": SimpleNamespace(list_data=["English"])})
        req = SimpleNamespace(external_req_id="rq", all_token_ids=step["prompt_ids"] + step["output_ids"],
                              prompt_token_ids=step["prompt_ids"], output_token_ids=step["output_ids"], additional_information=ai)
        out = SO.thinker2talker_async_chunk(tm, step["pooling_output"], req, is_finished=step["finished"])
        if step["out"] is None:
            assert out is None
        else:
            _same(out, step["out"], "t2t_chunk")
            tm.put_req_chunk["rq"] += 1


def test_omni_thinker2talker_and_talker2code2wav(omni_gold):
    from ht_vllm_omni_amd import stage_input_processors_omni as SO
    for case in omni_gold["t2t"]:
        stage = SimpleNamespace(engine_outputs=[SimpleNamespace(prompt_token_ids=case["prompt_ids"], outputs=[
            SimpleNamespace(multimodal_output=case["mm"], token_ids=case["output_ids"])])])
        out = SO.thinker2talker([stage], [0], prompt=case["prompt"], device="cpu")
        want = case["out"]
        assert len(out) == len(want)
        for o, w in zip(out, want):
            assert o["prompt_token_ids"] == w["prompt_token_ids"] and o["multi_modal_data"] is None
            _same(o["additional_information"], w["additional_information"], "t2t.info")
    for case in omni_gold["c2w"]:
        stage = SimpleNamespace(engine_outputs=[SimpleNamespace(outputs=[SimpleNamespace(
            multimodal_output={"code_predictor_codes": case["codes"]}, token_ids=list(range(case["n_token_ids"])))])])
        out = SO.talker2code2wav([stage], [0])
        assert [o["prompt_token_ids"] for o in out] == [w["prompt_token_ids"] for w in case["out"]]


@pytest.mark.parametrize("store", ["lists", "frame-buffers"])
def test_omni_talker2code2wav_streaming(omni_gold, store):
    from ht_vllm_omni_amd import stage_input_processors_omni as SO
    for case in omni_gold["c2w_chunks"]:
        tm = SimpleNamespace(code_prompt_token_ids=defaultdict(list if store == "lists" else (lambda: SP.FrameBuffer(16))),
                             connector=SimpleNamespace(config={"extra": {"codec_chunk_frames": case["chunk"],
                                                                         "codec_left_context_frames": case["left"]}}))
        for step in case["steps"]:
            out = SO.talker2code2wav_async_chunk(tm, {"code_predictor_codes": step["codes"]}, SimpleNamespace(external_req_id="r"),
                                                 is_finished=step["finished"])
            if step["out"] is None:
                assert out is None
            else:
                _same(out, step["out"], "c2w_chunk")
    assert SO.talker2code2wav_async_chunk(SimpleNamespace(), {}, SimpleNamespace(external_req_id="r")) is None
