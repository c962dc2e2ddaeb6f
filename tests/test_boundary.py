"""The boundary takes the reference's REAL objects (VERDICT r2 'next' item 1):
  * additional_information arrives as an AdditionalInformationPayload whose `entries` map keys to AdditionalInformationEntry
    structs (tensor_data / tensor_shape / tensor_dtype | list_data | scalar_data) -- V/engine/__init__.py:29-57, encoder
    V/engine/serialization.py:42-71, decoder :73-113, consumer V/worker/gpu_model_runner.py:915-936;
  * the worker is constructed by vLLM's executor with a VllmConfig: model_config / cache_config / parallel_config /
    scheduler_config (V/worker/gpu_ar_worker.py:22-106, V/worker/gpu_ar_model_runner.py:118-124);
  * tensor-parallel ranks agree collectively on the peer all-reduce or fall back together (V/worker/gpu_ar_worker.py:69-75).
msgspec is not installed here: the two structs are restated as plain classes with the reference's field names (NOT this
package's dataclasses), and the encoder below is the reference's: `tensor.numpy().tobytes()` + `dtype_to_name`."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.connectors import InProcConnector, OmniKVTransferManager
from ht_vllm_omni_amd.payloads import (OmniNewRequestData, OmniSchedulerOutput, SamplingParams, deserialize_additional_information,
                                       serialize_additional_information)
from ht_vllm_omni_amd.runner import MI355XARModelRunner
from tests.fakes import FakeEngine

BF16 = torch.bfloat16


class RefEntry:                      # V/engine/__init__.py:29-49 (msgspec.Struct there)
    def __init__(self, tensor_data=None, tensor_shape=None, tensor_dtype=None, list_data=None, scalar_data=None):
        self.tensor_data, self.tensor_shape, self.tensor_dtype = tensor_data, tensor_shape, tensor_dtype
        self.list_data, self.scalar_data = list_data, scalar_data


class RefPayload:                    # V/engine/__init__.py:52-57
    def __init__(self, entries):
        self.entries = entries


def ref_serialize(raw):              # V/engine/serialization.py:42-71, verbatim in behaviour
    names = {torch.float32: "float32", torch.float16: "float16", torch.int64: "int64", torch.int32: "int32", torch.uint8: "uint8",
             torch.bool: "bool", torch.float64: "float64"}
    entries = {}
    for k, v in raw.items():
        if isinstance(v, torch.Tensor):
            c = v.detach().to("cpu").contiguous()
            entries[k] = RefEntry(tensor_data=c.numpy().tobytes(), tensor_shape=list(c.shape), tensor_dtype=names[c.dtype])
        elif isinstance(v, list):
            entries[k] = RefEntry(list_data=v)
        else:
            entries[k] = RefEntry(scalar_data=v)
    return RefPayload(entries)


def test_reference_payload_objects_decode_entry_by_entry():
    g = torch.Generator().manual_seed(0)
    raw = {"thinker_prefill_embeddings": torch.randn(7, 12, generator=g), "thinker_sequences": torch.arange(9),
           "mask": torch.tensor([True, False]), "half": torch.randn(3, generator=g).to(torch.float16),
           "speaker": ["Vivian"], "chunk": 3, "none": None, "text": "hello"}
    got = deserialize_additional_information(ref_serialize(raw))
    for k in ("thinker_prefill_embeddings", "thinker_sequences", "mask", "half"):
        assert got[k].dtype == raw[k].dtype and torch.equal(got[k], raw[k]), k
    assert got["speaker"] == ["Vivian"] and got["chunk"] == 3 and got["none"] is None and got["text"] == "hello"
    # the bytes are the reference decoder's input: np.frombuffer(entry.tensor_data, np.dtype(entry.tensor_dtype)).reshape(shape)
    e = ref_serialize(raw).entries["thinker_prefill_embeddings"]
    assert np.array_equal(np.frombuffer(e.tensor_data, np.dtype(e.tensor_dtype)).reshape(e.tensor_shape), raw["thinker_prefill_embeddings"].numpy())
    # and this package's encoder emits what the reference's decoder reads (same field names, same byte order), bf16 included
    mine = serialize_additional_information({"a": raw["thinker_prefill_embeddings"], "b": torch.ones(2, 2).to(BF16), "l": [1], "s": 2.5})
    ea = mine.entries["a"]
    assert (ea.tensor_shape, ea.tensor_dtype) == ([7, 12], "float32") and ea.tensor_data == e.tensor_data
    back = deserialize_additional_information(mine)
    assert torch.equal(back["b"], torch.ones(2, 2).to(BF16)) and back["l"] == [1] and back["s"] == 2.5
    assert deserialize_additional_information(None) == {} and deserialize_additional_information({"x": 1}) == {"x": 1}
    with pytest.raises(TypeError):
        deserialize_additional_information(RefPayload(entries=None))


def test_runner_admits_a_request_whose_additional_information_is_the_reference_payload():
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, kv_transfer=OmniKVTransferManager(InProcConnector()), use_graphs=False)
    g = torch.Generator().manual_seed(3)
    pe, tail = torch.randn(5, d.hidden, generator=g), torch.randn(2, d.hidden, generator=g)
    payload = ref_serialize({"talker_prompt_embeds": pe, "tailing_text_hidden": tail, "tts_pad_embed": torch.zeros(d.hidden)})
    nr = OmniNewRequestData(req_id="a", prompt_token_ids=[d.codec_pad_id] * 5, block_ids=([1],),
                            sampling_params=SamplingParams(temperature=0.0), additional_information=payload)
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[nr], num_scheduled_tokens={"a": 5}, total_num_scheduled_tokens=5))
    out = run.sample_tokens(None)
    assert out.req_ids == ["a"] and len(out.sampled_token_ids[0]) == 1
    st = run.requests["a"]
    assert torch.equal(st.prompt_embeds, pe.to(BF16)) and torch.equal(st.tail.cpu(), tail.to(BF16))


def test_a_bad_sampling_request_is_refused_before_any_runner_state_changes():
    """SamplingParams are validated before the row is admitted; top_p < 1 with top_k disabled is served (top_k 1024)."""
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, kv_transfer=OmniKVTransferManager(InProcConnector()), use_graphs=False)
    info = {"talker_prompt_embeds": torch.zeros(3, d.hidden), "tts_pad_embed": torch.zeros(d.hidden)}
    bad = OmniNewRequestData(req_id="bad", prompt_token_ids=[0] * 3, block_ids=([1],), additional_information=info,
                             sampling_params=SamplingParams(temperature=0.7, repetition_penalty=0.0))
    with pytest.raises(ValueError, match="repetition_penalty"):
        run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[bad], num_scheduled_tokens={"bad": 3}, total_num_scheduled_tokens=3))
    assert run.rows == [] and run.requests == {}
    ok = OmniNewRequestData(req_id="p", prompt_token_ids=[0] * 3, block_ids=([1],), additional_information=info,
                            sampling_params=SamplingParams(temperature=0.7, top_k=0, top_p=0.8))
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[ok], num_scheduled_tokens={"p": 3}, total_num_scheduled_tokens=3))
    run.sample_tokens(None)
    assert int(eng.row_top_k[0]) == 1024 and abs(float(eng.row_top_p[0]) - 0.8) < 1e-6


def _vllm_config(model, **kw):
    """The nested shape vLLM hands a worker (gpu_ar_worker.py reads model_config / cache_config / parallel_config;
    gpu_ar_model_runner.py:118-124 cache_config.block_size / cache_dtype; chunk_size_utils.py max_num_seqs)."""
    return SimpleNamespace(
        model_config=SimpleNamespace(model=model, seed=7, enforce_eager=kw.get("enforce_eager", False), max_model_len=4096, dtype=torch.bfloat16,
                                     hf_config=kw.get("hf_config")),
        cache_config=SimpleNamespace(cache_dtype=kw.get("cache_dtype", "fp8"), block_size=16, gpu_memory_utilization=0.85,
                                     num_gpu_blocks_override=kw.get("blocks")),
        parallel_config=SimpleNamespace(tensor_parallel_size=kw.get("tp", 1), pipeline_parallel_size=1),
        scheduler_config=SimpleNamespace(max_num_seqs=kw.get("max_num_seqs", 64)),
        additional_config=kw.get("additional_config"))


def test_worker_is_constructed_from_a_vllm_config_shaped_object(tmp_path):
    from ht_vllm_omni_amd.checkpoint import hf_config_from_dims
    from ht_vllm_omni_amd.worker import MI355XARWorker, config_from_vllm, make_config
    import json
    w = MI355XARWorker(_vllm_config("Qwen/Qwen3-TTS-1.7B-Base", tp=2, blocks=128, cache_dtype="fp8_e4m3"), local_rank=1, rank=1,
                       distributed_init_method="tcp://127.0.0.1:1")
    assert w.dims.name == "tts-1.7b" and w.tp_size == 2 and w.rank == 1
    c = w.vllm_config
    assert (c.kv_cache_dtype, c.block_size, c.max_num_seqs, c.num_gpu_blocks_override, c.gpu_memory_utilization, c.seed) == \
        ("fp8_e4m3", 16, 64, 128, 0.85, 7)
    assert w.kv_bytes_per_block() == 28 * 2 * 16 * 4 * 128          # 8 kv heads / TP 2, fp8
    # a checkpoint directory decides the dimensions (config.json), "auto" / torch dtype names map to the bf16 cache
    d06 = get_dims("tts-0.6b")
    json.dump(hf_config_from_dims(d06), open(tmp_path / "config.json", "w"))
    w2 = MI355XARWorker(_vllm_config(str(tmp_path), cache_dtype="auto", max_num_seqs=16))
    assert (w2.dims.hidden, w2.dims.inter, w2.dims.cp_hidden) == (d06.hidden, d06.inter, d06.cp_hidden)
    assert w2.vllm_config.kv_cache_dtype == "auto" and w2.vllm_config.model_path == str(tmp_path) and w2.vllm_config.max_num_seqs == 16
    assert config_from_vllm(_vllm_config("tts-0.6b", cache_dtype="bfloat16")).kv_cache_dtype == "bf16"
    # an hf_config object carrying talker_config works too; the flat make_config namespace passes through unchanged
    hf = SimpleNamespace(to_dict=lambda: hf_config_from_dims(get_dims("tts-1.7b")))
    assert MI355XARWorker(_vllm_config("some/opaque-id", hf_config=hf)).dims.hidden == 2048
    flat = make_config("tiny")
    assert config_from_vllm(flat) is flat
    with pytest.raises(ValueError, match="cache_dtype"):
        config_from_vllm(_vllm_config("tts-1.7b", cache_dtype="fp8_e5m2"))
    with pytest.raises(ValueError, match="dimensions"):
        config_from_vllm(_vllm_config("some/unknown-model"))
    # stage-YAML extras ride in additional_config
    w3 = MI355XARWorker(_vllm_config("tts-1.7b", additional_config={"connector": "inproc", "default_sampling_params": SamplingParams(top_k=20)}))
    assert w3.vllm_config.connector == "inproc" and w3.vllm_config.default_sampling_params.top_k == 20


def test_kv_cache_spec_platform_methods_and_stage_yaml():
    """VERDICT r3 missing #4: what the executor / stage initialiser call beyond the step itself.
      * MI355XARWorker.get_kv_cache_spec(): one spec per attention layer with vLLM's FullAttentionSpec fields and page size; the
        KVCacheConfig-shaped answer (num_blocks, or per-layer kv_cache_tensors with sizes) decides the block count
        (V/worker/base.py:78-156, gpu_ar_model_runner.py:118-124);
      * the OmniPlatform methods of V/platforms/interface.py:32-35,53,113,129 the stage initialiser touches;
      * the stage YAML behind get_default_stage_config_path: stage 0 = this worker, same pipeline as the reference's file."""
    import yaml
    from ht_vllm_omni_amd.platform import MI355XOmniPlatform
    from ht_vllm_omni_amd.worker import AttentionSpec, MI355XARWorker
    w = MI355XARWorker(_vllm_config("Qwen/Qwen3-TTS-1.7B-Base", tp=2, blocks=None, cache_dtype="fp8"), local_rank=0, rank=0)
    spec = w.get_kv_cache_spec()
    assert len(spec) == 28 and all(isinstance(v, AttentionSpec) for v in spec.values())
    s0 = spec["model.layers.0.self_attn.attn"]
    assert (s0.block_size, s0.num_kv_heads, s0.head_size, s0.dtype) == (16, 4, 128, torch.uint8)
    assert s0.page_size_bytes == 2 * 16 * 4 * 128 and w.kv_bytes_per_block() == 28 * s0.page_size_bytes
    wi = MI355XARWorker(_vllm_config("tts-0.6b", cache_dtype="int8", blocks=None))
    si = next(iter(wi.get_kv_cache_spec().values()))
    assert si.page_size_bytes == 16 * (2 * 8 * 128 + 2 * 8 * 4)           # + the per-(token, head) fp32 scales
    wb = MI355XARWorker(_vllm_config("tts-0.6b", cache_dtype="auto", blocks=None))
    assert next(iter(wb.get_kv_cache_spec().values())).dtype == torch.bfloat16

    # the block count a KVCacheConfig implies: per-layer tensors, the smallest one bounds it
    class _Rec(Exception):
        pass
    got = {}
    import ht_vllm_omni_amd.engine as E
    def fake_engine(dims, weights, **kw):
        got.update(kw)
        raise _Rec()
    real = E.TalkerEngine
    E.TalkerEngine = fake_engine
    try:
        w = MI355XARWorker(_vllm_config("Qwen/Qwen3-TTS-1.7B-Base", blocks=None, cache_dtype="fp8"))      # single rank: no process group here
        w.device = torch.device("cpu")
        w._weights = {}
        page = next(iter(w.get_kv_cache_spec().values())).page_size_bytes
        tensors = [SimpleNamespace(size=(100 + l) * page, shared_by=[f"model.layers.{l}.self_attn.attn"]) for l in range(28)]
        with pytest.raises(_Rec):
            w.initialize_from_config(SimpleNamespace(num_blocks=None, kv_cache_tensors=tensors))
        assert got["num_blocks"] == 100 and got["kv_dtype"] == "fp8" and got["calculate_kv_scales"] is False
        with pytest.raises(_Rec):
            w.initialize_from_config(SimpleNamespace(num_blocks=77, kv_cache_tensors=tensors))
        assert got["num_blocks"] == 77
    finally:
        E.TalkerEngine = real
    # cache_config.calculate_kv_scales reaches the engine
    cfg = _vllm_config("tts-1.7b", cache_dtype="fp8")
    cfg.cache_config.calculate_kv_scales = True
    assert MI355XARWorker(cfg).vllm_config.calculate_kv_scales is True

    p = MI355XOmniPlatform()
    assert p.is_rocm() and not (p.is_cuda() or p.is_npu() or p.is_xpu())
    assert MI355XOmniPlatform.supports_cpu_offload() is False and MI355XOmniPlatform.supports_torch_inductor() is False
    assert MI355XOmniPlatform.get_profiler_cls().endswith("OmniTorchProfilerWrapper")
    path = MI355XOmniPlatform.get_default_stage_config_path()
    y = yaml.safe_load(open(os.path.join(path, "qwen3_tts.yaml")))
    st0, st1 = y["stage_args"]
    assert st0["engine_args"]["worker_type"] == "ar" and st0["engine_args"]["worker_cls"] == MI355XOmniPlatform.get_omni_ar_worker_cls()
    assert st0["engine_args"]["kv_cache_dtype"] == "fp8" and st0["engine_args"]["calculate_kv_scales"] is True
    assert st0["engine_args"]["max_num_seqs"] == 64 and st0["default_sampling_params"]["stop_token_ids"] == [2150]
    assert st1["engine_args"]["worker_type"] == "generation" and st0["runtime"]["devices"] == st1["runtime"]["devices"] == "0"
    ref = "/root/reference/vllm_omni/model_executor/stage_configs/qwen3_tts.yaml"
    if os.path.exists(ref):          # in the build container: everything but the four documented changes is the reference's file
        r = yaml.safe_load(open(ref))
        a, b = dict(st0["engine_args"]), dict(r["stage_args"][0]["engine_args"])
        for k in ("worker_cls", "kv_cache_dtype", "calculate_kv_scales"):
            a.pop(k)
        a["max_num_seqs"] = b["max_num_seqs"]
        assert a == b and y["runtime"] == r["runtime"] and y["stage_args"][1] == r["stage_args"][1]


def test_hf_config_without_head_dim_uses_hidden_over_heads():
    """the talker config class has no head_dim field; HF / vLLM Qwen3 falls back to hidden_size // heads."""
    from ht_vllm_omni_amd.checkpoint import dims_from_hf_config
    d = dims_from_hf_config({"talker_config": {"hidden_size": 2048, "num_attention_heads": 16, "code_predictor_config": {}}})
    assert d.head_dim == 128
    assert dims_from_hf_config({"talker_config": {"hidden_size": 1024, "num_attention_heads": 16, "head_dim": 128}}).head_dim == 128
    assert dims_from_hf_config({"talker_config": {"hidden_size": 1024, "num_attention_heads": 16}}).head_dim == 64


def _fallback_worker(rank, world, port, q):
    import torch.distributed as dist
    from ht_vllm_omni_amd.tp_comm import setup_peer_allreduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    logs = []
    ar = setup_peer_allreduce(256, 16, rank, world, log=logs.append)     # no GPU here: the buffers cannot be allocated on ANY rank
    q.put((rank, ar is None, any("unavailable" in m for m in logs)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ranks_agree_to_fall_back_when_the_peer_allreduce_cannot_be_set_up():
    """Every rank makes the same sequence of collective calls whatever fails locally: on this GPU-less host the allocation
    fails on both ranks, the all-reduce(MIN) agrees on the fall-back and both return None (the engine then runs RCCL)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_runner_emits_the_references_five_profiler_ranges():
    """V/worker/gpu_ar_model_runner.py:138,293,314,454,514 name their phases "gpu_model_runner: preprocess | forward |
    postprocess | sample | bookkeep" (record_function_or_nullcontext); the worker's profile() turns the same names on here
    (torch.profiler + roctx)."""
    from ht_vllm_omni_amd.runner import _Range
    d = get_dims("tiny")
    eng = FakeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, kv_transfer=OmniKVTransferManager(InProcConnector()), use_graphs=False)
    info = {"talker_prompt_embeds": torch.zeros(3, d.hidden), "tts_pad_embed": torch.zeros(d.hidden)}
    nr = OmniNewRequestData(req_id="a", prompt_token_ids=[0] * 3, block_ids=([1],), additional_information=info,
                            sampling_params=SamplingParams(temperature=0.0))
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[nr], num_scheduled_tokens={"a": 3}, total_num_scheduled_tokens=3))
    run.sample_tokens(None)
    from ht_vllm_omni_amd.payloads import OmniCachedRequestData
    keep, _Range.enabled = _Range.enabled, True
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            run.execute_model(OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["a"], new_block_ids=[None]),
                                                  num_scheduled_tokens={"a": 1}, total_num_scheduled_tokens=1))
            run.sample_tokens(None)
    finally:
        _Range.enabled = keep
    names = {e.name for e in prof.events()}
    for phase in ("preprocess", "forward", "postprocess", "sample", "bookkeep"):
        assert f"gpu_model_runner: {phase}" in names, phase


def test_update_intermediate_buffer_known_answers_of_the_reference_tests():
    """T/worker/test_omni_gpu_model_runner.py:172-227 restated on this runner: the update lands in model_intermediate_buffer AND
    in the request's additional_information_cpu; successive updates accumulate; an empty update and an unknown request id are
    no-ops; tensors (also inside lists) are detached host copies."""
    d = get_dims("tiny")
    run = MI355XARModelRunner(FakeEngine(d, max_batch=4), kv_transfer=OmniKVTransferManager(InProcConnector()), use_graphs=False)
    info = {"talker_prompt_embeds": torch.zeros(3, d.hidden), "tts_pad_embed": torch.zeros(d.hidden)}
    nr = OmniNewRequestData(req_id="r1", prompt_token_ids=[0] * 3, block_ids=([1],), additional_information=info,
                            sampling_params=SamplingParams(temperature=0.0))
    run.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[nr], num_scheduled_tokens={"r1": 3}, total_num_scheduled_tokens=3))
    run.sample_tokens(None)
    assert "talker_prompt_embeds" in run.model_intermediate_buffer["r1"]          # admission seeds the buffer (:935)
    t = torch.tensor([1.0, 2.0], requires_grad=True)
    run._update_intermediate_buffer("r1", {"my_tensor": t, "my_list": [3, 4]})
    buf = run.model_intermediate_buffer["r1"]
    assert torch.allclose(buf["my_tensor"], torch.tensor([1.0, 2.0])) and not buf["my_tensor"].requires_grad and buf["my_list"] == [3, 4]
    assert run.requests["r1"].additional_information_cpu is buf                      # backward-compatible mirror
    run._update_intermediate_buffer("r1", {"a": torch.tensor([1.0])})
    run._update_intermediate_buffer("r1", {"b": [torch.tensor([2.0]), 5]})
    assert torch.allclose(buf["a"], torch.tensor([1.0])) and torch.allclose(buf["b"][0], torch.tensor([2.0])) and buf["b"][1] == 5
    before = dict(buf)
    run._update_intermediate_buffer("r1", {})
    assert dict(run.model_intermediate_buffer["r1"]).keys() == before.keys()
    run._update_intermediate_buffer("unknown_req", {"key": torch.tensor([1.0])})
    assert "unknown_req" not in run.model_intermediate_buffer
    run.execute_model(OmniSchedulerOutput(finished_req_ids={"r1"}))                  # :259: the buffer goes with the request
    assert "r1" not in run.model_intermediate_buffer


def test_runner_carries_a_requests_mrope_ids_into_prefill_and_decode():
    """A request that arrives with M-RoPE ids whose rows differ (vLLM CachedRequestState.mrope_positions / .mrope_position_delta,
    V/worker/gpu_model_runner.py:121-180): the runner hands the engine [3, T] rotary ids for the prefill tokens -- the request's own
    ids, plain index ids for its batch mates -- and sets the row's rotary offset for the decode steps; rows are re-packed and
    freed rows reset with that offset.  An engine without an mrope_section refuses such a request."""
    import pytest
    import torch
    from ht_vllm_omni_amd.config import get_dims
    from ht_vllm_omni_amd.payloads import OmniCachedRequestData, OmniNewRequestData, OmniSchedulerOutput, SamplingParams, encode_tensor
    from ht_vllm_omni_amd.runner import MI355XARModelRunner
    from tests.fakes import FakeEngine

    class _MRopeEngine(FakeEngine):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.rope_delta = torch.zeros(self.max_batch, dtype=torch.int32)
            self.rope_seen = []

        def prefill(self, x, positions, req_of_tok, slot_mapping, block_table=None, rope_positions=None):
            self.rope_seen.append(None if rope_positions is None else rope_positions.clone())
            return super().prefill(x, positions, req_of_tok, slot_mapping, block_table)

    d = get_dims("tiny")
    sp = SamplingParams(temperature=0.0, max_tokens=8)
    g = torch.Generator().manual_seed(0)

    def req(rid, n, blocks, **extra):
        info = {"talker_prompt_embeds": encode_tensor(torch.randn(n, d.hidden, generator=g).to(torch.bfloat16)),
                "tts_pad_embed": encode_tensor(torch.zeros(d.hidden).to(torch.bfloat16))}
        info.update(extra)
        return OmniNewRequestData(req_id=rid, prompt_token_ids=[d.codec_pad_id] * n, block_ids=(blocks,), sampling_params=sp, additional_information=info)

    mp = torch.tensor([[0, 1, 2, 2, 2, 3], [0, 1, 2, 2, 3, 3], [0, 1, 2, 3, 2, 3]])
    eng = _MRopeEngine(d, max_batch=4)
    run = MI355XARModelRunner(eng, use_graphs=False)
    so = OmniSchedulerOutput(scheduled_new_reqs=[req("plain", 4, [1]), req("mm", 6, [2], mrope_positions=mp, mrope_position_delta=-2)],
                             num_scheduled_tokens={"plain": 4, "mm": 6}, total_num_scheduled_tokens=10)
    run.execute_model(so)
    run.sample_tokens(None)
    rope = eng.rope_seen[-1]
    assert rope is not None and rope.shape == (3, 10) and rope.dtype == torch.int32
    assert rope[:, :4].tolist() == [[0, 1, 2, 3]] * 3 and rope[:, 4:].tolist() == mp.tolist()
    r_mm, r_plain = run.rows.index("mm"), run.rows.index("plain")
    assert int(eng.rope_delta[r_mm]) == -2 and int(eng.rope_delta[r_plain]) == 0
    # one decode step later the offsets still sit on the rows of their requests; a finished request's row is reset
    run.execute_model(OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["plain", "mm"], new_block_ids=[None, None]),
                                          num_scheduled_tokens={"plain": 1, "mm": 1}, total_num_scheduled_tokens=2))
    run.sample_tokens(None)
    assert int(eng.rope_delta[run.rows.index("mm")]) == -2
    run.execute_model(OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=["plain"], new_block_ids=[None]),
                                          num_scheduled_tokens={"plain": 1}, total_num_scheduled_tokens=1, finished_req_ids={"mm"}))
    run.sample_tokens(None)
    assert "mm" not in run.rows and int(eng.rope_delta[run.rows.index("plain")]) == 0 and int(eng.rope_delta.abs().sum()) == 0
    # an engine whose model has no mrope_section cannot honour differing rows
    run2 = MI355XARModelRunner(FakeEngine(d, max_batch=4), use_graphs=False)
    with pytest.raises(ValueError, match="mrope"):
        run2.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[req("mm", 6, [2], mrope_positions=mp, mrope_position_delta=-2)],
                                               num_scheduled_tokens={"mm": 6}, total_num_scheduled_tokens=6))
    # ... while three identical plain rows are the same thing as no ids at all
    ok = torch.arange(6).expand(3, -1)
    run3 = MI355XARModelRunner(FakeEngine(d, max_batch=4), use_graphs=False)
    run3.execute_model(OmniSchedulerOutput(scheduled_new_reqs=[req("t", 6, [2], mrope_positions=ok, mrope_position_delta=0)],
                                           num_scheduled_tokens={"t": 6}, total_num_scheduled_tokens=6))
