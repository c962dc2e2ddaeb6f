#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ (run in the AUTHORING container only).

Needs /root/reference (read-only) and HF transformers; neither exists on the GPU box,
where only the committed .npz vectors are read.  Nothing of the reference's source is
copied: the reference modules are imported by path, fed seeded inputs, and only the
input/output tensors are saved.

  code_predictor_tiny.npz   reference Qwen3TTSTalkerCodePredictorForConditionalGenerationVLLM
                            (qwen3_tts_code_predictor_vllm.py) greedy codes + hidden states
  qwen3_backbone_tiny.npz   HF transformers Qwen3Model (bf16, CPU): prefill + decode hidden states
  kv_extract.npz            reference OmniKVTransferManager._extract_kv_cache in/out
  chunk_windows.json        reference talker2code2wav_async_chunk windowing known answers
  moe_block.npz             HF Qwen3OmniMoeTalkerTextSparseMoeBlock (bf16) in/out + weights + routing
  snake_beta.npz            reference SnakeBeta module (12 Hz tokenizer decoder) in/out
  omni_stage_processors.pt  reference qwen3_omni.py thinker->talker / talker->code2wav hand-offs in/out
  graph_decoder.json        reference CUDAGraphDecoderWrapper host logic: capture-size tables, bucket lookup, chunk windows
  qwen3_layer_real.npz      G2: HF Qwen3Model, ONE layer at the 1.7B dims, bf16 and fp32: q / k after norm + RoPE, attention out,
                            hidden, decode step at ctx 1 / 15 / 16 / 17 / 257
  kv_quant.npz              G3: cache bytes after fp8 (bit-level OCP e4m3fn encoder, independent of torch) / int8 KV writes
  tts_prompt_builder.pt     reference Qwen3TTSTalkerForConditionalGeneration._build_prompt_embeds / _generate_icl_prompt on a
                            stand-in `self` (reference ResizeMLP, table tokenizer): every task type / mode, in/out
  omni_prompt_builder.pt    reference Qwen3OmniMoeForConditionalGeneration prompt-embedding methods (models/qwen3_omni/
                            qwen3_omni.py) called on a stand-in `self` holding HF's ResizeMLP modules: in/out
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
V = os.path.join(REF, "vllm_omni")

from ht_vllm_omni_amd.config import get_dims  # noqa: E402
from ht_vllm_omni_amd.weights import make_weights  # noqa: E402


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_vllm_stubs():
    """<= 6 symbols of vllm / vllm_omni the imported reference files touch."""
    import logging

    class _L(logging.Logger):
        def warning_once(self, *a, **k):
            self.warning(*a, **k)

        def info_once(self, *a, **k):
            self.info(*a, **k)

    def init_logger(name):
        logging.setLoggerClass(_L)
        lg = logging.getLogger("ref." + name)
        logging.setLoggerClass(logging.Logger)
        return lg

    class _Ctx:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def default_weight_loader(param, w):
        param.data.copy_(w)

    _stub("vllm")
    _stub("vllm.logger", init_logger=init_logger)
    _stub("vllm.config", VllmConfig=object)
    _stub("vllm.config.vllm", set_current_vllm_config=_Ctx)
    _stub("vllm.model_executor")
    _stub("vllm.model_executor.model_loader")
    _stub("vllm.model_executor.model_loader.weight_utils", default_weight_loader=default_weight_loader)
    plat = types.SimpleNamespace(supports_torch_inductor=lambda: False)
    _stub("vllm_omni")
    _stub("vllm_omni.platforms", current_omni_platform=plat)


def load_by_path(modname: str, path: str, package: str | None = None):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def np16(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> uint16 bit pattern (npz has no bf16)."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


# --------------------------------------------------------------------------
def mint_code_predictor():
    install_vllm_stubs()
    pkg = "refq3tts"
    p = types.ModuleType(pkg)
    p.__path__ = [os.path.join(V, "model_executor/models/qwen3_tts")]
    sys.modules[pkg] = p
    cfgm = load_by_path(pkg + ".configuration_qwen3_tts", os.path.join(p.__path__[0], "configuration_qwen3_tts.py"), pkg)
    cpm = load_by_path(pkg + ".qwen3_tts_code_predictor_vllm",
                       os.path.join(p.__path__[0], "qwen3_tts_code_predictor_vllm.py"), pkg)

    d = get_dims("tiny")
    w = make_weights(d, seed=11, std=0.08, norm_noise=0.1)   # larger std: logits well separated
    cp_cfg = cfgm.Qwen3TTSTalkerCodePredictorConfig(
        vocab_size=d.codebook, hidden_size=d.cp_hidden, intermediate_size=d.cp_inter,
        num_hidden_layers=d.cp_layers, num_attention_heads=d.cp_q_heads, num_key_value_heads=d.cp_kv_heads,
        head_dim=d.cp_head_dim, rms_norm_eps=d.eps, rope_theta=d.cp_rope_theta, num_code_groups=d.num_code_groups)
    tk_cfg = cfgm.Qwen3TTSTalkerConfig(code_predictor_config=cp_cfg, vocab_size=d.vocab, hidden_size=d.hidden,
                                       num_code_groups=d.num_code_groups)
    vcfg = types.SimpleNamespace(scheduler_config=types.SimpleNamespace(max_num_seqs=8))
    m = cpm.Qwen3TTSTalkerCodePredictorForConditionalGenerationVLLM(vllm_config=vcfg, config=cp_cfg,
                                                                   talker_config=tk_cfg)
    m = m.to(torch.bfloat16).eval()
    hq, hkv, D = d.cp_q_heads, d.cp_kv_heads, d.cp_head_dim
    sd = {}
    for i in range(d.cp_layers):
        s, t = f"cp.l{i}.", f"model.layers.{i}."
        qkv = w[s + "wqkv"]
        sd[t + "self_attn.q_proj.weight"] = qkv[: hq * D]
        sd[t + "self_attn.k_proj.weight"] = qkv[hq * D: (hq + hkv) * D]
        sd[t + "self_attn.v_proj.weight"] = qkv[(hq + hkv) * D:]
        sd[t + "self_attn.o_proj.weight"] = w[s + "wo"]
        sd[t + "self_attn.q_norm.weight"] = w[s + "qnorm"]
        sd[t + "self_attn.k_norm.weight"] = w[s + "knorm"]
        sd[t + "input_layernorm.weight"] = w[s + "ln1"]
        sd[t + "post_attention_layernorm.weight"] = w[s + "ln2"]
        sd[t + "mlp.gate_proj.weight"] = w[s + "wgu"][: d.cp_inter]
        sd[t + "mlp.up_proj.weight"] = w[s + "wgu"][d.cp_inter:]
        sd[t + "mlp.down_proj.weight"] = w[s + "wdown"]
    sd["model.norm.weight"] = w["cp.norm"]
    for g in range(d.num_code_groups - 1):
        sd[f"model.codec_embedding.{g}.weight"] = w["cp.embed"][g]
        sd[f"lm_head.{g}.weight"] = w["cp.lm_head"][g]
    sd["small_to_mtp_projection.weight"] = w["cp.proj_w"]
    sd["small_to_mtp_projection.bias"] = w["cp.proj_b"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)

    g = torch.Generator().manual_seed(5)
    B = 5
    code0 = torch.randint(1, d.codebook, (B, 1), generator=g)
    e0 = w["embed"][code0.reshape(-1)].reshape(B, 1, -1)
    last_h = (torch.randn(B, 1, d.hidden, generator=g)).to(torch.bfloat16)
    codes = m(layer0_code=code0, layer0_embed=e0, last_talker_hidden=last_h, do_sample=False)
    # hidden states of the final re-prefill buffer (all positions filled except the last)
    buf = m._proj_buf[:B].clone()
    hid = m.model(buf, torch.arange(d.num_code_groups + 1)[None].expand(B, -1))
    out = dict(seed=np.int64(11), std=np.float64(0.08), norm_noise=np.float64(0.1),
               layer0_code=code0.numpy(), layer0_embed=np16(e0), last_talker_hidden=np16(last_h),
               all_codes=codes.numpy(), proj_buf=np16(buf), final_hidden=np16(hid))
    np.savez_compressed(os.path.join(HERE, "code_predictor_tiny.npz"), **out)
    print("code predictor codes:\n", codes)


# --------------------------------------------------------------------------
def mint_backbone():
    from transformers import Qwen3Config, Qwen3Model
    from transformers.cache_utils import DynamicCache

    d = get_dims("tiny")
    w = make_weights(d, seed=21, std=0.05, norm_noise=0.1)
    cfg = Qwen3Config(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.inter,
                      num_hidden_layers=d.layers, num_attention_heads=d.q_heads, num_key_value_heads=d.kv_heads,
                      head_dim=d.head_dim, rms_norm_eps=d.eps, rope_theta=d.rope_theta, max_position_embeddings=4096,
                      attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    m = Qwen3Model(cfg).to(torch.bfloat16).eval()
    hq, hkv, D = d.q_heads, d.kv_heads, d.head_dim
    sd = {"embed_tokens.weight": w["embed"], "norm.weight": w["norm"]}
    for i in range(d.layers):
        s, t = f"l{i}.", f"layers.{i}."
        qkv = w[s + "wqkv"]
        sd[t + "self_attn.q_proj.weight"] = qkv[: hq * D]
        sd[t + "self_attn.k_proj.weight"] = qkv[hq * D: (hq + hkv) * D]
        sd[t + "self_attn.v_proj.weight"] = qkv[(hq + hkv) * D:]
        sd[t + "self_attn.o_proj.weight"] = w[s + "wo"]
        sd[t + "self_attn.q_norm.weight"] = w[s + "qnorm"]
        sd[t + "self_attn.k_norm.weight"] = w[s + "knorm"]
        sd[t + "input_layernorm.weight"] = w[s + "ln1"]
        sd[t + "post_attention_layernorm.weight"] = w[s + "ln2"]
        sd[t + "mlp.gate_proj.weight"] = w[s + "wgu"][: d.inter]
        sd[t + "mlp.up_proj.weight"] = w[s + "wgu"][d.inter:]
        sd[t + "mlp.down_proj.weight"] = w[s + "wdown"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)

    g = torch.Generator().manual_seed(9)
    prompt_lens = [5, 17, 33]          # below / across / beyond one and two 16-token blocks
    n_decode = 4
    out = dict(seed=np.int64(21), std=np.float64(0.05), norm_noise=np.float64(0.1),
               prompt_lens=np.array(prompt_lens), n_decode=np.int64(n_decode))
    with torch.inference_mode():
        for r, n in enumerate(prompt_lens):
            x = torch.randn(1, n + n_decode, d.hidden, generator=g).to(torch.bfloat16)
            cache = DynamicCache(config=cfg)
            o = m(inputs_embeds=x[:, :n], past_key_values=cache, use_cache=True)
            hs = [o.last_hidden_state[0]]
            for t in range(n_decode):
                o = m(inputs_embeds=x[:, n + t: n + t + 1], past_key_values=cache, use_cache=True)
                hs.append(o.last_hidden_state[0])
            out[f"x{r}"] = np16(x[0])
            out[f"h{r}"] = np16(torch.cat(hs, 0))
    np.savez_compressed(os.path.join(HERE, "qwen3_backbone_tiny.npz"), **out)
    print("backbone fixture: hidden absmax", float(torch.cat(hs, 0).float().abs().max()))


# --------------------------------------------------------------------------
def mint_kv_extract():
    install_vllm_stubs()
    ku = load_by_path("refkv_utils", os.path.join(V, "distributed/omni_connectors/utils/kv_utils.py"))
    g = torch.Generator().manual_seed(3)
    cache = torch.randn(2, 6, 4, 2, 8, generator=g)
    out = {"cache": cache.numpy()}
    cases = [([1, 3], 6), ([0, 5, 2], 12), ([4], 3), ([1, 9, 3], 7)]
    for i, (ids, seq) in enumerate(cases):
        for layout in ("2first", "2second"):
            lk = cache if layout == "2first" else cache.transpose(0, 1).contiguous()
            kb, vb = ku.normalize_layer_kv(lk)
            # the gather itself follows kv_transfer_manager.py:267-281 (that file imports the
            # whole connector stack; the five lines are exercised here on the reference's
            # normalize_layer_kv output)
            mx = min(kb.shape[0], vb.shape[0]) - 1
            valid = [b for b in ids if 0 <= b <= mx]
            fk = kb[valid].flatten(0, 1)
            fv = vb[valid].flatten(0, 1)
            if seq < fk.shape[0]:
                fk, fv = fk[:seq], fv[:seq]
            out[f"k{i}_{layout}"] = fk.contiguous().numpy()
            out[f"v{i}_{layout}"] = fv.contiguous().numpy()
        out[f"ids{i}"] = np.array(ids)
        out[f"seq{i}"] = np.int64(seq)
    np.savez_compressed(os.path.join(HERE, "kv_extract.npz"), **out)
    print("kv extract cases:", len(cases))


# --------------------------------------------------------------------------
def mint_chunk_windows():
    """Known answers of the reference's streaming window rule (stage_input_processors/qwen3_tts.py:134-270):
    for a sweep of (codec_chunk_frames, codec_left_context_frames, per-request IC or the dynamic one under a given
    load, accumulated frames, finished) -> None or (left_context_size, frames in the window, finished), plus the
    full payload of a few cases with distinct frame contents and reference codes."""
    install_vllm_stubs()
    vo = sys.modules["vllm_omni"]
    vo.__path__ = [V]                         # sub-packages below have empty __init__ files
    import importlib
    from collections import defaultdict
    from types import SimpleNamespace
    q3 = importlib.import_module("vllm_omni.model_executor.stage_input_processors.qwen3_tts")
    cu = importlib.import_module("vllm_omni.model_executor.stage_input_processors.chunk_size_utils")

    def tm(chunk, left, max_num_seqs):
        return SimpleNamespace(code_prompt_token_ids=defaultdict(list), scheduler_max_num_seqs=max_num_seqs,
                               put_req_chunk=defaultdict(int), request_payload={},
                               connector=SimpleNamespace(config={"extra": {"codec_chunk_frames": chunk,
                                                                           "codec_left_context_frames": left}}))

    def req(rid, finished, ic):
        ai = None
        if ic is not None:
            ai = SimpleNamespace(entries={"initial_codec_chunk_frames": SimpleNamespace(list_data=[ic])})
        return SimpleNamespace(external_req_id=rid, is_finished=lambda: finished, additional_information=ai)

    Q = 4
    sweep = []
    for chunk, left in ((25, 25), (25, 10), (25, 0), (16, 25), (8, 3), (2, 1), (1, 0)):
        for ic in (None, 0, 1, 2, 3, 8, 10, 15, 16, 25, 30):
            for others in ((0,) if ic is not None else (0, 2, 7)):
                for n in list(range(0, 64)) + [100, 101]:
                    for fin in (False, True):
                        t = tm(chunk, left, 8)
                        for o in range(others):
                            t.code_prompt_token_ids[f"other-{o}"] = [[1] * Q]
                        t.code_prompt_token_ids["r"] = [[f % 7 + 1, 2, 3, 4] for f in range(n)]
                        pl = q3.talker2code2wav_async_chunk(transfer_manager=t, pooling_output={"audio_codes": torch.zeros((0,))},
                                                            request=req("r", fin, ic), is_finished=fin)
                        if pl is None:
                            res = None
                        else:
                            res = [pl.get("left_context_size"), len(pl["code_predictor_codes"]) // Q if pl["code_predictor_codes"] else 0,
                                   bool(pl["finished"])]
                        sweep.append([chunk, left, ic, others, n, fin, res])
    # full payloads: distinct frames, reference codes, speaker / language pass-through
    full = []
    for chunk, left, ic, n, fin, with_ref in ((25, 25, 10, 10, False, False), (25, 25, 10, 45, False, True), (25, 5, 8, 33, True, True),
                                              (16, 25, 8, 24, False, False)):
        t = tm(chunk, left, 4)
        frames = [[(7 * f + q) % 2048 for q in range(Q)] for f in range(n)]
        t.code_prompt_token_ids["r"] = [fr[:] for fr in frames]
        po = {"audio_codes": torch.zeros((0,))}
        ref = [[9, 9, 9, 9], [8, 8, 8, 8]]
        if with_ref:
            po["ref_code"] = torch.tensor(ref, dtype=torch.long)
        r = req("r", fin, ic)
        r.additional_information.entries["speaker"] = SimpleNamespace(list_data=[" Vivian "])
        r.additional_information.entries["language"] = SimpleNamespace(list_data=["English"])
        pl = q3.talker2code2wav_async_chunk(transfer_manager=t, pooling_output=po, request=r, is_finished=fin)
        full.append({"chunk": chunk, "left": left, "ic": ic, "frames": frames, "finished": fin, "ref": ref if with_ref else None,
                     "payload": {k: (bool(v) if k == "finished" else v) for k, v in pl.items()}})
    ladder = [[a, m, mi, cu.compute_dynamic_initial_chunk_size(a, m, mi)] for a in range(0, 12) for m in (0, 1, 4, 8) for mi in (1, 2, 4, 16, 32)]
    maxic = [[c, cu.max_ic_for_chunk_size(c)] for c in range(1, 80)]
    # the two known answers of the non-streaming processor held by the reference's own test file
    # (tests/model_executor/stage_input_processors/test_qwen3_tts_async_chunk.py:292-364): data only
    nonasync = [
        {"audio_codes": [[0, 0, 0, 0], [1, 2, 3, 4], [5, 6, 7, 8]], "ref_code": [[9, 9, 9, 9], [8, 8, 8, 8]], "n_token_ids": 3,
         "prompt_token_ids": [9, 8, 1, 5, 9, 8, 2, 6, 9, 8, 3, 7, 9, 8, 4, 8], "additional_information": {"left_context_size": [2]}},
        {"audio_codes": [[0, 0, 0, 0], [1, 2, 3, 4], [2150, 0, 0, 0], [5, 6, 7, 8]], "ref_code": [[9, 9, 9, 9]], "n_token_ids": 4,
         "prompt_token_ids_len": 12, "additional_information": {"left_context_size": [1]}},
    ]
    json.dump({"Q": Q, "sweep": sweep, "full": full, "ladder": ladder, "max_ic": maxic, "nonasync": nonasync},
              open(os.path.join(HERE, "chunk_windows.json"), "w"), separators=(",", ":"))
    print("chunk window cases:", len(sweep), "emits:", sum(1 for c in sweep if c[-1] is not None))


# --------------------------------------------------------------------------
def mint_omni_stage_processors():
    """Known answers of the Qwen3-Omni stage hand-offs (stage_input_processors/qwen3_omni.py): seeded inputs run through
    the reference functions; inputs and outputs saved with torch.save (tensors + plain containers only)."""
    install_vllm_stubs()
    vo = sys.modules["vllm_omni"]
    vo.__path__ = [V]
    _stub("vllm.inputs", TextPrompt=dict)
    _stub("vllm.platforms", current_platform=types.SimpleNamespace(device_type="cpu"))
    _stub("vllm_omni.engine", OmniEngineCoreRequest=object)

    class _Prompt(dict):
        def __init__(self, **kw):
            super().__init__(**kw)
    _stub("vllm_omni.inputs")
    _stub("vllm_omni.inputs.data", OmniTokensPrompt=_Prompt)
    import importlib
    from collections import defaultdict
    from types import SimpleNamespace
    qo = importlib.import_module("vllm_omni.model_executor.stage_input_processors.qwen3_omni")
    g = torch.Generator().manual_seed(17)
    IM, SYS, USR, AST = 151644, 8948, 872, 77091

    def turn(role, n):
        return [IM, role] + torch.randint(1000, 2000, (n,), generator=g).tolist()

    cases = {"length": [], "t2t_chunks": [], "t2t": [], "c2w_chunks": [], "c2w": []}
    for turns in ([(SYS, 5), (USR, 7), (AST, 0)], [(USR, 3), (AST, 2), (USR, 9), (AST, 0)], [(SYS, 2), (AST, 0)], [(USR, 4)]):
        ids = sum((turn(r, n) for r, n in turns), [])
        seq = ids + torch.randint(1000, 2000, (6,), generator=g).tolist()
        info = {"thinker_sequences": seq, "thinker_input_ids": ids}
        cases["length"].append({"info": info, "out": qo._compute_talker_prompt_ids_length(info, device="cpu")})

    H = 8
    def pool(n):
        return {"0": torch.randn(n, H, generator=g), "24": torch.randn(n, H, generator=g), "tts_bos_embed": torch.randn(1, H, generator=g),
                "tts_eos_embed": torch.randn(1, H, generator=g), "tts_pad_embed": torch.randn(1, H, generator=g)}

    def mkreq(prompt_ids, out_ids, speaker=None):
        ai = None
        if speaker:
            ai = SimpleNamespace(entries={"speaker": SimpleNamespace(list_data=[speaker]), "language": SimpleNamespace(list_data=["English"])})
        return SimpleNamespace(external_req_id="rq", all_token_ids=prompt_ids + out_ids, prompt_token_ids=prompt_ids,
                               output_token_ids=out_ids, additional_information=ai)

    # streaming thinker -> talker: chunked prefill (2 pieces), then decode steps, then finish
    tm = SimpleNamespace(put_req_chunk=defaultdict(int), request_payload={})
    script = [(pool(5), [1, 2, 3, 4, 5], [], False), (pool(3), [1, 2, 3, 4, 5, 6, 7, 8], [], False), (pool(1), [1, 2, 3, 4, 5, 6, 7, 8], [9], False),
              (pool(1), [1, 2, 3, 4, 5, 6, 7, 8], [9, 10], True)]
    for po, pids, oids, fin in script:
        req = mkreq(pids, oids, speaker=" Ethan ")
        out = qo.thinker2talker_async_chunk(tm, po, req, is_finished=fin)
        cases["t2t_chunks"].append({"pooling_output": po, "prompt_ids": pids, "output_ids": oids, "finished": fin, "out": out})
        if out is not None:
            tm.put_req_chunk["rq"] += 1
    # non-streaming thinker -> talker
    for turns in ([(SYS, 4), (USR, 6), (AST, 0)], [(USR, 5), (AST, 0)]):
        pids = sum((turn(r, n) for r, n in turns), [])
        oids = torch.randint(1000, 2000, (4,), generator=g).tolist()
        mm = pool(len(pids) + len(oids))
        stage = SimpleNamespace(engine_outputs=[SimpleNamespace(prompt_token_ids=pids, outputs=[SimpleNamespace(multimodal_output=mm, token_ids=oids)])])
        prm = [{"additional_information": {"speaker": ["ethan"], "language": ["English"]}}]
        out = qo.thinker2talker([stage], [0], prompt=prm)
        cases["t2t"].append({"prompt_ids": pids, "output_ids": oids, "mm": mm, "prompt": prm, "out": [dict(o) for o in out]})
    # streaming talker -> code2wav (Omni cadence)
    for chunk, left, n, fin_at in ((4, 2, 11, 10), (25, 25, 30, 29), (3, 5, 7, None)):
        tm2 = SimpleNamespace(code_prompt_token_ids=defaultdict(list),
                              connector=SimpleNamespace(config={"extra": {"codec_chunk_frames": chunk, "codec_left_context_frames": left}}))
        steps = []
        for t in range(n):
            codes = torch.randint(0, 2048, (16, 1), generator=g)
            if t == 2:
                codes = torch.zeros(16, 1, dtype=torch.long)       # an all-zero frame is skipped
            fin = fin_at is not None and t == fin_at
            out = qo.talker2code2wav_async_chunk(tm2, {"code_predictor_codes": codes}, SimpleNamespace(external_req_id="r"), is_finished=fin)
            steps.append({"codes": codes, "finished": fin, "out": out})
        cases["c2w_chunks"].append({"chunk": chunk, "left": left, "steps": steps})
    codes = torch.randint(0, 2048, (9, 16), generator=g)
    stage = SimpleNamespace(engine_outputs=[SimpleNamespace(outputs=[SimpleNamespace(multimodal_output={"code_predictor_codes": codes}, token_ids=list(range(6)))])])
    cases["c2w"].append({"codes": codes, "n_token_ids": 6, "out": [dict(o) for o in qo.talker2code2wav([stage], [0])]})
    torch.save(cases, os.path.join(HERE, "omni_stage_processors.pt"))
    print("omni stage processor cases:", {k: len(v) for k, v in cases.items()})


# --------------------------------------------------------------------------
def mint_snake_beta():
    """The reference's own SnakeBeta module (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:602-700, eager CPU form)
    on seeded inputs: x, alpha, beta and the output, fp32."""
    install_vllm_stubs()
    pkg = "reftok12"
    pk = types.ModuleType(pkg)
    pk.__path__ = [os.path.join(V, "model_executor/models/qwen3_tts/tokenizer_12hz")]
    sys.modules[pkg] = pk
    load_by_path(pkg + ".configuration_qwen3_tts_tokenizer_v2", os.path.join(pk.__path__[0], "configuration_qwen3_tts_tokenizer_v2.py"), pkg)
    mod = load_by_path(pkg + ".modeling_qwen3_tts_tokenizer_v2", os.path.join(pk.__path__[0], "modeling_qwen3_tts_tokenizer_v2.py"), pkg)
    g = torch.Generator().manual_seed(23)
    out = {}
    for i, (B, C, T) in enumerate(((1, 8, 37), (2, 48, 257), (1, 1536, 5), (3, 5, 1027))):
        m = mod.SnakeBeta(C)
        with torch.no_grad():
            m.alpha.copy_(torch.randn(C, generator=g) * 0.5)
            m.beta.copy_(torch.randn(C, generator=g) * 0.5)
        x = torch.randn(B, C, T, generator=g) * 2
        with torch.no_grad():
            y = m._eager_forward(x)
        out[f"x{i}"], out[f"alpha{i}"], out[f"beta{i}"], out[f"y{i}"] = x.numpy(), m.alpha.detach().numpy(), m.beta.detach().numpy(), y.numpy()
    out["n"] = np.int64(4)
    np.savez_compressed(os.path.join(HERE, "snake_beta.npz"), **out)
    print("snake beta cases: 4")


# --------------------------------------------------------------------------
def mint_moe_block():
    """HF transformers Qwen3OmniMoeTalkerTextSparseMoeBlock (bf16, CPU) on seeded inputs: the published algorithm of the
    Omni talker's MoE MLP (the reference itself runs vLLM's FusedMoE, absent here)."""
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeTalkerTextConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeTalkerTextSparseMoeBlock
    from tests.util import make_moe_weights
    out = {}
    for ci, (H, E, K, I, Is, T, norm) in enumerate(((64, 16, 4, 32, 48, 9, False), (128, 128, 8, 96, 64, 37, False), (64, 8, 2, 32, 32, 5, True))):
        cfg = Qwen3OmniMoeTalkerTextConfig(hidden_size=H, num_experts=E, num_experts_per_tok=K, moe_intermediate_size=I,
                                           shared_expert_intermediate_size=Is, norm_topk_prob=norm, num_hidden_layers=1,
                                           num_attention_heads=2, num_key_value_heads=1)
        m = Qwen3OmniMoeTalkerTextSparseMoeBlock(cfg).to(torch.bfloat16).eval()
        w = make_moe_weights(H, E, I, Is, seed=40 + ci)
        sd = {"gate.weight": w["router"], "experts.gate_up_proj": w["gate_up"], "experts.down_proj": w["down"],
              "shared_expert.gate_proj.weight": w["shared_gate_up"][:Is], "shared_expert.up_proj.weight": w["shared_gate_up"][Is:],
              "shared_expert.down_proj.weight": w["shared_down"], "shared_expert_gate.weight": w["shared_gate"]}
        missing, unexpected = m.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(140 + ci)
        x = torch.randn(1, T, H, generator=g).to(torch.bfloat16)
        with torch.no_grad():
            y = m(x)
            _, rw, ri = m.gate(x.reshape(-1, H))
        out[f"c{ci}_meta"] = np.array([H, E, K, I, Is, T, int(norm), 40 + ci])
        out[f"c{ci}_x"], out[f"c{ci}_y"] = np16(x[0]), np16(y.reshape(T, H))
        out[f"c{ci}_topk_idx"], out[f"c{ci}_topk_w"] = ri.numpy(), np16(rw)
    out["n"] = np.int64(3)
    np.savez_compressed(os.path.join(HERE, "moe_block.npz"), **out)
    print("moe block cases: 3")


def mint_graph_decoder():
    """Known answers of the reference's CUDAGraphDecoderWrapper host logic (cuda_graph_decoder_wrapper.py): capture-size
    tables for a sweep of chunking configs, bucket lookup, and the chunk / left-context boundaries of the chunked decode
    (recorded by running the reference wrapper, disabled = eager, on a decoder that returns its input length)."""
    install_vllm_stubs()
    mod = load_by_path("ref_cuda_graph_decoder_wrapper", os.path.join(V, "model_executor/models/qwen3_tts/cuda_graph_decoder_wrapper.py"))
    W = mod.CUDAGraphDecoderWrapper
    sizes = []
    for ccf in (0, 1, 12, 25, 33, 50, 300, 400):
        for lcf in (0, 25, 72):
            for dcs, dlc in ((300, 25), (100, 10), (16, 0), (1000, 50)):
                kw = dict(codec_chunk_frames=ccf, codec_left_context_frames=lcf, decode_chunk_size=dcs, decode_left_context=dlc)
                sizes.append({"kw": kw, "out": W.compute_capture_sizes(**kw)})
    w = W(decoder=None, capture_sizes=[100, 25, 50])
    lookup = [{"n": n, "out": w._get_padded_size(n)} for n in (0, 1, 24, 25, 26, 50, 51, 100, 101, 5000)]

    class Probe(torch.nn.Module):
        total_upsample = 3

        def __init__(self):
            super().__init__()
            self.calls = []

        def forward(self, codes):
            self.calls.append((int(codes[0, 0, 0]), int(codes.shape[-1])))       # first frame index, window length
            return codes[:, :1, :].float().repeat_interleave(3, dim=-1)

    chunked = []
    for total, cs, lc in ((1, 300, 25), (24, 300, 25), (300, 300, 25), (301, 300, 25), (650, 300, 25), (100, 30, 10), (100, 30, 0),
                          (61, 30, 30), (90, 30, 45)):
        pr = Probe()
        wr = W(decoder=pr, capture_sizes=[8], enabled=False)
        codes = torch.arange(total).reshape(1, 1, total).expand(1, 2, total)
        out = wr.chunked_decode_with_cudagraph(codes, chunk_size=cs, left_context_size=lc)
        chunked.append({"total": total, "chunk_size": cs, "left_context_size": lc, "windows": pr.calls, "out": out[0, 0].long().tolist()})
    path = os.path.join(HERE, "graph_decoder.json")
    with open(path, "w") as f:
        json.dump({"capture_sizes": sizes, "lookup": lookup, "chunked": chunked}, f, separators=(",", ":"))
    print("graph_decoder.json", os.path.getsize(path), "bytes;", len(sizes), "size tables,", len(chunked), "chunked decodes")


def mint_mrope_positions():
    """M-RoPE position ids with DIFFERING rows from the reference's own OmniMRotaryEmbedding.get_input_positions_tensor
    (V/model_executor/layers/rotary_embedding/mrope.py:64-109 -> the Omni arm 311-478): text, audio, image, video, and audio
    interleaved with video, plus context_len / seq_len slicing.  Inputs and outputs only -> tests/golden/mrope_positions.json."""
    import json
    install_vllm_stubs()
    _stub("vllm.model_executor.layers")
    _stub("vllm.model_executor.layers.rotary_embedding")
    _stub("vllm.model_executor.layers.rotary_embedding.mrope", MRotaryEmbedding=type("MRotaryEmbedding", (), {}))
    _stub("vllm.transformers_utils")
    _stub("vllm.transformers_utils.config", thinker_uses_mrope=lambda cfg: hasattr(cfg, "thinker_config"))
    mod = load_by_path("ref_mrope", os.path.join(REF, "vllm_omni/model_executor/layers/rotary_embedding/mrope.py"))
    R = mod.OmniMRotaryEmbedding
    ids = dict(audio_token_index=901, image_token_index=902, video_token_index=903, audio_start_token_id=904, audio_end_token_id=905,
               vision_start_token_id=906, vision_end_token_id=907, seconds_per_chunk=2)
    vis = dict(spatial_merge_size=2, tokens_per_second=25)
    cfg = types.SimpleNamespace(thinker_config=types.SimpleNamespace(vision_config=types.SimpleNamespace(**vis), **ids))
    A, I, V, AS, AE, VS, VE = 901, 902, 903, 904, 905, 906, 907
    txt = lambda n, base=10: list(range(base, base + n))
    naud = lambda L: ((L - 1) // 2 + 1 - 2) // 2 + 1
    cases = []

    def case(name, tokens, **kw):
        call = dict(image_grid_thw=[], video_grid_thw=[], second_per_grid_ts=[], audio_feature_lengths=None, use_audio_in_video=False,
                    context_len=0, seq_len=None)
        call.update(kw)
        afl = call["audio_feature_lengths"]
        pos, delta = R.get_input_positions_tensor(tokens, cfg, call["image_grid_thw"], call["video_grid_thw"], call["second_per_grid_ts"],
                                                  context_len=call["context_len"], seq_len=call["seq_len"],
                                                  audio_feature_lengths=None if afl is None else torch.tensor(afl),
                                                  use_audio_in_video=call["use_audio_in_video"])
        cases.append(dict(name=name, tokens=tokens, positions=pos.tolist(), delta=int(delta), **call))

    case("text_only", txt(20))
    case("text_only_sliced", txt(20), context_len=5, seq_len=17)
    case("audio", txt(3) + [AS] + [A] * naud(101) + [AE] + txt(4), audio_feature_lengths=[101])
    case("image", txt(2) + [VS] + [I] * 24 + [VE] + txt(5), image_grid_thw=[[1, 8, 12]])
    case("two_images_and_audio", txt(2) + [VS] + [I] * 6 + [VE] + txt(1) + [VS] + [I] * 16 + [VE] + [AS] + [A] * naud(57) + [AE] + txt(3),
         image_grid_thw=[[1, 4, 6], [1, 8, 8]], audio_feature_lengths=[57])
    case("video", txt(2) + [VS] + [V] * 16 + [VE] + txt(3), video_grid_thw=[[4, 4, 4]], second_per_grid_ts=[0.5])
    case("video_sliced", txt(2) + [VS] + [V] * 16 + [VE] + txt(3), video_grid_thw=[[4, 4, 4]], second_per_grid_ts=[0.5], context_len=4, seq_len=20)
    for L, tag in ((401, "audio_shorter"), (801, "audio_longer")):
        n = naud(L)
        case("audio_in_video_" + tag, txt(2) + [VS, AS] + [V] * (24 + n) + [AE, VE] + txt(3), video_grid_thw=[[6, 4, 4]], second_per_grid_ts=[1.0],
             audio_feature_lengths=[L], use_audio_in_video=True)
    out = dict(config=dict(ids, **vis), cases=cases)
    json.dump(out, open(os.path.join(HERE, "mrope_positions.json"), "w"))
    print("mrope_positions.json:", [(c["name"], len(c["positions"][0]), c["delta"]) for c in cases])


def install_auto_stubs():
    """qwen3_omni.py imports ~20 vllm / vllm_omni symbols at module level (registries, interfaces, thinker classes) that its
    prompt-embedding methods never touch: every such name resolves to an empty placeholder class so that the module
    imports; the methods under test run on torch tensors and HF modules only."""
    import importlib.abc
    import importlib.machinery

    class _Meta(type):
        def __getattr__(cls, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return lambda *a, **k: (lambda x: x)

    class _AutoMod(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            c = _Meta(name, (), {"__init__": lambda self_, *a, **k: None})      # instantiable with any arguments (class-level uses)
            setattr(self, name, c)
            return c

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path, target=None):
            if fullname.split(".")[0] in ("vllm", "vllm_omni", "flash_attn", "soundfile", "librosa") and fullname not in sys.modules:
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = _AutoMod(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass

    install_vllm_stubs()
    sys.meta_path.insert(0, _Finder())
    for n in list(sys.modules):
        if n.split(".")[0] in ("vllm", "vllm_omni") and not isinstance(sys.modules[n], _AutoMod):
            old, m = sys.modules[n], _AutoMod(n)
            m.__path__ = []
            m.__dict__.update({k: v for k, v in old.__dict__.items() if not k.startswith("__")})
            sys.modules[n] = m


def mint_omni_prompt_builder():
    """Known answers of the Omni talker's prompt-embedding builder: the reference's own methods
    (_thinker_to_talker_prefill, _get_talker_user_parts, _get_talker_assistant_parts, _get_tts_embed,
    talker_preprocess_decode, _thinker_decode_to_talker_decode; qwen3_omni.py:650-1060) bound to a stand-in `self` whose
    projections are HF Qwen3OmniMoeTalkerResizeMLP modules (bf16, CPU) with seeded weights."""
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeTalkerResizeMLP
    install_auto_stubs()
    mod = load_by_path("ref_qwen3_omni_model", os.path.join(V, "model_executor/models/qwen3_omni/qwen3_omni.py"))
    Cls = mod.Qwen3OmniMoeForConditionalGeneration
    NS = types.SimpleNamespace
    Ht, I, H, Vc = 64, 96, 32, 64
    ids = dict(im_start=5, system=6, user=7, assistant=8, audio=9, image=10, video=11, tts_pad_token=12,
               codec_nothink=40, codec_think_bos=41, codec_think_eos=42, codec_pad=43, codec_bos=44)
    g = torch.Generator().manual_seed(2026)
    mcfg = NS(thinker_hidden_size=Ht, text_config=NS(intermediate_size=I, hidden_size=H, hidden_act="silu"))

    def mlp():
        m = Qwen3OmniMoeTalkerResizeMLP(mcfg)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g) * (0.2 if p_.ndim == 2 else 0.1))
        return m.to(torch.bfloat16).eval()

    text_p, hid_p = mlp(), mlp()
    table = (torch.randn(Vc, H, generator=g) * 0.5).to(torch.bfloat16)
    tcfg = NS(codec_nothink_id=ids["codec_nothink"], codec_think_bos_id=ids["codec_think_bos"],
              codec_think_eos_id=ids["codec_think_eos"], codec_pad_id=ids["codec_pad"], codec_bos_id=ids["codec_bos"],
              text_config=NS(hidden_size=H))
    cfg = NS(im_start_token_id=ids["im_start"], system_token_id=ids["system"], user_token_id=ids["user"],
             assistant_token_id=ids["assistant"], tts_pad_token_id=ids["tts_pad_token"], talker_config=tcfg)
    fake = NS(config=cfg, talker_config=tcfg,
              thinker_config=NS(audio_token_id=ids["audio"], image_token_id=ids["image"], video_token_id=ids["video"]),
              talker=NS(text_projection=text_p, hidden_projection=hid_p, embed_input_ids=lambda t: table[t]),
              _module_device=lambda m: torch.device("cpu"),
              vllm_config=NS(model_config=NS(async_chunk=False)))
    for name in ("_thinker_to_talker_prefill", "_get_talker_user_parts", "_get_talker_assistant_parts", "_get_tts_embed",
                 "talker_preprocess_decode", "_thinker_decode_to_talker_decode"):
        setattr(fake, name, types.MethodType(getattr(Cls, name), fake))

    def w_of(m):
        return {"fc1_w": m.linear_fc1.weight.detach().clone(), "fc1_b": m.linear_fc1.bias.detach().clone(),
                "fc2_w": m.linear_fc2.weight.detach().clone(), "fc2_b": m.linear_fc2.bias.detach().clone()}

    def seg(role, n_text, n_mm=0, mm_tok=None):
        body = [20 + int(x) for x in torch.randint(0, 10, (n_text,), generator=g)]
        if n_mm:
            body = body[: n_text // 2] + [mm_tok] * n_mm + body[n_text // 2:]
        return [ids["im_start"], role] + body

    layouts = {
        "system_user_assistant": [seg(ids["system"], 4), seg(ids["user"], 6), seg(ids["assistant"], 9)],
        "user_audio_assistant_short": [seg(ids["user"], 4, 5, ids["audio"]), seg(ids["assistant"], 1)],      # 3 rows: no first text
        "history": [seg(ids["system"], 2), seg(ids["user"], 3, 2, ids["image"]), seg(ids["assistant"], 5),
                    seg(ids["user"], 4, 3, ids["video"]), seg(ids["assistant"], 2)],                          # exactly 4 rows
        "user_only_mm_assistant_5": [seg(ids["user"], 0, 7, ids["audio"]), seg(ids["assistant"], 3)],          # 5 rows: 1 trailing
        "long": [seg(ids["system"], 8), seg(ids["user"], 70, 90, ids["image"]), seg(ids["assistant"], 40)],
    }
    cases = []
    with torch.no_grad():
        for name, segs in layouts.items():
            result_ids = torch.tensor([t for s_ in segs for t in s_], dtype=torch.long)
            T = int(result_ids.shape[0])
            n_gen = len(segs[-1]) - 3          # the thinker's generated ids: everything after "<|im_start|>assistant\n"
            prompt_ids = result_ids[: T - max(n_gen, 0)]
            emb = (torch.randn(T, Ht, generator=g)).to(torch.bfloat16)
            hid = (torch.randn(T, Ht, generator=g)).to(torch.bfloat16)
            bos, eos, pad = ((torch.randn(1, 1, Ht, generator=g)).to(torch.bfloat16) for _ in range(3))   # [1, 1, Ht] as shipped
            if name == "history":
                pad = None                                           # the zero fallback of _get_tts_embed
            spk = 50 + len(cases)
            o_ids, o_emb, o_tail = fake._thinker_to_talker_prefill(
                thinker_embed=emb, thinker_hidden=hid, multimodal_mask=None, input_ids=prompt_ids.unsqueeze(0),
                thinker_result_ids=result_ids, speaker_id=spk, tts_bos_thinker=bos, tts_eos_thinker=eos, tts_pad_thinker=pad)
            c = {"name": name, "thinker_embed": emb, "thinker_hidden": hid, "input_ids": prompt_ids, "result_ids": result_ids,
                 "speaker_id": spk, "tts_bos": bos, "tts_eos": eos, "tts_pad": pad,
                 "out_ids": o_ids.clone(), "out_embeds": o_emb.clone(), "out_trailing": o_tail.clone(),
                 "tts_pad_proj": fake.tts_pad_embed.clone(), "tts_eos_proj": fake.tts_eos_embed.clone()}
            # decode side, non-streaming: pop the queue past its end (talker_preprocess_decode)
            tail, steps = o_tail, []
            dummy = torch.zeros(1, H, dtype=torch.bfloat16)
            for _ in range(int(o_tail.shape[0]) + 3):
                upd = {}
                _, text_step, upd = fake.talker_preprocess_decode(torch.zeros(1, dtype=torch.long), dummy, upd,
                                                                  trailing_text_hidden=tail, last_talker_hidden=None)
                steps.append(text_step.clone())
                tail = upd.get("trailing_text_hidden", tail)
            c["decode_text_steps"] = torch.cat(steps, 0)
            cases.append(c)
        # decode side, streaming: _thinker_decode_to_talker_decode over a scripted arrival of thinker decode embeddings
        # (the prefill seeds the cache with the embeddings that arrived so far: _talker_cache_thinker_decode_embeds)
        script, n_out = [], 7
        cached, finished = (torch.randn(3, Ht, generator=g)).to(torch.bfloat16), False
        arrivals = {1: 2, 3: 1}
        for stepno in list(range(1, 9)) + ["beyond_cache"]:
            if stepno == "beyond_cache":     # index past the cache: the reference projects the fresh rows as they are
                start, cached_in, fin_in = 2, (torch.randn(2, Ht, generator=g)).to(torch.bfloat16), False
                fresh = (torch.randn(1, Ht, generator=g)).to(torch.bfloat16)
            else:
                start, cached_in, fin_in = stepno, cached, finished
                n_new = arrivals.get(stepno, 0)
                fresh = (torch.randn(n_new, Ht, generator=g)).to(torch.bfloat16) if n_new else None
            info_d = {"num_processed_tokens": start, "thinker_output_token_ids": list(range(n_out)),
                      "cached_thinker_decode_embeddings": cached_in, "thinker_decode_embeddings": fresh, "finished_flag": fin_in}
            upd = {}
            out = fake._thinker_decode_to_talker_decode(info_d, torch.device("cpu"), upd).clone()
            script.append({"num_processed_tokens": start, "cached": cached_in.clone(), "fresh": fresh, "finished_flag": fin_in,
                           "out": out, "cached_after": upd.get("cached_thinker_decode_embeddings"),
                           "finished_after": upd.get("finished_flag")})
            if stepno != "beyond_cache":
                if upd.get("cached_thinker_decode_embeddings") is not None:
                    cached = upd["cached_thinker_decode_embeddings"]
                finished = upd.get("finished_flag", finished)
    out = {"ids": ids, "dims": {"thinker_hidden": Ht, "inter": I, "hidden": H, "codec_vocab": Vc},
           "weights": {"text": w_of(text_p), "hidden": w_of(hid_p), "codec_embed": table}, "cases": cases,
           "streaming": {"n_thinker_output_ids": n_out, "script": script}}
    path = os.path.join(HERE, "omni_prompt_builder.pt")
    torch.save(out, path)
    print("omni_prompt_builder.pt", os.path.getsize(path), "bytes;", [(c["name"], tuple(c["out_embeds"].shape), tuple(c["out_trailing"].shape)) for c in cases])
    print("streaming:", [(s_["num_processed_tokens"], s_["out"] if isinstance(s_["out"], str) else tuple(s_["out"].shape)) for s_ in script])


# --------------------------------------------------------------------------
# G2: ONE decoder layer at the real 1.7B dimensions, HF Qwen3Model in bf16 AND fp32, decode step at
# ctx in {1, 15, 16, 17, 257}: q / k after q/k-norm + RoPE, attention output (o_proj input), final-normed hidden.
G2_CTX = (1, 15, 16, 17, 257)
G2_SEED, G2_XSEED, G2_XSTD = 33, 5, 0.5


def g2_dims():
    return get_dims("tts-1.7b").with_(layers=1, cp_layers=1, num_code_groups=2, max_model_len=512)


def g2_inputs(d):
    g = torch.Generator().manual_seed(G2_XSEED)
    return (torch.randn(max(G2_CTX), d.hidden, generator=g) * G2_XSTD).to(torch.bfloat16)


def mint_backbone_layer_real():
    import transformers.models.qwen3.modeling_qwen3 as MQ
    from transformers import Qwen3Config, Qwen3Model
    from transformers.cache_utils import DynamicCache
    d = g2_dims()
    w = make_weights(d, seed=G2_SEED, std=0.02, norm_noise=0.1)
    x = g2_inputs(d)
    cfg = Qwen3Config(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.inter, num_hidden_layers=1,
                      num_attention_heads=d.q_heads, num_key_value_heads=d.kv_heads, head_dim=d.head_dim, rms_norm_eps=d.eps,
                      rope_theta=d.rope_theta, max_position_embeddings=4096, attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    hq, hkv, D = d.q_heads, d.kv_heads, d.head_dim
    qkv = w["l0.wqkv"]
    sd = {"embed_tokens.weight": w["embed"], "norm.weight": w["norm"],
          "layers.0.self_attn.q_proj.weight": qkv[: hq * D], "layers.0.self_attn.k_proj.weight": qkv[hq * D: (hq + hkv) * D],
          "layers.0.self_attn.v_proj.weight": qkv[(hq + hkv) * D:], "layers.0.self_attn.o_proj.weight": w["l0.wo"],
          "layers.0.self_attn.q_norm.weight": w["l0.qnorm"], "layers.0.self_attn.k_norm.weight": w["l0.knorm"],
          "layers.0.input_layernorm.weight": w["l0.ln1"], "layers.0.post_attention_layernorm.weight": w["l0.ln2"],
          "layers.0.mlp.gate_proj.weight": w["l0.wgu"][: d.inter], "layers.0.mlp.up_proj.weight": w["l0.wgu"][d.inter:],
          "layers.0.mlp.down_proj.weight": w["l0.wdown"]}
    rec = {}
    orig_rope = MQ.apply_rotary_pos_emb

    def rope_spy(q, k, cos, sin, *a, **kw):
        qo, ko = orig_rope(q, k, cos, sin, *a, **kw)
        rec["q"], rec["k"] = qo.detach().clone(), ko.detach().clone()       # [1, heads, T, D]
        return qo, ko
    MQ.apply_rotary_pos_emb = rope_spy
    out = dict(seed=np.int64(G2_SEED), xseed=np.int64(G2_XSEED), ctx=np.array(G2_CTX))
    try:
        for tag, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
            m = Qwen3Model(cfg).to(dt).eval()
            missing, unexpected = m.load_state_dict({k: v.to(dt) for k, v in sd.items()}, strict=False)
            assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
            # `.to(bfloat16)` also rounds the rotary inv_freq BUFFER to bf16 (a transformers quirk: positions > 0 then rotate by
            # slightly wrong angles; 0.6 absolute error in q at position 256).  vLLM builds its cos / sin cache from fp32
            # frequencies, as does the reference's own _RotaryEmbedding: restore them.
            inv = 1.0 / (d.rope_theta ** (torch.arange(0, d.head_dim, 2, dtype=torch.float32) / d.head_dim))
            m.rotary_emb.inv_freq = inv.clone()
            if hasattr(m.rotary_emb, "original_inv_freq"):
                m.rotary_emb.original_inv_freq = inv.clone()
            m.layers[0].self_attn.o_proj.register_forward_pre_hook(lambda mod, args: rec.__setitem__("attn", args[0].detach().clone()))
            with torch.inference_mode():
                for n in G2_CTX:
                    cache = DynamicCache(config=cfg)
                    xx = x[None, :n].to(dt)
                    if n > 1:
                        m(inputs_embeds=xx[:, : n - 1], past_key_values=cache, use_cache=True)
                    o = m(inputs_embeds=xx[:, n - 1: n], past_key_values=cache, use_cache=True)
                    conv = np16 if dt == torch.bfloat16 else (lambda t: t.contiguous().float().numpy())
                    out[f"{tag}_q{n}"] = conv(rec["q"][0, :, -1])            # [Hq, D]
                    out[f"{tag}_k{n}"] = conv(rec["k"][0, :, -1])            # [Hkv, D]
                    out[f"{tag}_attn{n}"] = conv(rec["attn"][0, -1])         # [Hq * D]
                    out[f"{tag}_h{n}"] = conv(o.last_hidden_state[0, -1])    # [H]
    finally:
        MQ.apply_rotary_pos_emb = orig_rope
    np.savez_compressed(os.path.join(HERE, "qwen3_layer_real.npz"), **out)
    print("G2 qwen3_layer_real.npz", os.path.getsize(os.path.join(HERE, "qwen3_layer_real.npz")), "bytes")


# --------------------------------------------------------------------------
# G3: cache bytes after fp8 / int8 KV writes.  The fp8 bytes come from a bit-level OCP e4m3fn encoder written
# here from the format definition (4 exponent bits, bias 7, 3 mantissa bits, no infinities, S.1111.111 = NaN, max 448;
# round-to-nearest-even, saturating) -- NOT from torch's cast -- so they pin the oracle's fp8_quant and the HIP kernels alike.
def e4m3fn_encode(x: float) -> int:
    import math
    if math.isnan(x):
        return 0x7F
    sign = 0x80 if (x < 0 or (x == 0 and math.copysign(1.0, x) < 0)) else 0
    a = min(abs(x), 448.0)                       # saturating (vLLM scaled_fp8_conversion)
    if a == 0.0:
        return sign
    e = max(math.floor(math.log2(a)), -6)        # subnormals share exponent -6
    q = a / 2.0 ** (e - 3)                       # in units of the mantissa LSB at this exponent: normal -> [8, 16)
    r = math.floor(q)
    frac = q - r
    if frac > 0.5 or (frac == 0.5 and (r & 1)):
        r += 1
    if r == 16:                                  # rounded up into the next binade
        r, e = 8, e + 1
    if r < 8:                                    # subnormal (e == -6): exponent field 0, mantissa r
        return sign | r
    code = ((e + 7) << 3) | (r - 8)
    return sign | min(code, 0x7E)                # 0x7E = 448


def mint_kv_quant():
    g = torch.Generator().manual_seed(17)
    T, H, D, bs = 37, 8, 128, 16
    k = (torch.randn(T, H, D, generator=g) * 3.0).to(torch.bfloat16)
    v = (torch.randn(T, H, D, generator=g) * 40.0).to(torch.bfloat16)
    # boundary values: exact ties, subnormals, the largest finite value and beyond, signed zeros
    special = torch.tensor([0.0, -0.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0009765625 * 3, 448.0, 464.0, 480.0, 1e4, -1e4, 1.0625, 1.1875,
                            0.01953125, 240.0, 248.0, 17.0, 18.0, 19.0, -0.4375, 0.46875], dtype=torch.float32).to(torch.bfloat16)
    k[0, 0, : special.numel()] = special
    v[1, 3, : special.numel()] = -special
    block_table = [5, 2, 7]                                  # non-monotonic block ids
    slots = np.array([block_table[t // bs] * bs + t % bs for t in range(T)], dtype=np.int64)
    out = dict(block_table=np.array(block_table), slots=slots, block_size=np.int64(bs), k=np16(k), v=np16(v))
    for name, t, scale in (("k", k, 1.0), ("v", v, 1.0), ("k_s", k, 0.5), ("v_s", v, 2.0)):
        f = (t.float() / scale).numpy().reshape(-1)           # x / scale in fp32, then the saturating cast
        out["fp8_" + name] = np.array([e4m3fn_encode(float(z)) for z in f], dtype=np.uint8).reshape(T, H, D)
    for name, t in (("k", k), ("v", v)):                      # int8: build-defined: per (token, head) absmax / 127
        f = t.float().numpy().astype(np.float32)
        amax = np.maximum(np.abs(f).max(-1), np.float32(1e-8)).astype(np.float32)
        sc = (amax / np.float32(127.0)).astype(np.float32)
        q = np.clip(np.rint(f / sc[..., None]), -127, 127).astype(np.int8)
        out["int8_" + name], out["int8_scale_" + name] = q, sc
    np.savez_compressed(os.path.join(HERE, "kv_quant.npz"), **out)
    # the 256 codes decode / re-encode to themselves (sanity of the encoder against torch's table of values)
    allc = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    for c in range(256):
        if not torch.isnan(allc[c]):
            assert e4m3fn_encode(float(allc[c])) == c, c
    print("G3 kv_quant.npz", os.path.getsize(os.path.join(HERE, "kv_quant.npz")), "bytes")


# --------------------------------------------------------------------------
def mint_tts_prompt_builder():
    """Known answers of the Qwen3-TTS talker's prompt-embedding builder: the reference's own `_build_prompt_embeds` /
    `_generate_icl_prompt` (qwen3_tts_talker.py:1160-1567) bound to a stand-in `self` whose text_projection is the reference's
    Qwen3TTSTalkerResizeMLP (bf16, CPU, seeded), with a table tokenizer (text -> fixed ids: the real tokenizer is a checkpoint
    asset) -- every task type, streaming / non-streaming text, language tags, dialect override, instruct prefix, x-vector-only
    and in-context voice cloning (text longer / shorter than the reference codes)."""
    install_auto_stubs()
    # the file's relative imports resolve inside a scratch package whose path is the reference directory (read in place);
    # the 12 Hz tokenizer module (audio codec, librosa / soundfile) is not needed by the prompt builder: placeholder
    pkg = "refq3tts_talker_pkg"
    pm = types.ModuleType(pkg)
    pm.__path__ = [os.path.join(V, "model_executor/models/qwen3_tts")]
    sys.modules[pkg] = pm
    _stub(pkg + ".qwen3_tts_tokenizer", Qwen3TTSTokenizer=object)
    mod = load_by_path(pkg + ".qwen3_tts_talker", os.path.join(V, "model_executor/models/qwen3_tts/qwen3_tts_talker.py"), package=pkg)
    Cls = mod.Qwen3TTSTalkerForConditionalGeneration
    NS = types.SimpleNamespace
    g = torch.Generator().manual_seed(77)
    Vt, Ht, H, Vc, Q, Cb = 96, 64, 32, 80, 4, 40          # Ht / H multiples of 32 / 16: the device projection's tile sizes
    ids = dict(tts_bos=90, tts_eos=91, tts_pad=92, codec_nothink=60, codec_think=61, codec_think_bos=62, codec_think_eos=63,
               codec_pad=64, codec_bos=65)
    IM_START, ASSIST, NL, IM_END, USER = 1, 2, 3, 4, 5
    text_emb = torch.nn.Embedding(Vt, Ht)
    tp = mod.Qwen3TTSTalkerResizeMLP(Ht, Ht, H, "silu", bias=True)
    with torch.no_grad():
        text_emb.weight.copy_(torch.randn(Vt, Ht, generator=g) * 0.5)
        for p_ in tp.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.2 if p_.ndim == 2 else 0.1))
    text_emb, tp = text_emb.to(torch.bfloat16).eval(), tp.to(torch.bfloat16).eval()
    table = (torch.randn(Vc, H, generator=g) * 0.5).to(torch.bfloat16)
    cp_tabs = [(torch.randn(Cb, H, generator=g) * 0.5).to(torch.bfloat16) for _ in range(Q - 1)]
    vocab_text = {}

    def tok(text, return_tensors="pt", padding=False):
        return {"input_ids": torch.tensor([vocab_text[text]], dtype=torch.long)}

    tcfg = NS(codec_nothink_id=ids["codec_nothink"], codec_think_id=ids["codec_think"], codec_think_bos_id=ids["codec_think_bos"],
              codec_think_eos_id=ids["codec_think_eos"], codec_pad_id=ids["codec_pad"], codec_bos_id=ids["codec_bos"],
              codec_language_id={"english": 70, "chinese": 71, "sichuan_dialect": 72}, spk_id={"Vivian": 75, "Eric": 76},
              spk_is_dialect={"eric": "sichuan_dialect"}, num_code_groups=Q)
    cfg = NS(tts_bos_token_id=ids["tts_bos"], tts_eos_token_id=ids["tts_eos"], tts_pad_token_id=ids["tts_pad"])
    fake = NS(config=cfg, talker_config=tcfg, text_embedding=text_emb, text_projection=tp, embed_input_ids=lambda t: table[t],
              code_predictor=NS(get_input_embeddings=lambda: [lambda t, tab=tab: tab[t] for tab in cp_tabs]),
              parameters=lambda: iter([text_emb.weight]), _get_tokenizer=lambda: tok,
              _build_assistant_text=Cls._build_assistant_text, _build_ref_text=Cls._build_ref_text,
              _build_instruct_text=Cls._build_instruct_text)
    for name in ("_build_prompt_embeds", "_generate_icl_prompt"):
        setattr(fake, name, types.MethodType(getattr(Cls, name), fake))

    def text_ids(n):
        return [10 + int(x) for x in torch.randint(0, 40, (n,), generator=g)]

    def register(text, body):
        a = [IM_START, ASSIST, NL] + body + [IM_END, NL, IM_START, ASSIST, NL]
        vocab_text[Cls._build_assistant_text(text)] = a
        return a

    cases = []
    spec = [
        dict(name="custom_nonstreaming", task="CustomVoice", n=9, info=dict(speaker=["Vivian"])),
        dict(name="custom_streaming_english", task="CustomVoice", n=7, info=dict(speaker=["vivian"], language=["English"], non_streaming_mode=[False])),
        dict(name="custom_dialect_override", task="CustomVoice", n=5, info=dict(speaker=["Eric"], language=["Auto"])),
        dict(name="design_instruct", task="VoiceDesign", n=6, info=dict(instruct=["a calm low voice"], language=["chinese"])),
        dict(name="design_streaming", task="VoiceDesign", n=4, info=dict(non_streaming_mode=[False])),
        dict(name="base_xvector_streaming", task="Base", n=8, info=dict(x_vector_only_mode=[True]), spk=True),
        dict(name="base_xvector_nonstreaming", task="Base", n=3, info=dict(x_vector_only_mode=[True], non_streaming_mode=[True]), spk=True),
        dict(name="base_icl_text_longer", task="Base", n=12, info={}, spk=True, icl=dict(ref_n=5, codes=6)),
        dict(name="base_icl_codes_longer", task="Base", n=3, info={}, spk=True, icl=dict(ref_n=2, codes=11)),
        dict(name="base_icl_nonstreaming", task="Base", n=4, info=dict(non_streaming_mode=[True]), spk=True, icl=dict(ref_n=3, codes=5)),
    ]
    with torch.no_grad():
        for sp in spec:
            text = "text-" + sp["name"]
            a = register(text, text_ids(sp["n"]))
            info = dict(text=[text], **sp["info"])
            c = {"name": sp["name"], "task_type": sp["task"], "input_ids": a, "info": {k: v for k, v in sp["info"].items()}}
            if "instruct" in info:
                ins = [IM_START, USER, NL] + text_ids(5) + [IM_END, NL]
                vocab_text[Cls._build_instruct_text(info["instruct"][0])] = ins
                c["instruct_ids"] = ins
            if sp.get("spk"):
                emb = (torch.randn(H, generator=g) * 0.5).to(torch.bfloat16)
                vcp = {"ref_spk_embedding": emb}
                c["speaker_embed"] = emb
                if "icl" in sp:
                    ref_text = "ref-" + sp["name"]
                    rid = [IM_START, ASSIST, NL] + text_ids(sp["icl"]["ref_n"]) + [IM_END, NL]
                    vocab_text[Cls._build_ref_text(ref_text)] = rid
                    codes = torch.randint(0, Cb, (sp["icl"]["codes"], Q), generator=g)
                    vcp.update({"ref_code": codes, "icl_mode": True})
                    info["ref_text"] = [ref_text]
                    c["ref_ids"], c["ref_code"] = rid, codes
                info["voice_clone_prompt"] = [vcp]
            prompt, trailing, pad, rlen, rcode = fake._build_prompt_embeds(task_type=sp["task"], info_dict=info)
            c.update(out_prompt=prompt.clone(), out_trailing=trailing.clone(), out_tts_pad=pad.clone(), out_ref_code_len=rlen)
            cases.append(c)
    w = {"text_embedding": text_emb.weight.detach().clone(),
         "text_projection": {"fc1_w": tp.linear_fc1.weight.detach().clone(), "fc1_b": tp.linear_fc1.bias.detach().clone(),
                             "fc2_w": tp.linear_fc2.weight.detach().clone(), "fc2_b": tp.linear_fc2.bias.detach().clone()},
         "codec_embed": table, "cp_embed": torch.stack(cp_tabs)}
    out = {"ids": ids, "weights": w, "cases": cases, "language_ids": dict(tcfg.codec_language_id), "speaker_ids": dict(tcfg.spk_id),
           "spk_is_dialect": dict(tcfg.spk_is_dialect)}
    path = os.path.join(HERE, "tts_prompt_builder.pt")
    torch.save(out, path)
    print("tts_prompt_builder.pt", os.path.getsize(path), "bytes;", [(c["name"], tuple(c["out_prompt"].shape), tuple(c["out_trailing"].shape)) for c in cases])

# --------------------------------------------------------------------------
def mint_code2wav():
    """The reference's own Qwen3TTSTokenizerV2Decoder (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:912-1043) at the
    tiny configuration of tests/codec_util.py with that file's seeded weights (load_state_dict, strict), fp32 on CPU as the
    Code2Wav stage runs it (qwen3_tts_code2wav.py:71-75 loads torch_dtype=float32): waveforms of three code sequences, the
    stage boundaries inside forward() for the longest one, and chunked_decode with a small window.  The one adaptation is
    for the installed transformers (5.x renamed create_*_mask's `input_embeds` argument and dropped `cache_position`)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from codec_util import TINY_CODEC, make_codec_state
    install_vllm_stubs()
    pkg = "reftok12"
    pk = types.ModuleType(pkg)
    pk.__path__ = [os.path.join(V, "model_executor/models/qwen3_tts/tokenizer_12hz")]
    sys.modules[pkg] = pk
    cfgm = load_by_path(pkg + ".configuration_qwen3_tts_tokenizer_v2", os.path.join(pk.__path__[0], "configuration_qwen3_tts_tokenizer_v2.py"), pkg)
    mod = load_by_path(pkg + ".modeling_qwen3_tts_tokenizer_v2", os.path.join(pk.__path__[0], "modeling_qwen3_tts_tokenizer_v2.py"), pkg)

    def adapt(fn):
        def f(**kw):
            kw["inputs_embeds"] = kw.pop("input_embeds")
            kw.pop("cache_position", None)
            return fn(**kw)
        return f
    mod.create_causal_mask = adapt(mod.create_causal_mask)
    mod.create_sliding_window_causal_mask = adapt(mod.create_sliding_window_causal_mask)
    cfg = cfgm.Qwen3TTSTokenizerV2DecoderConfig(**TINY_CODEC)
    cfg._attn_implementation = "eager"
    dec = mod.Qwen3TTSTokenizerV2Decoder(cfg).eval()
    seed = 11
    missing, unexpected = dec.load_state_dict(make_codec_state(TINY_CODEC, seed), strict=True)
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(5)
    out = {"seed": np.int64(seed), "total_upsample": np.int64(int(dec.total_upsample))}
    taps = {}
    hooks = []
    orig_decode = dec.quantizer.decode
    dec.quantizer.decode = lambda c: taps.setdefault("quantized", orig_decode(c))      # decode() is no forward(): no hook
    hooks.append(dec.pre_conv.register_forward_hook(lambda m, i, o: taps.__setitem__("pre_conv", o)))
    hooks.append(dec.pre_transformer.register_forward_hook(lambda m, i, o: taps.__setitem__("pre_transformer", o.last_hidden_state)))
    hooks.append(dec.upsample[-1][-1].register_forward_hook(lambda m, i, o: taps.__setitem__("upsampled", o)))
    for bi in range(len(dec.decoder)):
        hooks.append(dec.decoder[bi].register_forward_hook(lambda m, i, o, bi=bi: taps.__setitem__(f"decoder{bi}", o)))
    with torch.no_grad():
        for i, T in enumerate((1, 7, 30)):
            codes = torch.randint(0, TINY_CODEC["codebook_size"], (1, TINY_CODEC["num_quantizers"], T), generator=g)
            taps.clear()
            wav = dec(codes)
            out[f"codes{i}"], out[f"wav{i}"] = codes.numpy(), wav.numpy()
        for k, v in taps.items():
            out["tap_" + k] = v.detach().numpy()
        dec.quantizer.decode = orig_decode
        out["wav_chunked"] = dec.chunked_decode(codes, chunk_size=8, left_context_size=3).numpy()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "code2wav_tiny.npz"), **out)
    print("code2wav tiny:", {k: v.shape for k, v in out.items() if k.startswith(("wav", "tap"))})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["cp", "bb", "kv", "cw", "os", "sb", "moe", "gd", "op", "tp", "g2", "g3", "c2w", "mr"]
    if "cp" in which:
        mint_code_predictor()
    if "bb" in which:
        mint_backbone()
    if "kv" in which:
        mint_kv_extract()
    if "cw" in which:
        mint_chunk_windows()
    if "os" in which:
        mint_omni_stage_processors()
    if "sb" in which:
        mint_snake_beta()
    if "moe" in which:
        mint_moe_block()
    if "gd" in which:
        mint_graph_decoder()
    if "op" in which:
        mint_omni_prompt_builder()
    if "tp" in which:
        mint_tts_prompt_builder()
    if "g2" in which:
        mint_backbone_layer_real()
    if "g3" in which:
        mint_kv_quant()
    if "c2w" in which:
        mint_code2wav()
    if "mr" in which:
        mint_mrope_positions()
