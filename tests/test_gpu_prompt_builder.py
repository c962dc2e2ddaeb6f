"""Omni talker prompt builder on the device (ht_vllm_omni_amd.prompt_builder_omni, omni_resize_mlp / omni_silu through the
C-ABI) against the oracle restatement and the known answers minted from the reference's own methods."""
import os

import pytest
import torch

from oracle import talker_oracle as O
from tests.util import BF16, assert_bf16_close, assert_e2e_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return torch.load(os.path.join(golden_dir, "omni_prompt_builder.pt"), weights_only=True)


@pytest.fixture(scope="module")
def builder(gold):
    from ht_vllm_omni_amd.prompt_builder_omni import OmniPromptIds, OmniTalkerPromptBuilder
    ids = OmniPromptIds.from_dict(gold["ids"], speaker_ids={"a": 50, "b": 51}, default_speaker="a")
    return OmniTalkerPromptBuilder(gold["weights"], ids, "cuda")


@pytest.mark.parametrize("T,H_in,I,H_out", [(1, 64, 96, 32), (37, 2048, 2048, 1024), (200, 2048, 2048, 1024)])
def test_resize_mlp_matches_oracle(T, H_in, I, H_out):
    """omni_resize_mlp at the Omni talker's real projection shape (thinker 2048 -> 2048 -> talker 1024), > 64 rows included."""
    from ht_vllm_omni_amd import ops
    g = torch.Generator().manual_seed(T + H_in)
    w = {"fc1_w": (torch.randn(I, H_in, generator=g) * 0.03).to(BF16), "fc1_b": (torch.randn(I, generator=g) * 0.1).to(BF16),
         "fc2_w": (torch.randn(H_out, I, generator=g) * 0.03).to(BF16), "fc2_b": (torch.randn(H_out, generator=g) * 0.1).to(BF16)}
    x = torch.randn(T, H_in, generator=g).to(BF16)
    got = ops.resize_mlp(x.cuda(), {k: v.cuda() for k, v in w.items()})
    # stage by stage on the device's own intermediate: each stage within one rounding of the oracle's
    h_dev = ops.gemm(x[:64].cuda(), w["fc1_w"].cuda(), bias=w["fc1_b"].cuda())
    assert_bf16_close(h_dev, O.linear(x[:64], w["fc1_w"], w["fc1_b"]), ulps=1, max_mismatch=0.03, what="fc1")
    hf = h_dev.cpu().float()
    assert torch.equal(ops.silu(h_dev).cpu(), (hf / (1.0 + torch.exp(-hf))).to(BF16)), "silu bit-exact on the same input"
    a_dev = ops.silu(h_dev)
    assert_bf16_close(ops.gemm(a_dev, w["fc2_w"].cuda(), bias=w["fc2_b"].cuda()), O.linear(a_dev.cpu(), w["fc2_w"], w["fc2_b"]),
                      ulps=1, max_mismatch=0.03, what="fc2 on the device's own activation")
    assert torch.equal(got[:64], ops.gemm(a_dev, w["fc2_w"].cuda(), bias=w["fc2_b"].cuda())), "fused entry == the three ops"
    # whole MLP: a rounding flip in the 2048-wide activation moves every output by an absolute amount -> bound at the
    # tensor's scale (tests/util.assert_e2e_close)
    ref = O.resize_mlp(x, w)
    assert got.shape == ref.shape
    assert_e2e_close(got, ref, what="resize mlp")
    noB = {k: v.cuda() for k, v in w.items() if not k.endswith("_b")}
    assert_e2e_close(ops.resize_mlp(x.cuda(), noB), O.resize_mlp(x, {k: v for k, v in w.items() if not k.endswith("_b")}),
                     what="resize mlp, no bias")


def test_prompt_builder_matches_reference_known_answers(gold, builder):
    for c in gold["cases"]:
        p = builder.prefill(c["thinker_embed"], c["thinker_hidden"], c["input_ids"], c["result_ids"], c["speaker_id"],
                            c["tts_bos"], c["tts_eos"], c["tts_pad"])
        assert torch.equal(p.input_ids, c["out_ids"]), c["name"]
        assert p.embeds.is_cuda and p.embeds.shape == c["out_embeds"].shape, c["name"]
        assert_bf16_close(p.embeds, c["out_embeds"], ulps=2, max_mismatch=0.08, what=c["name"] + " embeds vs reference")
        assert_bf16_close(p.trailing_text_hidden, c["out_trailing"], ulps=2, max_mismatch=0.08, what=c["name"] + " trailing vs reference")
        o_ids, o_emb, o_tail = O.omni_talker_prompt(c["thinker_embed"], c["thinker_hidden"], c["input_ids"], c["result_ids"],
                                                    c["speaker_id"], c["tts_bos"], c["tts_eos"], c["tts_pad"], gold["weights"], gold["ids"])
        assert torch.equal(p.input_ids, o_ids)
        assert_bf16_close(p.embeds, o_emb, ulps=2, max_mismatch=0.08, what=c["name"] + " embeds vs oracle")
        assert_bf16_close(p.tts_pad, c["tts_pad_proj"], ulps=2, max_mismatch=0.08, what="tts_pad")
        # rows that involve no arithmetic on the device side are exact: the assistant block's ids, the zero codec rows
        tail, steps = p.trailing_text_hidden, []
        for _ in range(c["decode_text_steps"].shape[0]):
            step, tail = builder.pop_text_step(tail, p.tts_pad)
            steps.append(step)
        got_steps = torch.cat(steps, 0).cpu()
        n = p.trailing_text_hidden.shape[0]
        assert torch.equal(got_steps[:n], p.trailing_text_hidden.cpu()) and torch.equal(got_steps[n:], p.tts_pad.cpu().expand(3, -1))


def test_prompt_builder_from_info_and_errors(gold, builder):
    c = gold["cases"][0]
    info = {"thinker_prefill_embeddings": c["thinker_embed"].float(), "thinker_hidden_states": c["thinker_hidden"].float(),
            "thinker_sequences": c["result_ids"].tolist(), "thinker_input_ids": c["input_ids"].tolist(),
            "tts_bos_embed": c["tts_bos"].float(), "tts_eos_embed": c["tts_eos"].float(), "tts_pad_embed": c["tts_pad"].float(),
            "speaker": ["B "], "thinker_decode_embeddings": torch.ones(2, 64)}
    p, upd = builder.from_info(info)
    want = builder.prefill(c["thinker_embed"], c["thinker_hidden"], c["input_ids"], c["result_ids"], 51, c["tts_bos"], c["tts_eos"], c["tts_pad"])
    assert torch.equal(p.embeds, want.embeds) and torch.equal(p.trailing_text_hidden, want.trailing_text_hidden)
    assert upd["prefill_consumed_text_tokens"] == 1 and upd["thinker_decode_embeddings"] is None
    assert upd["cached_thinker_decode_embeddings"].shape == (2, 64) and upd["tts_pad_embed_projected"].shape == (1, 1, 32)
    assert torch.equal(upd["trailing_text_hidden"], p.trailing_text_hidden)
    assert builder.ids.speaker_token(None) == 50 and builder.ids.speaker_token("nobody") == 50 and builder.ids.speaker_token("b") == 51
    with pytest.raises(ValueError):
        builder.from_info({"thinker_hidden_states": c["thinker_hidden"]})
    with pytest.raises(ValueError):
        builder.prefill(c["thinker_embed"], c["thinker_hidden"], c["input_ids"][:1], c["result_ids"], 50)
    bad = c["input_ids"].clone()
    bad[1] = 99
    with pytest.raises(AssertionError):
        builder.prefill(c["thinker_embed"], c["thinker_hidden"], bad, c["result_ids"], 50)


def test_streaming_text_steps_match_reference(gold, builder):
    pad, eos = gold["cases"][-1]["tts_pad_proj"].cuda(), gold["cases"][-1]["tts_eos_proj"].cuda()
    for s in gold["streaming"]["script"]:
        st = {"num_processed_tokens": s["num_processed_tokens"], "finished_flag": s["finished_flag"],
              "cached": s["cached"].cuda(), "fresh": None if s["fresh"] is None else s["fresh"].cuda()}
        out = builder.streaming_text_step(st, gold["streaming"]["n_thinker_output_ids"], eos, pad)
        assert out.shape == s["out"].shape, s["num_processed_tokens"]
        assert_bf16_close(out, s["out"], ulps=2, max_mismatch=0.08, what="streaming text step")
        if s["cached_after"] is not None:
            assert torch.equal(st["cached"].cpu(), s["cached_after"])
        assert bool(st.get("finished_flag")) == bool(s["finished_after"] or s["finished_flag"])


# ------------------------------------------------------------------ Qwen3-TTS flavour (prompt_builder_tts)
def test_tts_prompt_builder_matches_reference_fixture_and_oracle(golden_dir):
    """TTSTalkerPromptBuilder (one batched omni_resize_mlp + device gathers) against the known answers minted from the
    reference's own _build_prompt_embeds / _generate_icl_prompt (tests/golden/tts_prompt_builder.pt): every task type / mode;
    row counts exact, embeddings within one bf16 rounding of the projection GEMMs; request-level resolution (language tag,
    dialect override, speaker id, mode defaults) through from_info."""
    import os
    from ht_vllm_omni_amd.prompt_builder_tts import TTSPromptIds, TTSTalkerPromptBuilder
    from tests.test_oracle_golden import tts_case_kwargs
    g = torch.load(os.path.join(golden_dir, "tts_prompt_builder.pt"), weights_only=True)
    i = g["ids"]
    ids = TTSPromptIds(tts_bos=i["tts_bos"], tts_eos=i["tts_eos"], tts_pad=i["tts_pad"], codec_nothink=i["codec_nothink"],
                       codec_think=i["codec_think"], codec_think_bos=i["codec_think_bos"], codec_think_eos=i["codec_think_eos"],
                       codec_pad=i["codec_pad"], codec_bos=i["codec_bos"], language_ids=g["language_ids"], speaker_ids=g["speaker_ids"],
                       spk_is_dialect=g["spk_is_dialect"])
    b = TTSTalkerPromptBuilder(g["weights"], ids, "cuda")
    for c in g["cases"]:
        p = b.build(c["task_type"], c["input_ids"], **tts_case_kwargs(g, c))
        assert p.embeds.shape == c["out_prompt"].shape and p.trailing_text_hidden.shape == c["out_trailing"].shape, c["name"]
        assert p.ref_code_len == c["out_ref_code_len"], c["name"]
        assert_bf16_close(p.embeds.cpu(), c["out_prompt"], ulps=1, max_mismatch=0.03, what=c["name"] + " prompt")
        assert_bf16_close(p.trailing_text_hidden.cpu(), c["out_trailing"], ulps=1, max_mismatch=0.03, what=c["name"] + " trailing")
        assert_bf16_close(p.tts_pad.cpu(), c["out_tts_pad"], ulps=1, max_mismatch=0.03, what=c["name"] + " tts_pad")
        # the request-level path: the same case as additional_information with tokenised fields
        info = dict(c["info"], task_type=[c["task_type"]], input_ids=c["input_ids"])
        if "instruct_ids" in c:
            info["instruct_ids"] = c["instruct_ids"]
        if "speaker_embed" in c:
            vcp = {"ref_spk_embedding": c["speaker_embed"]}
            if "ref_code" in c:
                vcp.update(ref_code=c["ref_code"], icl_mode=True)
                info["ref_ids"] = c["ref_ids"]
            info["voice_clone_prompt"] = [vcp]
        p2 = b.from_info(info)
        assert torch.equal(p2.embeds, p.embeds) and torch.equal(p2.trailing_text_hidden, p.trailing_text_hidden), c["name"]
    with pytest.raises(ValueError, match="speaker"):
        b.from_info({"task_type": ["CustomVoice"], "input_ids": g["cases"][0]["input_ids"]})
    with pytest.raises(ValueError, match="Unsupported speaker"):
        b.from_info({"task_type": ["CustomVoice"], "input_ids": g["cases"][0]["input_ids"], "speaker": ["nobody"]})
    with pytest.raises(ValueError, match="tokenizer"):
        b.from_info({"task_type": ["VoiceDesign"], "text": ["hello"]})
