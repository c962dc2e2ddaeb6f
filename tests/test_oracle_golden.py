"""Pin the CPU oracle against the golden vectors minted from the reference
(tests/golden/make_fixtures.py): the reference's own code predictor file and HF Qwen3Model."""
import os

import numpy as np
import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from oracle import talker_oracle as O
from tests.util import assert_bf16_close


def _bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def test_code_predictor_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "code_predictor_tiny.npz"))
    d = get_dims("tiny")
    w = make_weights(d, seed=int(z["seed"]), std=float(z["std"]), norm_noise=float(z["norm_noise"]))
    orc = O.TalkerOracle(d, w)
    code0 = torch.from_numpy(z["layer0_code"]).reshape(-1)
    e0 = _bf16(z["layer0_embed"]).reshape(code0.shape[0], -1)
    lh = _bf16(z["last_talker_hidden"]).reshape(code0.shape[0], -1)
    codes = orc.code_predictor(code0, e0, lh, do_sample=False)
    assert torch.equal(codes, torch.from_numpy(z["all_codes"]))          # integer codes: bit-exact
    buf = _bf16(z["proj_buf"])
    hid = orc.cp_model(buf)
    ref = _bf16(z["final_hidden"]).float()
    # bf16 activations; the reference's SDPA (CPU flash path) rounds P to bf16 before PV and sums
    # GEMMs in another order, the oracle keeps P in fp32: all
    # positions agree to ~1 bf16 ulp (2^-7 at |x|~1..2).
    err = (hid.float() - ref).abs()
    assert err.max().item() <= 6.3e-2
    assert err.mean().item() <= 8e-3


def test_backbone_matches_hf_qwen3(golden_dir):
    z = np.load(os.path.join(golden_dir, "qwen3_backbone_tiny.npz"))
    d = get_dims("tiny")
    w = make_weights(d, seed=int(z["seed"]), std=float(z["std"]), norm_noise=float(z["norm_noise"]))
    n_dec = int(z["n_decode"])
    for r, n in enumerate(z["prompt_lens"].tolist()):
        orc = O.TalkerOracle(d, w, kv_dtype="bf16", num_blocks=8, block_size=16)
        x = _bf16(z[f"x{r}"])
        ref = _bf16(z[f"h{r}"]).float()
        bt = [[3, 1, 6, 2]]   # deliberately non-monotonic block ids
        hs = [orc.backbone(x[:n], torch.arange(n), [0] * n, bt, [n])]
        for t in range(n_dec):
            hs.append(orc.backbone(x[n + t: n + t + 1], torch.tensor([n + t]), [0], bt, [n + t + 1]))
        got = torch.cat(hs, 0).float()
        err = (got - ref).abs()
        assert err.max().item() <= 6.3e-2, (r, err.max().item())      # <= 2 bf16 ulp at |h|<8
        assert err.mean().item() <= 8e-3, (r, err.mean().item())  # HF eager attention rounds P to bf16; ~0.5 ulp mean


def test_kv_extract_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "kv_extract.npz"))
    cache = torch.from_numpy(z["cache"])
    i = 0
    while f"ids{i}" in z:
        ids, seq = z[f"ids{i}"].tolist(), int(z[f"seq{i}"])
        for layout in ("2first", "2second"):
            lk = cache if layout == "2first" else cache.transpose(0, 1).contiguous()
            k, v = O.extract_kv(lk, ids, seq)
            assert torch.equal(k, torch.from_numpy(z[f"k{i}_{layout}"]))
            assert torch.equal(v, torch.from_numpy(z[f"v{i}_{layout}"]))
        i += 1
    assert i == 4


def test_snake_beta_matches_reference_module(golden_dir):
    """Oracle restatement of the Code2Wav decoder's SnakeBeta against outputs of the reference's own module."""
    z = np.load(os.path.join(golden_dir, "snake_beta.npz"))
    for i in range(int(z["n"])):
        x, a, b, y = (torch.from_numpy(z[f"{k}{i}"]) for k in ("x", "alpha", "beta", "y"))
        got = O.snake_beta(x, a, b)
        torch.testing.assert_close(got, y, rtol=1e-6, atol=1e-6)


def test_moe_block_matches_hf_module(golden_dir):
    """Oracle restatement of the Omni talker's sparse-MoE MLP against HF's Qwen3OmniMoeTalkerTextSparseMoeBlock (bf16):
    routing indices and weights bit-exact, block output bit-exact (same rounding points, same accumulation order)."""
    from tests.util import bf16_from_u16, make_moe_weights
    z = np.load(os.path.join(golden_dir, "moe_block.npz"))
    for ci in range(int(z["n"])):
        H, E, K, I, Is, T, norm, seed = (int(v) for v in z[f"c{ci}_meta"])
        w = make_moe_weights(H, E, I, Is, seed)
        x = bf16_from_u16(z[f"c{ci}_x"]).reshape(T, H)
        _, rw, ri = O.moe_route(x, w["router"], K, bool(norm))
        assert torch.equal(ri, torch.from_numpy(z[f"c{ci}_topk_idx"]))
        assert torch.equal(rw.view(torch.int16), bf16_from_u16(z[f"c{ci}_topk_w"]).reshape(T, K).view(torch.int16))
        y = O.moe_block(x, w, K, bool(norm))
        ref = bf16_from_u16(z[f"c{ci}_y"]).reshape(T, H)
        diff = (y.float() - ref.float()).abs()
        assert (y.view(torch.int16) != ref.view(torch.int16)).float().mean().item() < 0.02, f"case {ci}: max diff {diff.max().item()}"
        assert diff.max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item())


@pytest.fixture(scope="module")
def omni_prompt_gold(golden_dir):
    return torch.load(os.path.join(golden_dir, "omni_prompt_builder.pt"), weights_only=True)


def test_omni_prompt_builder_matches_reference_methods(omni_prompt_gold):
    """oracle.omni_talker_prompt == the reference's _thinker_to_talker_prefill (+ user / assistant parts, _get_tts_embed) run
    on HF ResizeMLP modules: ids exact, embeddings / trailing text rows within one bf16 rounding of the module's GEMMs."""
    g = omni_prompt_gold
    w, ids = g["weights"], g["ids"]
    for c in g["cases"]:
        o_ids, o_emb, o_tail = O.omni_talker_prompt(c["thinker_embed"], c["thinker_hidden"], c["input_ids"], c["result_ids"],
                                                    c["speaker_id"], c["tts_bos"], c["tts_eos"], c["tts_pad"], w, ids)
        assert torch.equal(o_ids, c["out_ids"]), c["name"]
        assert o_emb.shape == c["out_embeds"].shape and o_tail.shape == c["out_trailing"].shape, c["name"]
        assert_bf16_close(o_emb, c["out_embeds"], ulps=1, max_mismatch=0.02, what=c["name"] + " embeds")
        assert_bf16_close(o_tail, c["out_trailing"], ulps=1, max_mismatch=0.02, what=c["name"] + " trailing")
        # decode side: the queue pops, then tts_pad for ever (talker_preprocess_decode)
        tail, steps = c["out_trailing"], []
        for _ in range(c["decode_text_steps"].shape[0]):
            step, tail = O.omni_text_step_pop(tail, c["tts_pad_proj"])
            steps.append(step)
        assert torch.equal(torch.cat(steps, 0), c["decode_text_steps"]), c["name"]
    with pytest.raises(ValueError):
        c = g["cases"][0]
        O.omni_talker_prompt(c["thinker_embed"], c["thinker_hidden"], c["input_ids"][:1], c["result_ids"], 1, None, None, None, w, ids)


def test_omni_streaming_text_steps_match_reference(omni_prompt_gold):
    g = omni_prompt_gold
    pad, eos = g["cases"][-1]["tts_pad_proj"], g["cases"][-1]["tts_eos_proj"]
    for s in g["streaming"]["script"]:
        st = {"num_processed_tokens": s["num_processed_tokens"], "finished_flag": s["finished_flag"], "cached": s["cached"],
              "fresh": s["fresh"]}
        out = O.omni_text_step_streaming(st, g["streaming"]["n_thinker_output_ids"], eos, pad, g["weights"]["text"])
        assert out.shape == s["out"].shape, s["num_processed_tokens"]
        assert_bf16_close(out, s["out"], ulps=1, max_mismatch=0.02, what="streaming text step")
        if s["cached_after"] is not None:
            assert torch.equal(st["cached"], s["cached_after"])
        assert bool(st.get("finished_flag")) == bool(s["finished_after"] or s["finished_flag"])
