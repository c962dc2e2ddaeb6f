"""Host logic of the Code2Wav stage (ht_vllm_omni_amd/code2wav.py::MI355XCode2Wav) on CPU with a stand-in decoder: request
splitting, malformed requests, left-context trimming, code range validation -- the rules of Qwen3TTSCode2Wav.forward
(qwen3_tts_code2wav.py:172-309).  Also Code2WavConfig's checks.  The decoder itself has no CPU path (tests/test_gpu_code2wav.py)."""
import pytest
import torch

from ht_vllm_omni_amd.code2wav import Code2WavConfig, Code2WavDecoder, MI355XCode2Wav
from tests.codec_util import FULL_CODEC, TINY_CODEC


class _FakeDecoder:
    """wav sample (frame f, phase p) = f + p / 1000 + sum of the frame's codes / 1e6: every sample says which frame made it."""
    total_upsample = 4

    def __init__(self):
        self.cfg = Code2WavConfig.from_dict(TINY_CODEC)
        self.calls = []

    def chunked_decode(self, codes):
        self.calls.append(codes.shape)
        q, f = codes.shape[1:]
        base = torch.arange(f, dtype=torch.float32)[:, None] + torch.arange(4, dtype=torch.float32)[None] / 1000
        return (base + codes[0].sum(0)[:, None].float() / 1e6).reshape(1, 1, -1)


def test_requests_are_split_trimmed_and_malformed_ones_skipped():
    dec = _FakeDecoder()
    st = MI355XCode2Wav(dec, output_sample_rate=24000)
    a = torch.randint(0, 64, (4, 5))
    b = torch.randint(0, 64, (4, 8))
    ids = torch.cat([a.reshape(-1), torch.tensor([7, 7, 7]), b.reshape(-1)])
    out = st(ids, seq_token_counts=[20, 3, 32], runtime_additional_information=[{"left_context_size": 0}, {}, {"left_context_size": [3]}])
    wa, wm, wb = out["model_outputs"]
    assert dec.calls == [torch.Size([1, 4, 5]), torch.Size([1, 4, 8])]
    assert wa.shape == (20,) and wa[0].item() == pytest.approx(a[:, 0].sum().item() / 1e6)
    assert wm.numel() == 0                                        # 3 ids are no whole frame of 4 quantizers
    assert wb.shape == (20,) and int(wb[0].item()) == 3           # the first 3 of 8 frames were context: 3/8 of the samples dropped
    assert all(int(s) == 24000 and s.dtype == torch.int32 for s in out["sr"])


def test_single_request_without_counts_and_empty_input():
    dec = _FakeDecoder()
    st = MI355XCode2Wav(dec)
    ids = torch.randint(0, 64, (4 * 6,))
    out = st(ids)
    assert out["model_outputs"][0].shape == (24,) and len(out["model_outputs"]) == 1
    assert st(None)["model_outputs"][0].numel() == 0 and st(torch.zeros(0, dtype=torch.long))["model_outputs"][0].numel() == 0
    # context as large as the request: nothing new to emit
    out = st(ids, runtime_additional_information=[{"left_context_size": torch.tensor([6])}])
    assert out["model_outputs"][0].numel() == 0


def test_out_of_range_codes_are_refused_before_any_launch():
    dec = _FakeDecoder()
    st = MI355XCode2Wav(dec)
    with pytest.raises(ValueError, match="outside"):
        st(torch.tensor([0, 1, 2, 64]))
    with pytest.raises(ValueError, match="outside"):
        st(torch.tensor([0, -1, 2, 3]))
    assert dec.calls == []
    MI355XCode2Wav(dec, validate_codes=False)(torch.tensor([0, 1, 2, 64]))
    assert len(dec.calls) == 1


def test_config_from_dict_and_checks():
    c = Code2WavConfig.from_dict({**FULL_CODEC, "unknown_key": 1, "upsample_rates": [8, 5, 4, 3]})
    assert c.total_upsample == 1920 and c.head_dim == 64 and c.upsample_rates == (8, 5, 4, 3)
    c.check()
    with pytest.raises(ValueError, match="multiples of 32"):
        Code2WavConfig.from_dict({**TINY_CODEC, "decoder_dim": 64}).check()          # 64 / 4 = 16 channels in the last block
    with pytest.raises(ValueError, match="head_dim"):
        Code2WavConfig.from_dict({**TINY_CODEC, "num_attention_heads": 4, "num_key_value_heads": 4}).check()   # head_dim 32


def test_decoder_has_no_cpu_path():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ht_vllm_omni_amd._lib import OmniError
    with pytest.raises(OmniError, match="no CPU fallback"):
        Code2WavDecoder(TINY_CODEC, {})


def test_code2wav_checkpoint_loader_reads_the_decoder_of_a_speech_tokenizer_directory(tmp_path):
    """A synthetic speech_tokenizer/ directory (config.json + model.safetensors with decoder.* and encoder.* tensors): the loader
    returns the decoder's config and ONLY its tensors, prefix stripped, bit-exact."""
    import json, os
    from safetensors.torch import save_file
    from ht_vllm_omni_amd.checkpoint import load_code2wav_checkpoint
    from tests.codec_util import make_codec_state
    sd = make_codec_state(TINY_CODEC, 5)
    st = tmp_path / "model" / "speech_tokenizer"
    os.makedirs(st)
    tensors = {"decoder." + k: v.contiguous() for k, v in sd.items()}
    tensors["encoder.layers.0.weight"] = torch.zeros(4, 4)
    save_file(tensors, str(st / "model.safetensors"))
    json.dump({"decoder_config": {**TINY_CODEC, "upsample_rates": list(TINY_CODEC["upsample_rates"]),
                                  "upsampling_ratios": list(TINY_CODEC["upsampling_ratios"])}, "output_sample_rate": 24000,
               "encoder_config": {}}, open(st / "config.json", "w"))
    for root in (tmp_path / "model", st):
        cfg, state = load_code2wav_checkpoint(str(root))
        assert set(state) == set(sd) and all(torch.equal(state[k], sd[k]) for k in sd)
        c = Code2WavConfig.from_dict(cfg)
        assert c.total_upsample == 48 and c.num_quantizers == 4 and c.output_sample_rate == 24000
    json.dump({"decoder_config": {}}, open(st / "config.json", "w"))
    with pytest.raises(ValueError, match="num_quantizers"):
        load_code2wav_checkpoint(str(st))
