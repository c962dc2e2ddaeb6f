"""Pins oracle/code2wav_oracle.py to the reference's own decoder: tests/golden/code2wav_tiny.npz holds outputs of
Qwen3TTSTokenizerV2Decoder (fp32, CPU) at the tiny configuration with the seeded weights of tests/codec_util.py
(minted by tests/golden/make_fixtures.py::mint_code2wav)."""
import os

import numpy as np
import pytest
import torch

from oracle.code2wav_oracle import Code2WavOracle
from tests.codec_util import TINY_CODEC, make_codec_state, total_upsample


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "code2wav_tiny.npz"))


@pytest.fixture(scope="module")
def orc(z):
    return Code2WavOracle(TINY_CODEC, make_codec_state(TINY_CODEC, int(z["seed"])))


def test_waveforms_match_the_reference_decoder(z, orc):
    assert int(z["total_upsample"]) == orc.total_upsample == total_upsample(TINY_CODEC) == 48
    for i in range(3):
        wav = orc(torch.from_numpy(z[f"codes{i}"]))
        ref = torch.from_numpy(z[f"wav{i}"])
        assert wav.shape == ref.shape
        assert (wav - ref).abs().max().item() <= 2e-5, i          # fp32 both sides; summation order differs


def test_every_stage_boundary_matches(z, orc):
    taps = {}
    orc(torch.from_numpy(z["codes2"]), taps)
    for k in ("quantized", "pre_conv", "pre_transformer", "upsampled", "decoder0", "decoder1", "decoder2", "decoder3", "decoder4"):
        ref = torch.from_numpy(z["tap_" + k])[0]
        got = taps[k]
        assert got.shape == ref.shape, k
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k


def test_chunked_decode_matches(z, orc):
    wav = orc.chunked_decode(torch.from_numpy(z["codes2"]), chunk_size=8, left_context_size=3)
    assert (wav - torch.from_numpy(z["wav_chunked"])).abs().max().item() <= 2e-5


def test_wrong_quantizer_count_raises(orc):
    with pytest.raises(ValueError):
        orc(torch.zeros(1, 3, 5, dtype=torch.long))


def test_bf16_points_oracle_stays_close_to_the_fp32_one(z):
    """The rounding model of the HIP path (bf16 weights and stored activations) against the fp32 reference values: the
    size of the bf16 effect that the GPU parity test's tolerance has to allow."""
    sd = make_codec_state(TINY_CODEC, int(z["seed"]))
    wav = Code2WavOracle(TINY_CODEC, sd, bf16_points=True)(torch.from_numpy(z["codes2"]))
    ref = torch.from_numpy(z["wav2"])
    d = (wav - ref).abs()
    assert d.mean().item() <= 6e-3 and d.max().item() <= 6e-2, (d.mean().item(), d.max().item())
