"""hipGraph replay of the Code2Wav (12 Hz tokenizer) decoder over fixed code-length buckets.

Mirrors `CUDAGraphDecoderWrapper` (/root/reference/vllm_omni/model_executor/models/qwen3_tts/cuda_graph_decoder_wrapper.py:17-177):
same constructor / `compute_capture_sizes` / `warmup` / `decode` / `chunked_decode_with_cudagraph` surface and the same
fallback rules (disabled, not warmed up, batch != 1, longer than every bucket, capture failed -> eager decoder).  Differences,
for a 288 GB MI355X that serves many buckets from one process:
  * all bucket graphs are captured into ONE private memory pool, largest bucket first, so the activations of the 11-19
    buckets share storage instead of adding up;
  * the static input of a bucket is not cleared on every call: only the stale tail beyond the new length is zeroed (the
    bucket remembers how much of it the previous call filled), i.e. one copy launch for back-to-back equal lengths -- the
    streaming steady state -- instead of a fill + a copy;
  * replay and the output slice copy run on the caller's current stream (torch's graph replay is stream-ordered).
The decoder itself is any module with `total_upsample` (its convolutions / transformer are not part of this package; the
SnakeBeta activation inside it is `omni_snake_beta`).
"""
from __future__ import annotations

import bisect
import logging

import torch

logger = logging.getLogger(__name__)


class HipGraphDecoderWrapper:
    def __init__(self, decoder: torch.nn.Module, capture_sizes: list[int] | None = None, num_quantizers: int = 8,
                 enabled: bool = True):
        self.decoder = decoder
        self._explicit_sizes = capture_sizes is not None
        self.capture_sizes = sorted(capture_sizes) if capture_sizes else []
        self.num_quantizers = num_quantizers
        self.enabled = enabled
        self.graphs: dict[int, torch.cuda.CUDAGraph] = {}
        self.static_inputs: dict[int, torch.Tensor] = {}
        self.static_outputs: dict[int, torch.Tensor] = {}
        self._filled: dict[int, int] = {}          # bucket -> columns of its static input that may be non-zero
        self._pool = None
        self._warmed_up = False
        self._device = None
        self.stats = {"replays": 0, "eager": 0}

    @staticmethod
    def compute_capture_sizes(codec_chunk_frames: int = 0, codec_left_context_frames: int = 0, decode_chunk_size: int = 300,
                              decode_left_context: int = 25) -> list[int]:
        """Bucket lengths worth a graph (the reference's table, cuda_graph_decoder_wrapper.py:52-78, pinned by
        tests/golden/graph_decoder.json): every power of two from 2 to 256 that a full non-streaming window
        (decode_chunk_size + decode_left_context frames) can hold, that window itself, and the two streaming window lengths
        (a bare chunk; a chunk with its left context)."""
        full = decode_chunk_size + decode_left_context
        buckets = {1 << e for e in range(1, 9) if (1 << e) <= full} | {full}
        if codec_chunk_frames > 0:
            buckets |= {codec_chunk_frames} | ({codec_chunk_frames + codec_left_context_frames} if codec_left_context_frames > 0 else set())
        return sorted(buckets)

    def _get_padded_size(self, actual_size: int) -> int | None:
        """Smallest captured bucket that holds `actual_size` frames (capture_sizes is kept sorted)."""
        i = bisect.bisect_left(self.capture_sizes, actual_size)
        return self.capture_sizes[i] if i < len(self.capture_sizes) else None

    def warmup(self, device: torch.device, dtype: torch.dtype = torch.long, codec_chunk_frames: int = 0,
               codec_left_context_frames: int = 0) -> None:
        device = torch.device(device)
        if device.type != "cuda" or not self.enabled or self._warmed_up:
            return
        self._device = device
        self.decoder.eval()
        if not self._explicit_sizes:
            self.capture_sizes = self.compute_capture_sizes(codec_chunk_frames=codec_chunk_frames,
                                                            codec_left_context_frames=codec_left_context_frames)
        logger.info("hipGraph warmup for %d sizes: %s", len(self.capture_sizes), self.capture_sizes)
        with torch.no_grad():
            for size in self.capture_sizes:        # eager passes first: lazy allocations / kernel loads happen outside capture
                self.decoder(torch.zeros(1, self.num_quantizers, size, dtype=dtype, device=device))
        torch.cuda.synchronize(device)
        self._pool = torch.cuda.graph_pool_handle()
        for size in sorted(self.capture_sizes, reverse=True):      # largest first: later buckets fit into its pool blocks
            try:
                self._capture(size, device, dtype)
            except Exception:
                logger.warning("failed to capture the decoder graph for size=%d", size, exc_info=True)
        self._warmed_up = True
        logger.info("hipGraph warmup complete: %d/%d captured", len(self.graphs), len(self.capture_sizes))

    def _capture(self, size: int, device: torch.device, dtype: torch.dtype) -> None:
        static_input = torch.zeros(1, self.num_quantizers, size, dtype=dtype, device=device)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, pool=self._pool):
            static_output = self.decoder(static_input)
        self.graphs[size], self.static_inputs[size], self.static_outputs[size] = graph, static_input, static_output
        self._filled[size] = 0

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        if not self.enabled or not self._warmed_up or codes.shape[0] != 1:
            self.stats["eager"] += 1
            return self.decoder(codes)
        n = codes.shape[-1]
        size = self._get_padded_size(n)
        if size is None or size not in self.graphs:
            self.stats["eager"] += 1
            return self.decoder(codes)
        buf = self.static_inputs[size]
        if self._filled[size] > n:
            buf[:, :, n:self._filled[size]].zero_()
        buf[:, :, :n].copy_(codes)
        self._filled[size] = n
        self.graphs[size].replay()
        self.stats["replays"] += 1
        return self.static_outputs[size][..., : n * self.decoder.total_upsample].clone()

    @staticmethod
    def chunk_windows(total: int, chunk_size: int, left_context_size: int):
        """(first frame, last frame + 1, context frames) of every window of a non-streaming decode: consecutive chunks, each
        preceded by `left_context_size` frames of its predecessor -- except that a chunk starting within the first
        `left_context_size` frames takes everything before it (…wrapper.py:154-177)."""
        for start in range(0, total, chunk_size):
            ctx = left_context_size if start > left_context_size else start
            yield start - ctx, min(start + chunk_size, total), ctx

    def chunked_decode_with_cudagraph(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        """Non-streaming decode window by window; the samples of each window's context frames are dropped."""
        up = self.decoder.total_upsample
        return torch.cat([self.decode(codes[..., lo:hi])[..., ctx * up:]
                          for lo, hi, ctx in self.chunk_windows(codes.shape[-1], chunk_size, left_context_size)], dim=-1)

    chunked_decode = chunked_decode_with_cudagraph
