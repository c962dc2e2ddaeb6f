"""CPU oracle of the Code2Wav (12 Hz speech-tokenizer) decoder -- TEST INFRASTRUCTURE, not product code: only tests/,
__graft_entry__.smoke() and bench scripts' cpu legs may import it.

A functional fp32 restatement (plain torch ops on CPU, no nn.Module, no reference import) of
  /root/reference/vllm_omni/model_executor/models/qwen3_tts/tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py
    Qwen3TTSTokenizerV2Decoder.forward / chunked_decode   :1009-1043
    SplitResidualVectorQuantizer.decode                   :768-909
    CausalConvNet / CausalTransConvNet / ConvNeXtBlock    :174-258
    DecoderTransformerModel (sliding-window attention)    :261-600
    SnakeBeta                                             :602-724
    DecoderResidualUnit / DecoderBlock                    :726-765
over a state dict with the reference's parameter names, in fp32 -- the dtype the Code2Wav stage loads the decoder in
(qwen3_tts_code2wav.py:71-75).  PINNED: tests/test_code2wav_oracle.py checks it against tests/golden/code2wav_tiny.npz,
outputs of the reference's own module (waveforms, every stage boundary, chunked_decode), minted by
tests/golden/make_fixtures.py::mint_code2wav.
Activations here are channel-major [C, T] like the reference; `bf16_points=True` additionally rounds to bf16 wherever the
HIP path stores an activation or holds a weight in bf16 (the oracle of the product's ROUNDING, used for tight kernel tests).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _r(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).float() if on else x


class Code2WavOracle:
    def __init__(self, cfg: dict, sd: dict[str, torch.Tensor], bf16_points: bool = False):
        self.cfg, self.bf = cfg, bf16_points
        self.sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.total_upsample = math.prod(cfg["upsample_rates"]) * math.prod(cfg["upsampling_ratios"])

    # ---- primitives
    def w(self, name: str) -> torch.Tensor:
        """A GEMM / conv weight (bf16 in the product)."""
        return _r(self.sd[name], self.bf)

    def causal_conv(self, x: torch.Tensor, name: str, dilation: int = 1, groups: int = 1) -> torch.Tensor:
        """…:174-207, stride 1: left padding (k - 1) * dilation, no right padding.  x [C, T]."""
        wt = self.w(name + ".conv.weight") if groups == 1 else self.sd[name + ".conv.weight"]
        k = wt.shape[-1]
        xp = F.pad(x[None], ((k - 1) * dilation, 0))
        return F.conv1d(xp, wt, self.sd[name + ".conv.bias"], dilation=dilation, groups=groups)[0]

    def trans_conv(self, x: torch.Tensor, name: str, stride: int) -> torch.Tensor:
        """…:210-224: ConvTranspose1d(kernel k, stride), the last k - stride samples dropped."""
        wt = self.w(name + ".conv.weight")
        k = wt.shape[-1]
        y = F.conv_transpose1d(x[None], wt, self.sd[name + ".conv.bias"], stride=stride)[0]
        return y[:, : y.shape[-1] - (k - stride)] if k > stride else y

    def snake(self, x: torch.Tensor, name: str) -> torch.Tensor:
        """…:602-724: x + 1 / (exp(beta) + 1e-9) * sin^2(x * exp(alpha)), per channel."""
        a = torch.exp(self.sd[name + ".alpha"])[:, None]
        ib = (1.0 / (torch.exp(self.sd[name + ".beta"]) + 1e-9))[:, None]
        return x + ib * torch.sin(x * a) ** 2

    def rms_norm(self, x: torch.Tensor, name: str) -> torch.Tensor:          # x [T, H]  …:397-414
        v = x.pow(2).mean(-1, keepdim=True)
        return self.sd[name + ".weight"] * (x * torch.rsqrt(v + self.cfg["rms_norm_eps"]))

    # ---- stages
    def quantizer_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [Q, T] -> [codebook_dim, T]  (…:768-909: embedding = embedding_sum / clamp(usage, 1e-5); the first quantizer and
        the sum of the others each go through their own 1x1 output projection)."""
        out = 0
        for part, qs in (("rvq_first", codes[:1]), ("rvq_rest", codes[1:])):
            if qs.shape[0] == 0:
                continue
            acc = 0
            for q, c in enumerate(qs):
                p = f"quantizer.{part}.vq.layers.{q}._codebook."
                emb = self.sd[p + "embedding_sum"] / self.sd[p + "cluster_usage"].clamp(min=1e-5)[:, None]
                acc = acc + emb[c.long()]                                   # [T, dim]
            out = out + F.conv1d(acc.T[None], self.sd[f"quantizer.{part}.output_proj.weight"])[0]
        return out

    def attention(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """Sliding-window causal self-attention with RoPE (…:305-377, eager_attention_forward :136-160, mask = keys j with
        0 <= i - j < sliding_window).  x [T, H]."""
        c = self.cfg
        T = x.shape[0]
        nh, nkv = c["num_attention_heads"], c["num_key_value_heads"]
        hd = c["hidden_size"] // nh
        q = (x @ self.w(p + "q_proj.weight").T).view(T, nh, hd).transpose(0, 1)
        k = (x @ self.w(p + "k_proj.weight").T).view(T, nkv, hd).transpose(0, 1)
        v = (x @ self.w(p + "v_proj.weight").T).view(T, nkv, hd).transpose(0, 1)
        q, k, v = _r(q, self.bf), _r(k, self.bf), _r(v, self.bf)
        inv = 1.0 / (c["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float32, device=x.device) / hd))                # …:51-66
        fr = torch.arange(T, dtype=torch.float32, device=x.device)[:, None] * inv[None]
        cos, sin = torch.cat([fr, fr], -1).cos().to(x.dtype), torch.cat([fr, fr], -1).sin().to(x.dtype)

        def rot(t):
            return torch.cat([-t[..., hd // 2:], t[..., : hd // 2]], -1)
        q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
        q, k = _r(q, self.bf), _r(k, self.bf)
        if nkv != nh:
            k, v = k.repeat_interleave(nh // nkv, 0), v.repeat_interleave(nh // nkv, 0)
        s = (q @ k.transpose(1, 2)) * hd ** -0.5
        i, j = torch.arange(T, device=x.device)[:, None], torch.arange(T, device=x.device)[None]
        s = s.masked_fill(~((j <= i) & (i - j < c["sliding_window"])), float("-inf"))
        o = torch.softmax(s, -1) @ v
        o = _r(o.transpose(0, 1).reshape(T, nh * hd), self.bf)
        return o @ self.w(p + "o_proj.weight").T

    def pre_transformer(self, x: torch.Tensor) -> torch.Tensor:
        """x [T, latent] -> [T, latent]  (…:496-600).  Rounding model: the hidden state is an fp32 stream; GEMM operands bf16."""
        bf = self.bf
        h = x @ self.w("pre_transformer.input_proj.weight").T + self.sd["pre_transformer.input_proj.bias"]
        for l in range(self.cfg["num_hidden_layers"]):
            p = f"pre_transformer.layers.{l}."
            a = _r(self.rms_norm(h, p + "input_layernorm"), bf)
            h = h + self.attention(a, p + "self_attn.") * self.sd[p + "self_attn_layer_scale.scale"]
            a = _r(self.rms_norm(h, p + "post_attention_layernorm"), bf)
            g = _r(a @ self.w(p + "mlp.gate_proj.weight").T, bf)
            u = _r(a @ self.w(p + "mlp.up_proj.weight").T, bf)
            m = _r(_r(F.silu(g), bf) * u, bf)
            h = h + (m @ self.w(p + "mlp.down_proj.weight").T) * self.sd[p + "mlp_layer_scale.scale"]
        h = _r(self.rms_norm(h, "pre_transformer.norm"), bf)
        return _r(h @ self.w("pre_transformer.output_proj.weight").T + self.sd["pre_transformer.output_proj.bias"], bf)

    def convnext(self, x: torch.Tensor, p: str) -> torch.Tensor:
        """…:226-258.  x [C, T] (fp32 stream in the rounding model)."""
        bf = self.bf
        h = self.causal_conv(x, p + "dwconv", groups=x.shape[0]).T               # [T, C]
        h = _r(F.layer_norm(h, (h.shape[-1],), self.sd[p + "norm.weight"], self.sd[p + "norm.bias"], 1e-6), bf)
        h = _r(F.gelu(h @ self.w(p + "pwconv1.weight").T + self.sd[p + "pwconv1.bias"]), bf)
        h = (h @ self.w(p + "pwconv2.weight").T + self.sd[p + "pwconv2.bias"]) * self.sd[p + "gamma"]
        return _r(x + h.T, bf)

    def forward(self, codes: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
        """codes [1, Q, T] (or [Q, T]) -> waveform [1, 1, T * total_upsample] in [-1, 1]  (…:1009-1027).
        Rounding model (bf16_points): what a GEMM consumes is bf16 -- the quantizer output, normalised / activated tensors, every
        snake output -- while the residual signals (transformer hidden state, ConvNeXt input, the decoder blocks' signal) are
        fp32 and snake is taken from the fp32 value."""
        c, bf = self.cfg, self.bf
        codes = codes.reshape(-1, codes.shape[-1])
        if codes.shape[0] != c["num_quantizers"]:
            raise ValueError(f"Expected {c['num_quantizers']} layer of codes, got {codes.shape[0]}")

        def tap(k, v):
            if taps is not None:
                taps[k] = v
            return v
        h = tap("quantized", _r(self.quantizer_decode(codes), bf))
        h = tap("pre_conv", _r(self.causal_conv(h, "pre_conv"), bf))
        h = tap("pre_transformer", self.pre_transformer(h.T)).T
        for i, f in enumerate(c["upsampling_ratios"]):
            h = self.convnext(self.trans_conv(h, f"upsample.{i}.0", f), f"upsample.{i}.1.")
        tap("upsampled", h)
        h = tap("decoder0", self.causal_conv(h, "decoder.0"))
        for i, r in enumerate(c["upsample_rates"]):
            p = f"decoder.{i + 1}.block."
            h = self.trans_conv(_r(self.snake(h, p + "0"), bf), p + "1", r)
            for u, dil in enumerate((1, 3, 9)):
                q = f"{p}{u + 2}."
                t = self.causal_conv(_r(self.snake(h, q + "act1"), bf), q + "conv1", dilation=dil)
                h = h + self.causal_conv(_r(self.snake(t, q + "act2"), bf), q + "conv2")
            tap(f"decoder{i + 1}", h)
        n = len(c["upsample_rates"])
        h = tap(f"decoder{n + 1}", _r(self.snake(h, f"decoder.{n + 1}"), bf))
        wt = self.sd[f"decoder.{n + 2}.conv.weight"]
        k = wt.shape[-1]
        wav = F.conv1d(F.pad(h[None], (k - 1, 0)), wt, self.sd[f"decoder.{n + 2}.conv.bias"])
        tap(f"decoder{n + 2}", wav[0])
        return wav.clamp(min=-1, max=1)

    __call__ = forward

    def chunked_decode(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        """…:1029-1043 (eager branch)."""
        wavs, start, T = [], 0, codes.shape[-1]
        while start < T:
            end = min(start + chunk_size, T)
            ctx = left_context_size if start - left_context_size > 0 else start
            w = self.forward(codes[..., start - ctx: end])
            wavs.append(w[..., ctx * self.total_upsample:])
            start = end
        return torch.cat(wavs, -1)
