"""Code2Wav stage on MI355X: the 12 Hz speech-tokenizer decoder (codec codes -> 24 kHz waveform) on the HIP kernels.

Mirrors, behind the same call surface,
  * `Qwen3TTSTokenizerV2Decoder` (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:912-1043): `Code2WavDecoder` --
    `forward(codes[1, Q, T]) -> wav[1, 1, T * total_upsample]`, `chunked_decode`, `total_upsample`, `enable_cudagraph` /
    `disable_cudagraph`, `precompute_snake_caches`;
  * `Qwen3TTSCode2Wav.forward` (qwen3_tts_code2wav.py:172-309): `MI355XCode2Wav` -- the generation-runner model of stage 1:
    flat `input_ids` per request ([Q * F] codebook-major), `left_context_size` from the runtime info, context trimming.

Design (not the reference's: that is a tree of torch modules on channel-major [B, C, T] fp32 tensors):
  * activations are TIME-major [T, C]; every conv / transposed conv / linear is ONE `omni_gemm_tile` launch over row windows of
    the input (no im2col, no padding copies); a transposed conv of stride s writes [T, s * C_out] = the up-sampled [T * s, C_out];
  * GEMM operands are bf16 (fp32 accumulate on the matrix cores); the RESIDUAL STREAMS (transformer hidden state, ConvNeXt
    input, the decoder blocks' signal) stay fp32 in HBM, updated in place by the GEMM epilogue: SnakeBeta's sin(alpha * x) is
    taken from the fp32 value (a bf16 ulp at |x| ~ 16 would be a 0.1 rad phase error);
  * every SnakeBeta is fused into the epilogue of the GEMM that produces its input (second output bf16(snake(y))): no
    activation pass over the 24 kHz-rate tensors; LayerScale / ConvNeXt gamma / biases / GELU / SiLU * up likewise;
  * the residual-VQ lookup is one gather-sum over tables with the 1x1 output projections folded in at load.
There is no CPU path: the constructor needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
from dataclasses import dataclass, fields

import torch

from . import _lib as L
from . import ops
from .engine import frag_shuffle, gu8_shuffle

logger = logging.getLogger(__name__)
BF16 = torch.bfloat16


@dataclass
class Code2WavConfig:
    """Qwen3TTSTokenizerV2DecoderConfig (configuration_qwen3_tts_tokenizer_v2.py:74-118); codebook_dim comes from the checkpoint's
    config.json (the class has no default for it)."""
    codebook_size: int = 2048
    codebook_dim: int = 512
    hidden_size: int = 1024
    latent_dim: int = 1024
    max_position_embeddings: int = 8000
    rope_theta: float = 10000.0
    num_attention_heads: int = 16
    num_key_value_heads: int = 16
    sliding_window: int = 72
    intermediate_size: int = 3072
    rms_norm_eps: float = 1e-5
    num_hidden_layers: int = 8
    num_quantizers: int = 16
    upsample_rates: tuple = (8, 5, 4, 3)
    upsampling_ratios: tuple = (2, 2)
    decoder_dim: int = 1536
    output_sample_rate: int = 24000

    @classmethod
    def from_dict(cls, d: dict) -> "Code2WavConfig":
        names = {f.name for f in fields(cls)}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names}
        return cls(**kw)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def total_upsample(self) -> int:
        return math.prod(self.upsample_rates) * math.prod(self.upsampling_ratios)

    def check(self) -> None:
        chans = [self.codebook_dim, self.hidden_size, self.latent_dim, self.intermediate_size] + \
                [self.decoder_dim // 2 ** i for i in range(len(self.upsample_rates) + 1)]
        bad = [c for c in chans if c % 32]
        if bad:
            raise ValueError(f"Code2Wav on MI355X needs channel counts that are multiples of 32 (the MFMA k-step): {bad}")
        if self.head_dim not in (64, 128) or self.latent_dim > 1024 or self.num_attention_heads % self.num_key_value_heads:
            raise ValueError("Code2Wav on MI355X: head_dim 64 | 128, latent_dim <= 1024, q heads a multiple of kv heads")


def _split(a: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """fp32 -> (hi, lo) bf16 with hi + lo = a to 2^-17 relative: the two operands of a split-bf16 product."""
    hi = a.to(BF16)
    return hi, (a - hi.float()).to(BF16)


class _Conv:
    """One omni_gemm_tile launch: weights fragment-major bf16 [N, taps * C_in], fp32 bias / scale.  With `w_lo` (the bf16 residue
    of the fp32 weights, operand_precision="fp32") `f32()` computes the product on fp32-precision operands as three launches:
    hi.hi + hi.lo + lo.hi accumulated in fp32 (the lo.lo term is 2^-34 of the product)."""
    __slots__ = ("w", "w_lo", "bias", "scale", "taps", "dilation", "n")

    def __init__(self, w, bias, taps=1, dilation=1, scale=None, w_lo=None):
        self.w, self.bias, self.scale, self.taps, self.dilation, self.n, self.w_lo = w, bias, scale, taps, dilation, w.shape[0], w_lo

    def __call__(self, x, **kw):
        return ops.gemm_tile(x, self.w, bias=self.bias, scale=self.scale, taps=self.taps, dilation=self.dilation, **kw)

    def f32(self, x: torch.Tensor, resid: torch.Tensor | None = None) -> torch.Tensor:
        """x fp32 [rows, C_in] -> fp32 (x . W^T + bias) * scale (+ resid), operands at fp32 precision (split-bf16 on the MFMA tile kernel)."""
        hi, lo = _split(x)
        kw = dict(scale=self.scale, taps=self.taps, dilation=self.dilation, want="f")
        y = ops.gemm_tile(hi, self.w, bias=self.bias, resid=resid, **kw)
        ops.gemm_tile(hi, self.w_lo, resid=y, out_f32=y, **kw)
        ops.gemm_tile(lo, self.w, resid=y, out_f32=y, **kw)
        return y


class Code2WavDecoder:
    def __init__(self, cfg: Code2WavConfig | dict, state: dict[str, torch.Tensor], device: str = "cuda:0", fused_units: bool = True,
                 operand_precision: str = "bf16"):
        if not torch.cuda.is_available():
            raise L.OmniError("Code2WavDecoder needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
        self.lib = L.load()
        self.cfg = cfg if isinstance(cfg, Code2WavConfig) else Code2WavConfig.from_dict(cfg)
        self.cfg.check()
        self.device = torch.device(device)
        self.total_upsample = self.cfg.total_upsample
        self._graph = None
        # residual units of the 96- / 192-channel blocks as one launch each (omni_codec_res_unit) instead of two omni_gemm_tile launches
        self.fused_units = bool(fused_units)
        # "bf16" (default): GEMM operands bf16, fp32 accumulate and fp32 residual streams -- the fast path.  "fp32": the
        # reference's arithmetic (it loads the decoder in fp32, qwen3_tts_code2wav.py:71-75): every GEMM on fp32-precision
        # operands (split-bf16, three MFMA launches), every activation between them kept in fp32 (forward_fp32)
        if operand_precision not in ("bf16", "fp32"):
            raise ValueError(f"operand_precision={operand_precision!r}")
        self.operand_precision = operand_precision
        self._build(state)

    # ------------------------------------------------------------------ weights
    def _build(self, sd: dict[str, torch.Tensor]) -> None:
        c, dev = self.cfg, self.device

        def f32(name):
            return sd[name].detach().to(dev, torch.float32).contiguous()

        fp32_mode = self.operand_precision == "fp32"

        def frag(w2d):                      # [N, K] fp32 -> bf16 fragment-major on the device
            return frag_shuffle(w2d.to(dev, torch.float32).to(BF16).contiguous())

        def frag_lo(w2d, shuffle=frag_shuffle):     # the bf16 residue of the fp32 weights, same layout (fp32 mode only)
            if not fp32_mode:
                return None
            w32 = w2d.to(dev, torch.float32)
            return shuffle((w32 - w32.to(BF16).float()).to(BF16).contiguous())

        def linear(name, bias=True, scale=None):
            return _Conv(frag(sd[name + ".weight"]), f32(name + ".bias") if bias else None, scale=scale, w_lo=frag_lo(sd[name + ".weight"]))

        def conv(name, dilation=1):         # Conv1d weight [Cout, Cin, k] -> K index = tap * Cin + ci
            w = sd[name + ".conv.weight"].detach().float()
            co, ci, k = w.shape
            wk = w.permute(0, 2, 1).reshape(co, k * ci)
            return _Conv(frag(wk), f32(name + ".conv.bias"), taps=k, dilation=dilation, w_lo=frag_lo(wk))

        def tconv(name, stride):            # ConvTranspose1d weight [Cin, Cout, k], k = stride | 2 * stride -> rows n = r * Cout + co
            w = sd[name + ".conv.weight"].detach().float()
            ci, co, k = w.shape
            if k == stride:
                wk, taps = w, 1
            elif k == 2 * stride:           # y[s i + r] = x[i] w[:, :, r] + x[i - 1] w[:, :, r + s]: K = [x[i - 1] | x[i]]
                wk, taps = torch.cat([w[:, :, stride:], w[:, :, :stride]], dim=0), 2
            else:
                raise ValueError(f"{name}: transposed conv kernel {k} with stride {stride} (k = s or 2 s)")
            wk = wk.permute(2, 1, 0).reshape(stride * co, taps * ci)
            return _Conv(frag(wk), f32(name + ".conv.bias").repeat(stride), taps=taps, w_lo=frag_lo(wk))

        def snake(name):
            a, b = sd[name + ".alpha"].detach().float(), sd[name + ".beta"].detach().float()
            return (torch.exp(a).to(dev).contiguous(), (1.0 / (torch.exp(b) + 1e-9)).to(dev).contiguous())

        # residual VQ: per quantizer, (embedding_sum / clamp(usage)) . output_proj^T of its group
        tabs = []
        for q in range(c.num_quantizers):
            part, i = ("rvq_first", 0) if q == 0 else ("rvq_rest", q - 1)
            p = f"quantizer.{part}.vq.layers.{i}._codebook."
            emb = sd[p + "embedding_sum"].double() / sd[p + "cluster_usage"].double().clamp(min=1e-5)[:, None]
            proj = sd[f"quantizer.{part}.output_proj.weight"].double()[:, :, 0]              # [cd, cd / 2]
            tabs.append((emb @ proj.T).float())
        self.rvq_table = torch.stack(tabs).to(dev).contiguous()                              # [Q, bins, cd]
        self.pre_conv = conv("pre_conv")
        # transformer
        self.in_proj, self.out_proj = linear("pre_transformer.input_proj"), linear("pre_transformer.output_proj")
        self.final_norm = f32("pre_transformer.norm.weight")
        self.layers = []
        for l in range(c.num_hidden_layers):
            p = f"pre_transformer.layers.{l}."
            qkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"].detach().float() for n in "qkv"], 0)
            gu = torch.cat([sd[p + "mlp.gate_proj.weight"].detach().float(), sd[p + "mlp.up_proj.weight"].detach().float()], 0)
            self.layers.append(dict(
                ln1=f32(p + "input_layernorm.weight"), ln2=f32(p + "post_attention_layernorm.weight"),
                qkv=_Conv(frag(qkv), None, w_lo=frag_lo(qkv)),
                o=_Conv(frag(sd[p + "self_attn.o_proj.weight"]), None, scale=f32(p + "self_attn_layer_scale.scale"),
                        w_lo=frag_lo(sd[p + "self_attn.o_proj.weight"])),
                gu=_Conv(gu8_shuffle(gu.to(dev).to(BF16)), None, w_lo=frag_lo(gu, gu8_shuffle)),
                down=_Conv(frag(sd[p + "mlp.down_proj.weight"]), None, scale=f32(p + "mlp_layer_scale.scale"),
                           w_lo=frag_lo(sd[p + "mlp.down_proj.weight"]))))
        # upsample stages: transposed conv (k = s) + ConvNeXt
        self.ups = []
        for i, f in enumerate(c.upsampling_ratios):
            p = f"upsample.{i}.1."
            dw = sd[p + "dwconv.conv.weight"].detach().float()                                # [C, 1, 7]
            self.ups.append(dict(
                f=f, tc=tconv(f"upsample.{i}.0", f), dw_w=dw[:, 0].to(dev).contiguous(), dw_b=f32(p + "dwconv.conv.bias"),
                dw_taps=dw.shape[-1], ln_w=f32(p + "norm.weight"), ln_b=f32(p + "norm.bias"),
                pw1=linear(p + "pwconv1"), pw2=linear(p + "pwconv2", scale=f32(p + "gamma"))))
        # decoder: conv7, blocks (snake, transposed conv, 3 residual units), snake, conv7 -> 1
        self.dec0 = conv("decoder.0")
        self.blocks = []
        for i, r in enumerate(c.upsample_rates):
            p = f"decoder.{i + 1}.block."
            units = [dict(act1=snake(f"{p}{u + 2}.act1"), conv1=conv(f"{p}{u + 2}.conv1", dilation=d), act2=snake(f"{p}{u + 2}.act2"),
                          conv2=conv(f"{p}{u + 2}.conv2")) for u, d in enumerate((1, 3, 9))]
            self.blocks.append(dict(r=r, act=snake(p + "0"), tc=tconv(p + "1", r), units=units))
        n = len(c.upsample_rates)
        self.last_act = snake(f"decoder.{n + 1}")
        wl = sd[f"decoder.{n + 2}.conv.weight"].detach().float()                              # [1, C, taps]
        self.out_w = wl[0].T.to(dev).contiguous()                                             # [taps, C]
        if fp32_mode:       # the depthwise conv of the ConvNeXt blocks as torch weights for forward_fp32
            for i, u in enumerate(self.ups):
                u["dw_w3"] = sd[f"upsample.{i}.1.dwconv.conv.weight"].detach().to(dev, torch.float32).contiguous()
        self.out_taps, self.out_c = wl.shape[2], wl.shape[1]
        self.out_b = float(sd[f"decoder.{n + 2}.conv.bias"].detach().float().reshape(-1)[0])

    # ------------------------------------------------------------------ module-like surface
    def eval(self):
        return self

    def precompute_snake_caches(self) -> None:
        """exp(alpha) / 1 / (exp(beta) + eps) are computed once at load here (…:991-1000 does it on request)."""

    def __call__(self, codes: torch.Tensor) -> torch.Tensor:
        return self.forward(codes)

    # ------------------------------------------------------------------ forward
    def _check(self, code: int, what: str) -> None:
        L.check(code, what)

    @torch.no_grad()
    def forward(self, codes: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
        """codes long [1, Q, T] (or [Q, T]) on the device -> fp32 [1, 1, T * total_upsample] in [-1, 1]  (…:1009-1027)."""
        c, lib, st = self.cfg, self.lib, L.current_stream()
        if codes.dim() == 3:
            if codes.shape[0] != 1:
                return torch.cat([self.forward(codes[b:b + 1], taps) for b in range(codes.shape[0])], 0)
            codes = codes[0]
        if self.operand_precision == "fp32":
            return self.forward_fp32(codes, taps)
        if codes.shape[0] != c.num_quantizers:
            raise ValueError(f"Expected {c.num_quantizers} layer of codes, got {codes.shape[0]}")
        codes = codes.to(self.device, torch.long)
        if codes.stride(1) != 1:
            codes = codes.contiguous()
        T, dev = codes.shape[1], self.device

        def tap(k, v):
            if taps is not None:
                taps[k] = v
        # ---- quantizer + pre_conv
        q = torch.empty(T, c.codebook_dim, dtype=BF16, device=dev)
        self._check(lib.omni_codec_rvq_embed(codes.data_ptr(), codes.stride(0), self.rvq_table.data_ptr(), q.data_ptr(), T,
                                             c.num_quantizers, c.codebook_size, c.codebook_dim, st), "omni_codec_rvq_embed")
        tap("quantized", q)
        x = self.pre_conv(q)
        tap("pre_conv", x)
        # ---- transformer: fp32 stream h, bf16 operands
        H, nh, nkv, hd = c.hidden_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim
        h = self.in_proj(x, want="f")
        a = torch.empty(T, H, dtype=BF16, device=dev)
        attn = torch.empty(T, nh * hd, dtype=BF16, device=dev)
        for lw in self.layers:
            self._check(lib.omni_codec_rmsnorm(h.data_ptr(), H, lw["ln1"].data_ptr(), c.rms_norm_eps, a.data_ptr(), T, H, st), "omni_codec_rmsnorm")
            qkv = lw["qkv"](a)
            self._check(lib.omni_codec_rope(qkv.data_ptr(), qkv.stride(0), T, nh, nkv, hd, float(c.rope_theta), st), "omni_codec_rope")
            self._check(lib.omni_codec_window_attn(qkv.data_ptr(), qkv.stride(0), attn.data_ptr(), attn.stride(0), T, nh, nkv, hd,
                                                   c.sliding_window, hd ** -0.5, st), "omni_codec_window_attn")
            lw["o"](attn, resid=h, out_f32=h, want="f")
            self._check(lib.omni_codec_rmsnorm(h.data_ptr(), H, lw["ln2"].data_ptr(), c.rms_norm_eps, a.data_ptr(), T, H, st), "omni_codec_rmsnorm")
            act = lw["gu"](a, act=L.TILE_ACT_SILU_MUL_GU8)
            lw["down"](act, resid=h, out_f32=h, want="f")
        self._check(lib.omni_codec_rmsnorm(h.data_ptr(), H, self.final_norm.data_ptr(), c.rms_norm_eps, a.data_ptr(), T, H, st), "omni_codec_rmsnorm")
        x = self.out_proj(a)
        tap("pre_transformer", x)
        # ---- upsample stages
        Lt = c.latent_dim
        for u in self.ups:
            T = T * u["f"]
            xs = u["tc"](x, want="f").view(T, Lt)                        # [T, f * C] row-major IS the up-sampled [T * f, C]
            y = torch.empty(T, Lt, dtype=BF16, device=dev)
            self._check(lib.omni_codec_dwconv_ln(xs.data_ptr(), Lt, u["dw_w"].data_ptr(), u["dw_b"].data_ptr(), u["ln_w"].data_ptr(),
                                                 u["ln_b"].data_ptr(), 1e-6, y.data_ptr(), T, Lt, u["dw_taps"], st), "omni_codec_dwconv_ln")
            y = u["pw1"](y, act=L.TILE_ACT_GELU)
            x = u["pw2"](y, resid=xs)
        tap("upsampled", x)
        # ---- decoder: the snake in front of every conv is the producer's second output
        if taps is None:
            s = self.dec0(x, snake=self.blocks[0]["act"], want="s")
        else:
            f, s = self.dec0(x, snake=self.blocks[0]["act"], want="fs")
            tap("decoder0", f)
        nb = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            units = blk["units"]
            T = T * blk["r"]
            cout = blk["tc"].n // blk["r"]
            hstream, s = blk["tc"](s, snake=self._rep(units[0]["act1"], blk["r"]), want="fs")
            hstream, s = hstream.view(T, cout), s.view(T, cout)
            s_alt = None
            for ui, un in enumerate(units):
                nxt = units[ui + 1]["act1"] if ui + 1 < len(units) else (self.blocks[bi + 1]["act"] if bi + 1 < nb else self.last_act)
                c1 = un["conv1"]
                if self.fused_units and lib.omni_codec_res_unit_supported(cout, c1.taps, c1.dilation):
                    s_alt = torch.empty_like(s) if s_alt is None else s_alt           # the unit reads a halo of s: no in-place s
                    ru = L.ResUnit()
                    ru.s, ru.h, ru.s_next = s.data_ptr(), hstream.data_ptr(), s_alt.data_ptr()
                    ru.w1, ru.b1, ru.snake2_alpha, ru.snake2_inv_beta = c1.w.data_ptr(), c1.bias.data_ptr(), un["act2"][0].data_ptr(), un["act2"][1].data_ptr()
                    ru.w2, ru.b2, ru.next_alpha, ru.next_inv_beta = un["conv2"].w.data_ptr(), un["conv2"].bias.data_ptr(), nxt[0].data_ptr(), nxt[1].data_ptr()
                    ru.T, ru.C, ru.dilation = T, cout, c1.dilation
                    self._check(lib.omni_codec_res_unit(C.byref(ru), st), "omni_codec_res_unit")
                    s, s_alt = s_alt, s
                else:
                    t1 = c1(s, snake=un["act2"], want="s")
                    _, s = un["conv2"](t1, resid=hstream, out_f32=hstream, snake=nxt, want="fs")
            tap(f"decoder{bi + 1}", hstream)
        tap(f"decoder{nb + 1}", s)
        wav = torch.empty(T, dtype=torch.float32, device=dev)
        self._check(lib.omni_codec_out_conv(s.data_ptr(), self.out_w.data_ptr(), self.out_b, wav.data_ptr(), T, self.out_c, self.out_taps, st),
                    "omni_codec_out_conv")
        return wav.view(1, 1, T)

    def forward_fp32(self, codes: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
        """The decoder in the reference's arithmetic (fp32 module, qwen3_tts_code2wav.py:71-75): every conv / linear through
        `_Conv.f32` (fp32-precision operands on the MFMA tile kernel, fp32 accumulate), everything between them -- RVQ sum,
        RMSNorm, RoPE, sliding-window attention, depthwise conv + LayerNorm, GELU / SiLU, SnakeBeta, the final conv -- as fp32
        elementwise / reduction work on the device.  Several times slower than the bf16-operand path; selected by
        operand_precision="fp32" when a stage must match the reference to fp32 rounding.  codes long [Q, T]."""
        import torch.nn.functional as F
        c, dev = self.cfg, self.device
        if codes.shape[0] != c.num_quantizers:
            raise ValueError(f"Expected {c.num_quantizers} layer of codes, got {codes.shape[0]}")
        codes = codes.to(dev, torch.long)
        T = codes.shape[1]

        def tap(k, v):
            if taps is not None:
                taps[k] = v

        def snake(y, sn):
            return y + sn[1] * torch.sin(y * sn[0]) ** 2

        def rms(h, w_):
            return h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + c.rms_norm_eps) * w_

        q = self.rvq_table[torch.arange(c.num_quantizers, device=dev)[:, None], codes].sum(0)        # [T, cd] fp32
        tap("quantized", q)
        x = self.pre_conv.f32(q)
        tap("pre_conv", x)
        # ---- transformer
        H, nh, nkv, hd = c.hidden_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim
        h = self.in_proj.f32(x)
        pos = torch.arange(T, device=dev, dtype=torch.float32)
        inv = 1.0 / (float(c.rope_theta) ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))
        ang = pos[:, None] * inv[None, :]
        cos, sin = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :], torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
        rot = lambda v: torch.cat([-v[..., hd // 2:], v[..., :hd // 2]], -1)
        ti = torch.arange(T, device=dev)
        keep = (ti[None, :] <= ti[:, None]) & (ti[:, None] - ti[None, :] < c.sliding_window)         # causal, `sliding_window` keys
        for lw in self.layers:
            qkv = lw["qkv"].f32(rms(h, lw["ln1"]))
            qh = qkv[:, :nh * hd].view(T, nh, hd)
            kh = qkv[:, nh * hd:(nh + nkv) * hd].view(T, nkv, hd)
            vh = qkv[:, (nh + nkv) * hd:].view(T, nkv, hd)
            qh, kh = qh * cos + rot(qh) * sin, kh * cos + rot(kh) * sin
            rep = nh // nkv
            kh, vh = kh.repeat_interleave(rep, 1), vh.repeat_interleave(rep, 1)
            sc = torch.einsum("thd,shd->hts", qh, kh) * hd ** -0.5
            sc = sc.masked_fill(~keep[None], float("-inf")).softmax(-1)
            attn = torch.einsum("hts,shd->thd", sc, vh).reshape(T, nh * hd)
            h = lw["o"].f32(attn, resid=h)
            gu = lw["gu"].f32(rms(h, lw["ln2"]))                         # rows interleaved [8 gate | 8 up] (gu8 layout)
            gu = gu.view(T, -1, 2, 8)
            h = lw["down"].f32((F.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(T, -1), resid=h)
        x = self.out_proj.f32(rms(h, self.final_norm))
        tap("pre_transformer", x)
        # ---- upsample stages
        Lt = c.latent_dim
        for u in self.ups:
            T = T * u["f"]
            xs = u["tc"].f32(x).view(T, Lt)
            k = u["dw_taps"]
            xp = F.pad(xs, (0, 0, k - 1, 0))                                                        # causal depthwise conv: k shifted
            y = sum(xp[j:j + T] * u["dw_w3"][:, 0, j] for j in range(k)) + u["dw_b"]                # elementwise multiply-adds
            y = F.layer_norm(y, (Lt,), u["ln_w"], u["ln_b"], 1e-6)
            y = F.gelu(u["pw1"].f32(y))
            x = u["pw2"].f32(y, resid=xs)
        tap("upsampled", x)
        # ---- decoder
        f = self.dec0.f32(x)
        tap("decoder0", f)
        nb = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            T = T * blk["r"]
            cout = blk["tc"].n // blk["r"]
            hs = blk["tc"].f32(snake(f, blk["act"])).view(T, cout)
            for un in blk["units"]:
                t1 = un["conv1"].f32(snake(hs, un["act1"]))
                hs = un["conv2"].f32(snake(t1, un["act2"]), resid=hs)
            tap(f"decoder{bi + 1}", hs)
            f = hs
        s = snake(f, self.last_act)
        tap(f"decoder{nb + 1}", s)
        sp = F.pad(s, (0, 0, self.out_taps - 1, 0))                                                # causal left padding in time
        wav = sum((sp[j:j + T] * self.out_w[j]).sum(-1) for j in range(self.out_taps)) + self.out_b
        return wav.clamp(-1.0, 1.0).view(1, 1, T)

    def _rep(self, sn, r: int):
        """Snake parameters of a [T, r * C] transposed-conv output: channel n = phase * C + c."""
        key = (sn[0].data_ptr(), r)
        cache = self.__dict__.setdefault("_rep_cache", {})
        if key not in cache:
            cache[key] = (sn[0].repeat(r).contiguous(), sn[1].repeat(r).contiguous())
        return cache[key]

    # ------------------------------------------------------------------ chunking / graphs (…:1002-1043)
    def enable_cudagraph(self, capture_sizes: list[int] | None = None, device=None, codec_chunk_frames: int = 0,
                         codec_left_context_frames: int = 0) -> None:
        from .graph_decoder import HipGraphDecoderWrapper
        self._graph = HipGraphDecoderWrapper(self, capture_sizes=capture_sizes, num_quantizers=self.cfg.num_quantizers, enabled=True)
        self._graph.warmup(self.device if device is None else device, dtype=torch.long, codec_chunk_frames=codec_chunk_frames,
                           codec_left_context_frames=codec_left_context_frames)
        logger.info("hipGraph enabled for the Code2Wav decoder: seq_lens=%s", self._graph.capture_sizes)

    enable_hipgraph = enable_cudagraph

    def disable_cudagraph(self) -> None:
        self._graph = None

    def chunked_decode(self, codes: torch.Tensor, chunk_size: int = 300, left_context_size: int = 25) -> torch.Tensor:
        if self._graph is not None:
            return self._graph.chunked_decode_with_cudagraph(codes, chunk_size, left_context_size)
        from .graph_decoder import HipGraphDecoderWrapper
        up = self.total_upsample
        return torch.cat([self.forward(codes[..., lo:hi])[..., ctx * up:]
                          for lo, hi, ctx in HipGraphDecoderWrapper.chunk_windows(codes.shape[-1], chunk_size, left_context_size)], dim=-1)

    def flops(self, T: int) -> float:
        """Multiply-add FLOPs (x2) of one forward over T code frames: the GEMM-shaped work only (bench / roofline)."""
        c = self.cfg
        tot = 2.0 * T * 3 * c.codebook_dim * c.latent_dim + 2.0 * T * 2 * c.latent_dim * c.hidden_size
        nh, nkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        tot += c.num_hidden_layers * 2.0 * T * c.hidden_size * ((nh + 2 * nkv) * hd + nh * hd + 3 * c.intermediate_size)
        t = T
        for f in c.upsampling_ratios:
            tot += 2.0 * t * c.latent_dim * c.latent_dim * f
            t *= f
            tot += 2.0 * t * c.latent_dim * 4 * c.latent_dim * 2
        tot += 2.0 * t * 7 * c.latent_dim * c.decoder_dim
        for i, r in enumerate(c.upsample_rates):
            cin, cout = c.decoder_dim // 2 ** i, c.decoder_dim // 2 ** (i + 1)
            tot += 2.0 * t * 2 * cin * r * cout
            t *= r
            tot += 3 * 2.0 * t * cout * cout * 8
        return tot + 2.0 * t * 7 * (c.decoder_dim // 2 ** len(c.upsample_rates))


class MI355XCode2Wav:
    """Stage-1 generation model (Qwen3TTSCode2Wav, qwen3_tts_code2wav.py:21-334): consumes frame-aligned codec tokens from
    `input_ids` and returns one waveform per request.  `decoder` is a Code2WavDecoder (or anything with its surface)."""
    input_modalities = "audio"
    have_multimodal_outputs = True
    requires_raw_input_tokens = True

    def __init__(self, decoder, *, output_sample_rate: int = 24000, validate_codes: bool = True):
        self.decoder = decoder
        self.num_quantizers = int(decoder.cfg.num_quantizers)
        self.total_upsample = int(decoder.total_upsample)
        self.output_sample_rate = int(output_sample_rate)
        self.validate_codes = validate_codes

    @staticmethod
    def split_request_ids(ids: torch.Tensor, seq_token_counts: list[int] | None = None) -> list[torch.Tensor]:
        """Concatenated input_ids -> per-request segments (…:150-170; the micro-batch slices of vLLM's forward context do not
        exist on this runner: the runner passes seq_token_counts)."""
        if seq_token_counts is not None and len(seq_token_counts) > 1:
            out, lo, n = [], 0, ids.numel()
            for cnt in seq_token_counts:
                out.append(ids[lo:min(lo + cnt, n)])
                lo += cnt
            return out
        return [ids]

    @staticmethod
    def _ctx_value(info: dict) -> int:
        v = info.get("left_context_size", 0)          # an int, [int] or tensor([int]) after serialisation (…:222-230)
        if isinstance(v, list):
            v = v[0] if v else 0
        if isinstance(v, torch.Tensor):
            v = v.reshape(-1)[0].item() if v.numel() > 0 else 0
        return int(v)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor | None = None, positions=None, intermediate_tensors=None, inputs_embeds=None,
                runtime_additional_information: list[dict] | None = None, **kwargs) -> dict:
        """input_ids per request: flat codes, codebook-major [Q * F].  Returns {"model_outputs": [wav per request], "sr": [...]}
        (the multimodal_outputs of the reference's OmniOutput).  Requests whose length is not a multiple of Q are skipped with
        an empty waveform; a request's first `left_context_size` frames are context: their share of the samples is dropped."""
        q, up = self.num_quantizers, self.total_upsample
        sr = torch.tensor(self.output_sample_rate, dtype=torch.int32)
        empty = torch.zeros((0,), dtype=torch.float32)
        if input_ids is None or input_ids.numel() == 0:
            return {"model_outputs": [empty], "sr": [sr]}
        ids = input_ids.reshape(-1).to(dtype=torch.long)
        reqs = self.split_request_ids(ids, kwargs.get("seq_token_counts"))
        ctx = [0] * len(reqs)
        for i, info in enumerate(runtime_additional_information or []):
            if i < len(ctx) and "left_context_size" in info:
                ctx[i] = self._ctx_value(info)
        audios, srs = [empty] * len(reqs), [sr] * len(reqs)
        for i, r in enumerate(reqs):
            n = r.numel()
            if n == 0 or n % q != 0:
                if n > 0:
                    logger.warning("Code2Wav input_ids length %d not divisible by num_quantizers %d; skipping malformed request.", n, q)
                continue
            frames = n // q
            codes = r.reshape(q, frames)
            if self.validate_codes:
                lo, hi = int(codes.min()), int(codes.max())
                if lo < 0 or hi >= self.decoder.cfg.codebook_size:
                    raise ValueError(f"Code2Wav request {i}: codec ids in [{lo}, {hi}] outside [0, {self.decoder.cfg.codebook_size})")
            wav = self.decoder.chunked_decode(codes.unsqueeze(0)).reshape(-1)
            if ctx[i] <= 0:
                wav = wav[: frames * up]
            else:
                cut = int(ctx[i] / max(frames, 1) * wav.shape[0])
                if cut >= wav.shape[0]:
                    logger.warning("Context trim %d >= decoded length %d; returning empty audio.", cut, wav.shape[0])
                    continue
                wav = wav[cut:]
            if wav.shape[0] > 0:
                audios[i] = wav.to(torch.float32).reshape(-1)
        return {"model_outputs": audios, "sr": srs}

    __call__ = forward
