"""CPU restatement of the talker AR decode path (TEST INFRASTRUCTURE, not product).

What it restates (R/ = /root/reference, V/ = R/vllm_omni):

* one engine step of the Qwen3-TTS talker as driven by
  ``GPUARModelRunner.execute_model`` / ``sample_tokens``
  (V/worker/gpu_ar_model_runner.py:93-400, 403-660) and
  ``OmniGPUModelRunner._preprocess`` / ``_talker_mtp_forward``
  (V/worker/gpu_model_runner.py:1084-1303);
* the talker model glue: ``compute_logits`` (V/model_executor/models/qwen3_tts/
  qwen3_tts_talker.py:424-443), decode-branch ``preprocess`` (615-647),
  ``postprocess`` (649-655), ``talker_mtp`` (1594-1642);
* the code predictor, re-prefill form, exactly as
  V/model_executor/models/qwen3_tts/qwen3_tts_code_predictor_vllm.py:36-226,480-561;
* the backbone decoder (vLLM ``Qwen3Model`` -- third-party ``vllm==0.18.0``, source
  absent from R/): restated with HuggingFace ``transformers`` Qwen3 numerics
  (fp32 RMSNorm cast back then times weight; q/k-norm over head_dim; neox RoPE with
  fp32 cos/sin cast to bf16; GQA attention fp32 softmax; SwiGLU) over a paged KV
  cache ``[2, num_blocks, block_size, n_kv_heads, head_dim]`` with
  ``slot = block_table[r][p // bs] * bs + p % bs``.

Also restated here (each pinned to the reference module / method named beside the function): the Omni talker's sparse-MoE
block (HF ``Qwen3OmniMoeTalkerTextSparseMoeBlock``, tests/golden/moe_block.npz), its prompt-embedding builder and decode-side
text steps (the reference's own ``_thinker_to_talker_prefill`` & co., qwen3_omni.py:650-1060, run on HF ``ResizeMLP`` modules:
tests/golden/omni_prompt_builder.pt), SnakeBeta (the reference module, snake_beta.npz), KV extraction (kv_extract.npz).

Pinning status: the code predictor is pinned against the reference's own file
(tests/golden/code_predictor_*.npz, minted by tests/golden/make_fixtures.py); the
backbone is pinned against HF ``Qwen3Model`` (tests/golden/qwen3_backbone_*.npz).
vLLM-internal semantics (fp8 scale convention, sampler RNG) are stated here
explicitly and are themselves the oracle: **parity with vLLM 0.18.0 is unpinned**
(no source, no golden vectors in the reference -- SURVEY F2/F5).

Numeric conventions (the GPU path follows the same rounding points):
  * activations are bf16 between ops; every GEMM accumulates in fp32 and rounds
    its output to bf16 once (products of bf16 are exact in fp32, so only the
    summation order differs between implementations);
  * RMSNorm: fp32, ``w * bf16(x * rsqrt(mean(x^2) + eps))`` (the product rounds
    to bf16 again);
  * residual adds in bf16;
  * attention: q, k, v bf16 (k, v pass through the cache dtype), scores / softmax /
    PV in fp32, output rounded to bf16;
  * fp8 KV: OCP e4m3fn, ``q = sat(x / scale)`` (clamp to +-448, round-nearest-even),
    read ``float(q) * scale``; per-layer scalar k_scale / v_scale, default 1.0;
    ``calculate_kv_scales`` (the reference forces its FIRST forward eager for it, V/worker/gpu_ar_model_runner.py:122,269-275;
    the arithmetic is vLLM's attention layer, source absent -- SURVEY Appendix A): the first backbone pass sets, per layer,
    ``k_scale = max|k| / 200`` and ``v_scale = max|v| / 100`` over that pass's tokens (k after q/k-norm and RoPE; an all-zero
    tensor keeps scale 1), BEFORE the pass's own cache write; the scales stay fixed afterwards;
  * int8 KV (no reference semantics): per (token, kv-head) symmetric,
    ``scale = max(absmax, 1e-8) / 127`` held in fp32, ``q = rne(x / scale)``;
  * logits: fp32 accumulate, rounded to bf16 (vLLM ``LogitsProcessor`` returns the
    model dtype), then widened to fp32 for masking / sampling.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16
FP8_MAX = 448.0


# --------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------
def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(BF16)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """HF Qwen3RMSNorm / reference _RMSNorm
    (qwen3_tts_code_predictor_vllm.py:47-52)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None) -> torch.Tensor:
    """bf16 GEMM: fp32 accumulate, (+ bias in fp32), one rounding to bf16."""
    y = x.to(torch.float32) @ _f32(w).t()
    if b is not None:
        y = y + b.to(torch.float32)
    return y.to(BF16)


_F32_CACHE: dict[int, tuple[torch.Tensor, torch.Tensor]] = {}


def _f32(w: torch.Tensor) -> torch.Tensor:
    """fp32 view of a bf16 weight (exact), cached so the sgemm path is used."""
    if w.dtype == torch.float32:
        return w
    key = id(w)
    hit = _F32_CACHE.get(key)
    if hit is not None and hit[0] is w:
        return hit[1]
    wf = w.to(torch.float32)
    _F32_CACHE[key] = (w, wf)
    return wf


def clear_weight_cache() -> None:
    _F32_CACHE.clear()


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float) -> tuple[torch.Tensor, torch.Tensor]:
    """fp32 cos/sin cast to bf16, shape [T, head_dim]
    (reference _RotaryEmbedding.forward, code_predictor_vllm.py:80-93)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.to(torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16), emb.sin().to(BF16)


def mrope_cos_sin(positions3: torch.Tensor, head_dim: int, theta: float, section, interleaved: bool):
    """M-RoPE cos / sin [T, head_dim] from ids [3, T] (temporal / height / width).  The reference runs vLLM's MRotaryEmbedding
    (third party, vllm == 0.18.0, absent: `vllm/model_executor/layers/rotary_embedding/mrope.py` forward_native +
    `apply_interleaved_rope`); its published algorithm restated: cos_sin = cache[positions] per axis; chunked layout = the
    half-dim split by `mrope_section`, chunk i taken from axis i; interleaved layout (Qwen3-Omni) = axis 0 everywhere except
    pairs 1, 4, ... < 3 * section[1] (axis 1) and 2, 5, ... < 3 * section[2] (axis 2); the half is then duplicated (neox)."""
    half = head_dim // 2
    cs = [rope_cos_sin(positions3[i], head_dim, theta) for i in range(3)]
    cos3 = torch.stack([c[0][:, :half] for c in cs])
    sin3 = torch.stack([c[1][:, :half] for c in cs])
    if interleaved:
        cos, sin = cos3[0].clone(), sin3[0].clone()
        for ax in (1, 2):
            sl = slice(ax, int(section[ax]) * 3, 3)
            cos[:, sl] = cos3[ax][:, sl]
            sin[:, sl] = sin3[ax][:, sl]
    else:
        sec = [int(s) for s in section]
        cos = torch.cat([m[i] for i, m in enumerate(cos3.split(sec, dim=-1))], dim=-1)
        sin = torch.cat([m[i] for i, m in enumerate(sin3.split(sec, dim=-1))], dim=-1)
    return torch.cat((cos, cos), dim=-1), torch.cat((sin, sin), dim=-1)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [T, heads, D] bf16; cos/sin [T, D] bf16; bf16 arithmetic
    (code_predictor_vllm.py:162-163)."""
    return (x * cos[:, None, :]) + (rotate_half(x) * sin[:, None, :])


def silu_mul(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    return F.silu(g) * u


# --------------------------------------------------------------------------
# KV-cache quantisation
# --------------------------------------------------------------------------
def fp8_quant(x: torch.Tensor, scale: float) -> torch.Tensor:
    y = (x.to(torch.float32) / scale).clamp(-FP8_MAX, FP8_MAX)
    return y.to(torch.float8_e4m3fn)


def fp8_dequant(q: torch.Tensor, scale: float) -> torch.Tensor:
    return q.to(torch.float32) * scale


def int8_quant(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """x [..., D] -> (int8 [..., D], fp32 scale [...])."""
    xf = x.to(torch.float32)
    amax = xf.abs().amax(-1).clamp_min(1e-8)
    scale = amax / 127.0
    q = torch.round(xf / scale[..., None]).clamp(-127, 127).to(torch.int8)
    return q, scale


def int8_dequant(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.to(torch.float32) * scale[..., None]


# --------------------------------------------------------------------------
# paged KV cache
# --------------------------------------------------------------------------
def slot_of(block_table_row, pos: int, block_size: int) -> int:
    """slot = block_table[r][p // bs] * bs + p % bs."""
    return int(block_table_row[pos // block_size]) * block_size + pos % block_size


class PagedKV:
    """One layer's cache, layout [2, num_blocks, block_size, n_kv, D]
    (the stacked layout accepted by V/distributed/omni_connectors/utils/kv_utils.py:52-55)."""

    def __init__(self, num_blocks: int, block_size: int, n_kv: int, head_dim: int, kv_dtype: str,
                 k_scale: float = 1.0, v_scale: float = 1.0):
        self.kv_dtype = kv_dtype
        self.block_size = block_size
        self.k_scale, self.v_scale = float(k_scale), float(v_scale)
        shape = (2, num_blocks, block_size, n_kv, head_dim)
        if kv_dtype == "bf16":
            self.data = torch.zeros(shape, dtype=BF16)
        elif kv_dtype == "fp16":            # BASELINE config #2's wording: a half cache under a bf16 model (torch's bf16 -> half cast)
            self.data = torch.zeros(shape, dtype=torch.float16)
        elif kv_dtype == "fp8":
            self.data = torch.zeros(shape, dtype=torch.float8_e4m3fn)
        elif kv_dtype == "int8":
            self.data = torch.zeros(shape, dtype=torch.int8)
            self.scales = torch.zeros((2, num_blocks, block_size, n_kv), dtype=torch.float32)
        else:
            raise ValueError(kv_dtype)

    def write(self, slots: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> None:
        """k, v: [T, n_kv, D] bf16; slots [T] int64 (-1 = padded, skipped)."""
        nb, bs = self.data.shape[1], self.data.shape[2]
        flat = self.data.view(2, nb * bs, *self.data.shape[3:])
        keep = slots >= 0
        s = slots[keep]
        k, v = k[keep], v[keep]
        if self.kv_dtype == "bf16":
            flat[0, s] = k.to(BF16)
            flat[1, s] = v.to(BF16)
        elif self.kv_dtype == "fp16":
            flat[0, s] = k.to(BF16).to(torch.float16)
            flat[1, s] = v.to(BF16).to(torch.float16)
        elif self.kv_dtype == "fp8":
            flat[0, s] = fp8_quant(k, self.k_scale)
            flat[1, s] = fp8_quant(v, self.v_scale)
        else:
            sc = self.scales.view(2, nb * bs, -1)
            qk, sk = int8_quant(k)
            qv, sv = int8_quant(v)
            flat[0, s], sc[0, s] = qk, sk
            flat[1, s], sc[1, s] = qv, sv

    def gather(self, block_row, seq_len: int) -> tuple[torch.Tensor, torch.Tensor]:
        """Dequantised fp32 K, V [seq_len, n_kv, D] for one request."""
        bs = self.block_size
        nblk = (seq_len + bs - 1) // bs
        ids = torch.as_tensor(np.asarray(block_row[:nblk]), dtype=torch.long)
        k = self.data[0, ids].flatten(0, 1)[:seq_len]
        v = self.data[1, ids].flatten(0, 1)[:seq_len]
        if self.kv_dtype in ("bf16", "fp16"):
            return k.to(torch.float32), v.to(torch.float32)
        if self.kv_dtype == "fp8":
            return fp8_dequant(k, self.k_scale), fp8_dequant(v, self.v_scale)
        sk = self.scales[0, ids].flatten(0, 1)[:seq_len]
        sv = self.scales[1, ids].flatten(0, 1)[:seq_len]
        return int8_dequant(k, sk), int8_dequant(v, sv)


def attention_rows(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos: torch.Tensor, scale: float) -> torch.Tensor:
    """q [Tq, Hq, D] bf16 at absolute positions q_pos [Tq]; k, v fp32 [S, Hkv, D]
    (cache positions 0..S-1).  Causal: row i sees keys 0..q_pos[i].  fp32, out bf16."""
    hq, hkv = q.shape[1], k.shape[1]
    g = hq // hkv
    qf = q.to(torch.float32)
    kk = k.repeat_interleave(g, dim=1)  # [S, Hq, D]
    vv = v.repeat_interleave(g, dim=1)
    s = torch.einsum("thd,shd->hts", qf, kk) * scale
    key_pos = torch.arange(k.shape[0])
    mask = key_pos[None, :] > q_pos[:, None]
    s = s.masked_fill(mask[None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("hts,shd->thd", p, vv)
    return o.to(BF16)


# --------------------------------------------------------------------------
# sampler (vLLM-like order, RNG stated here: parity with vLLM unpinned)
# --------------------------------------------------------------------------
def hash_uniform(seed: int, step: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based uniform in (0,1): 24-bit mantissa from a murmur3-style mix of
    (seed, step, idx).  Mirrored bit-for-bit by the HIP sampler."""
    M = np.uint32
    x = (idx.astype(np.uint32) * M(0x9E3779B1)) ^ M(seed & 0xFFFFFFFF)
    x = x + M((step * 0x85EBCA77) & 0xFFFFFFFF)
    x ^= x >> M(16)
    x = x * M(0x85EBCA6B)
    x ^= x >> M(13)
    x = x * M(0xC2B2AE35)
    x ^= x >> M(16)
    return ((x >> M(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def top_p_filter(x: torch.Tensor, top_p: float) -> torch.Tensor:
    """Nucleus cut of qwen3_omni_moe_code_predictor_mtp.py:463-469 on a (top-k-masked) fp32 row: sort by
    (value desc, index asc), drop every entry whose PRECEDING cumulative softmax mass is >= top_p."""
    if not (0.0 < top_p < 1.0):
        return x
    order = sorted(range(x.numel()), key=lambda i: (-float(x[i]), i))
    xs = x[order].to(torch.float64)
    p = torch.softmax(xs, dim=-1)
    before = torch.cumsum(p, dim=-1) - p
    out = x.clone()
    drop = [order[j] for j in range(len(order)) if float(before[j]) >= top_p]
    if drop:
        out[torch.as_tensor(drop, dtype=torch.long)] = float("-inf")
    return out


def sample_row(logits: torch.Tensor, *, greedy: bool, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
               rep_penalty: float = 1.0, seen_ids=None, seed: int = 0, step: int = 0) -> int:
    """One row. Order: repetition penalty -> temperature -> top-k -> top-p -> softmax ->
    Gumbel-max, equivalent in distribution to vLLM's argmax(probs / Exp(1)) (SURVEY Appendix A
    'Sampler order'; vLLM's generator stream is not reproducible here).  Greedy = first argmax."""
    x = logits.to(torch.float32).clone()
    if rep_penalty != 1.0 and seen_ids is not None and len(seen_ids):
        ids = torch.as_tensor(sorted(set(int(i) for i in seen_ids if 0 <= int(i) < x.numel())), dtype=torch.long)
        sel = x[ids]
        x[ids] = torch.where(sel > 0, sel / rep_penalty, sel * rep_penalty)
    if greedy:
        return int(torch.argmax(x).item())
    x = x / temperature
    if top_k and top_k < x.numel():
        kth = torch.topk(x, top_k).values[-1]
        x = x.masked_fill(x < kth, float("-inf"))
    x = top_p_filter(x, top_p)
    # Gumbel-max: argmax(x + g), g = -log(-log u)  ==  argmax(softmax(x) / Exp(1)) in distribution
    xn = x.numpy().astype(np.float32)
    u = hash_uniform(seed, step, np.arange(xn.shape[0]))
    score = np.where(np.isfinite(xn), xn - np.log(-np.log(u)).astype(np.float32), -np.inf).astype(np.float32)
    return int(np.argmax(score))


def sample_row_margin(logits: torch.Tensor, **kw) -> float:
    """Gap between the best and second-best Gumbel score (tests skip near-ties: logf differs
    by ulps between numpy and the GPU)."""
    x = logits.to(torch.float32).clone()
    if kw.get("greedy"):
        v = torch.topk(x, 2).values
        return float(v[0] - v[1])
    x = x / kw.get("temperature", 1.0)
    top_k = kw.get("top_k", 0)
    if top_k and top_k < x.numel():
        kth = torch.topk(x, top_k).values[-1]
        x = x.masked_fill(x < kth, float("-inf"))
    x = top_p_filter(x, kw.get("top_p", 1.0))
    xn = x.numpy().astype(np.float32)
    u = hash_uniform(kw.get("seed", 0), kw.get("step", 0), np.arange(xn.shape[0]))
    score = np.where(np.isfinite(xn), xn - np.log(-np.log(u)).astype(np.float32), -np.inf)
    top = np.sort(score)[-2:]
    return float(top[1] - top[0])


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
@dataclass
class OracleState:
    """Per-request decode state (the reference keeps it in model_intermediate_buffer,
    gpu_model_runner.py:1330-1354)."""
    seq_len: int = 0
    last_id: int = 0
    last_hidden: torch.Tensor | None = None        # h[t] bf16 [H]
    tail_text: list = field(default_factory=list)  # queue of text-step vectors bf16 [H]
    tts_pad: torch.Tensor | None = None            # bf16 [H]
    out_ids: list = field(default_factory=list)
    prompt_len: int = 0
    rope_delta: int = 0                            # mrope_position_delta of the prompt: decode rotary position = index + delta


K_SCALE_CONSTANT, V_SCALE_CONSTANT = 200.0, 100.0       # vLLM envs.K_SCALE_CONSTANT / V_SCALE_CONSTANT


class TalkerOracle:
    def __init__(self, dims, weights: dict, kv_dtype: str = "bf16", num_blocks: int = 64,
                 block_size: int = 16, k_scale: float = 1.0, v_scale: float = 1.0, masked_logit: float = float("-inf"),
                 calculate_kv_scales: bool = False):
        self.d, self.w = dims, weights
        # fp8 KV: scales from the first backbone pass (module header); False = the constructor's constants
        self.calculate_kv_scales = bool(calculate_kv_scales) and kv_dtype == "fp8"
        # TTS: -inf (qwen3_tts_talker.py:424-443); the Omni talker suppresses with -1e9 (qwen3_omni.py:1143-1149)
        self.masked_logit = masked_logit
        self.block_size = block_size
        self.kv = [PagedKV(num_blocks, block_size, dims.kv_heads, dims.head_dim, kv_dtype, k_scale, v_scale)
                   for _ in range(dims.layers)]
        self.allowed = self.codec_allowed_mask(dims)

    # ---- constant logit mask (qwen3_tts_talker.py:386-394)
    @staticmethod
    def codec_allowed_mask(d) -> torch.Tensor:
        m = torch.zeros(d.vocab, dtype=torch.bool)
        lo, hi = 1, min(d.codebook, d.vocab)
        if hi > lo:
            m[lo:hi] = True
        if 0 <= d.eos_id < d.vocab:
            m[d.eos_id] = True
        return m

    # ---- backbone: tokens of several requests in one flat batch
    def backbone(self, x: torch.Tensor, positions: torch.Tensor, req_of_tok: list[int],
                 block_tables: list, seq_lens_after: list[int], rope_positions: torch.Tensor | None = None) -> torch.Tensor:
        """x [T, H] bf16 input embeddings; positions [T]; req_of_tok[t] = request row;
        block_tables[r] = block ids; seq_lens_after[r] = context length incl. this step.
        Writes K/V of every token at its slot, then attends through the cache.
        Returns final-normed hidden [T, H] bf16."""
        d, w = self.d, self.w
        T = x.shape[0]
        slots = torch.tensor([slot_of(block_tables[req_of_tok[t]], int(positions[t]), self.block_size)
                              for t in range(T)], dtype=torch.long)
        self.last_slots = slots
        # rope_positions: [T] (rotary position differs from the cache index by a delta) or [3, T] M-RoPE ids; default = positions
        if rope_positions is None:
            cos, sin = rope_cos_sin(positions, d.head_dim, d.rope_theta)
        elif rope_positions.ndim == 1:
            cos, sin = rope_cos_sin(rope_positions, d.head_dim, d.rope_theta)
        else:
            cos, sin = mrope_cos_sin(rope_positions, d.head_dim, d.rope_theta, d.mrope_section, d.mrope_interleaved)
        hq, hkv, D = d.q_heads, d.kv_heads, d.head_dim
        h = x
        for li in range(d.layers):
            p = f"l{li}."
            resid = h
            a = rms_norm(h, w[p + "ln1"], d.eps)
            qkv = linear(a, w[p + "wqkv"])
            q = qkv[:, : hq * D].reshape(T, hq, D)
            k = qkv[:, hq * D: (hq + hkv) * D].reshape(T, hkv, D)
            v = qkv[:, (hq + hkv) * D:].reshape(T, hkv, D)
            q = apply_rope(rms_norm(q, w[p + "qnorm"], d.eps), cos, sin)
            k = apply_rope(rms_norm(k, w[p + "knorm"], d.eps), cos, sin)
            trace = getattr(self, "trace", None)        # tests: per-layer intermediates (golden G2 pins them to HF Qwen3Model)
            if trace is not None:
                trace.append({"layer": li, "q": q.clone(), "k": k.clone(), "v": v.clone()})
            if self.calculate_kv_scales:
                ka, va = k.float().abs().max(), v.float().abs().max()          # fp32 tensors: the division below is an fp32 division
                self.kv[li].k_scale = float(ka / K_SCALE_CONSTANT) if float(ka) > 0 else 1.0
                self.kv[li].v_scale = float(va / V_SCALE_CONSTANT) if float(va) > 0 else 1.0
            self.kv[li].write(slots, k, v)
            o = torch.empty(T, hq, D, dtype=BF16)
            for r in sorted(set(req_of_tok)):
                idx = [t for t in range(T) if req_of_tok[t] == r]
                kk, vv = self.kv[li].gather(block_tables[r], seq_lens_after[r])
                o[idx] = attention_rows(q[idx], kk, vv, positions[idx], D ** -0.5)
            if trace is not None:
                trace[-1]["attn"] = o.reshape(T, hq * D).clone()
            h = resid + linear(o.reshape(T, hq * D), w[p + "wo"])
            resid = h
            a = rms_norm(h, w[p + "ln2"], d.eps)
            if getattr(d, "moe_experts", 0) > 0:       # Omni talker: sparse-MoE MLP
                mw = {"router": w[p + "moe_router"], "gate_up": w[p + "moe_gate_up"], "down": w[p + "moe_down"],
                      "shared_gate_up": w[p + "moe_shared_gate_up"], "shared_down": w[p + "moe_shared_down"],
                      "shared_gate": w[p + "moe_shared_gate"]}
                h = resid + moe_block(a, mw, d.moe_top_k, d.moe_norm_topk)
                continue
            gu = linear(a, w[p + "wgu"])
            act = silu_mul(gu[:, : d.inter], gu[:, d.inter:])
            h = resid + linear(act, w[p + "wdown"])
        self.calculate_kv_scales = False           # first pass only
        return rms_norm(h, w["norm"], d.eps)

    def compute_logits(self, hidden: torch.Tensor, round_bf16: bool = True) -> torch.Tensor:
        """lm_head + allowed-codec mask (qwen3_tts_talker.py:424-443). fp32 out."""
        y = hidden.to(torch.float32) @ _f32(self.w["lm_head"]).t()
        if round_bf16:
            y = y.to(BF16).to(torch.float32)
        return y.masked_fill(~self.allowed, self.masked_logit)

    # ---- code predictor, re-prefill form (code_predictor_vllm.py:480-561)
    def cp_model(self, buf: torch.Tensor) -> torch.Tensor:
        """buf [B, S, Hc] bf16 -> final-normed hidden [B, S, Hc]; full causal re-prefill,
        position ids 0..S-1 (code_predictor_vllm.py:263-273)."""
        d, w = self.d, self.w
        B, S, _ = buf.shape
        hq, hkv, D = d.cp_q_heads, d.cp_kv_heads, d.cp_head_dim
        pos = torch.arange(S)
        cos, sin = rope_cos_sin(pos, D, d.cp_rope_theta)
        h = buf
        for li in range(d.cp_layers):
            p = f"cp.l{li}."
            resid = h
            a = rms_norm(h, w[p + "ln1"], d.eps)
            qkv = linear(a.reshape(B * S, -1), w[p + "wqkv"]).reshape(B, S, -1)
            q = qkv[..., : hq * D].reshape(B, S, hq, D)
            k = qkv[..., hq * D: (hq + hkv) * D].reshape(B, S, hkv, D)
            v = qkv[..., (hq + hkv) * D:].reshape(B, S, hkv, D)
            q = rms_norm(q, w[p + "qnorm"], d.eps)
            k = rms_norm(k, w[p + "knorm"], d.eps)
            o = torch.empty(B, S, hq, D, dtype=BF16)
            for b in range(B):
                qb = apply_rope(q[b], cos, sin)
                kb = apply_rope(k[b], cos, sin)
                o[b] = attention_rows(qb, kb.to(torch.float32), v[b].to(torch.float32), pos, D ** -0.5)
            h = resid + linear(o.reshape(B * S, hq * D), w[p + "wo"]).reshape(B, S, -1)
            resid = h
            a = rms_norm(h, w[p + "ln2"], d.eps)
            gu = linear(a.reshape(B * S, -1), w[p + "wgu"])
            act = silu_mul(gu[:, : d.cp_inter], gu[:, d.cp_inter:])
            h = resid + linear(act, w[p + "wdown"]).reshape(B, S, -1)
        return rms_norm(h, w["cp.norm"], d.eps)

    def cp_project(self, x: torch.Tensor) -> torch.Tensor:
        """small_to_mtp_projection (code_predictor_vllm.py:340-343); identity when dims match."""
        if "cp.proj_w" not in self.w:
            return x
        return linear(x, self.w["cp.proj_w"], self.w["cp.proj_b"])

    def code_predictor(self, layer0_code: torch.Tensor, layer0_embed: torch.Tensor, last_hidden: torch.Tensor,
                       *, do_sample: bool = False, temperature: float = 0.9, top_k: int = 50, top_p: float = 1.0,
                       seed: int = 0, step: int = 0, return_logits: bool = False):
        """layer0_code [B] int64, layer0_embed [B,H] bf16, last_hidden [B,H] bf16 -> all_codes [B,Q].
        Sampling uses this oracle's hash RNG (the reference uses torch.multinomial on the
        global generator, code_predictor_vllm.py:545-551: not reproducible -> greedy for parity)."""
        d, w = self.d, self.w
        B, Q = layer0_code.shape[0], d.num_code_groups
        codes = torch.empty(B, Q, dtype=torch.long)
        codes[:, 0] = layer0_code
        buf = torch.zeros(B, Q + 1, d.cp_hidden, dtype=BF16)
        buf[:, 0] = self.cp_project(last_hidden)
        buf[:, 1] = self.cp_project(layer0_embed)
        all_logits = []
        for g in range(1, Q):
            hid = self.cp_model(buf)
            logits = linear(hid[:, g], w["cp.lm_head"][g - 1]).to(torch.float32)
            all_logits.append(logits)
            if do_sample and temperature > 0:
                st = step if hasattr(step, "__len__") else [step] * B
                sd = seed if hasattr(seed, "__len__") else [seed] * B      # per-request RNG keys (one generator per request)
                nxt = torch.tensor([sample_row(logits[b], greedy=False, temperature=max(temperature, 1e-6),
                                               top_k=top_k, top_p=top_p, seed=int(sd[b]), step=int(st[b]) * Q + g) for b in range(B)])
            else:
                nxt = logits.argmax(-1)
            codes[:, g] = nxt
            if g < Q - 1:
                buf[:, g + 1] = self.cp_project(w["cp.embed"][g - 1][nxt])
        # (test infrastructure: the group logits of the last call, for the near-tie check of a forked greedy frame)
        self.last_cp_logits = torch.stack(all_logits, 1) if all_logits else None
        if return_logits:
            return codes, self.last_cp_logits
        return codes

    def talker_mtp(self, input_ids: torch.Tensor, input_embeds: torch.Tensor, last_hidden: torch.Tensor,
                   text_step: torch.Tensor, **cp_kw):
        """qwen3_tts_talker.py:1594-1642 -> (inputs_embeds [B,H] bf16, audio_codes [B,Q] int64)."""
        d, w = self.d, self.w
        Q = d.num_code_groups
        codes = self.code_predictor(input_ids, input_embeds, last_hidden, **cp_kw)
        invalid0 = (codes[:, :1] < 0) | (codes[:, :1] >= d.codebook)
        codes = torch.where(invalid0.expand_as(codes), torch.zeros_like(codes), codes)
        embeds = [input_embeds[:, None, :]]
        for i in range(Q - 1):
            embeds.append(w["cp.embed"][i][codes[:, i + 1]][:, None, :])
        summed = torch.cat(embeds, dim=1).sum(1)          # bf16 sum: fp32 accumulate, one rounding
        return (summed + text_step), codes

    # ---- one engine step over a batch of decode requests (SURVEY 3.3 data-flow)
    def decode_step(self, states: list[OracleState], block_tables: list, *, greedy: bool = True,
                    sampling: dict | None = None, cp_kw: dict | None = None):
        """All requests have query_len 1. Returns (logits [B,V] fp32, sampled ids [B],
        hidden [B,H] bf16, audio_codes [B,Q], slots [B])."""
        d, w = self.d, self.w
        B = len(states)
        ids = torch.tensor([s.last_id for s in states], dtype=torch.long)
        e0 = w["embed"][ids]                                            # embed_input_ids (talker.py:637)
        last_h = torch.stack([s.last_hidden for s in states])
        text = torch.stack([(s.tail_text.pop(0) if s.tail_text else s.tts_pad) for s in states])  # talker.py:618-629
        cp_kw = dict(cp_kw or {})
        cp_kw.setdefault("step", [len(s.out_ids) for s in states])     # RNG key = tokens generated so far
        x, codes = self.talker_mtp(ids, e0, last_h, text, **cp_kw)
        positions = torch.tensor([s.seq_len for s in states], dtype=torch.long)
        seq_after = [s.seq_len + 1 for s in states]
        deltas = torch.tensor([s.rope_delta for s in states], dtype=torch.long)
        hidden = self.backbone(x, positions, list(range(B)), block_tables, seq_after,
                               rope_positions=positions + deltas if bool(deltas.any()) else None)
        logits = self.compute_logits(hidden)
        sampled = []
        for b, s in enumerate(states):
            # `sampling`: one dict for the batch, or one per request (vLLM samples per request, gpu_model_runner.py:315-319);
            # a per-request dict may carry its own "greedy"
            kw = dict(sampling[b]) if isinstance(sampling, (list, tuple)) else dict(sampling or {})
            g_b = bool(kw.pop("greedy", greedy))
            seen = None
            if kw.get("rep_penalty", 1.0) != 1.0:
                seen = [d.codec_pad_id] * s.prompt_len + s.out_ids   # prompt ids are pad placeholders (talker.py:603-605)
            tok = sample_row(logits[b], greedy=g_b, seen_ids=seen, step=len(s.out_ids), **kw)
            sampled.append(tok)
            s.seq_len += 1
            s.last_id = tok
            s.last_hidden = hidden[b]                                   # postprocess (talker.py:649-655)
            s.out_ids.append(tok)
        return logits, torch.tensor(sampled), hidden, codes, self.last_slots

    def prefill(self, states: list[OracleState], prompt_embeds: list[torch.Tensor], block_tables: list,
                *, greedy: bool = True, sampling: dict | None = None, rope_positions: torch.Tensor | None = None):
        """Whole-prompt prefill of each request (one chunk).  rope_positions [3, T]: M-RoPE ids of the flat token batch (the
        requests' get_input_positions_tensor outputs side by side); the states' rope_delta carry into the decode steps."""
        d = self.d
        xs, pos, req = [], [], []
        for r, pe in enumerate(prompt_embeds):
            n = pe.shape[0]
            xs.append(pe)
            pos += list(range(n))
            req += [r] * n
        x = torch.cat(xs, 0)
        seq_after = [pe.shape[0] for pe in prompt_embeds]
        hidden = self.backbone(x, torch.tensor(pos), req, block_tables, seq_after, rope_positions=rope_positions)
        last_idx = np.cumsum(seq_after) - 1
        hl = hidden[torch.as_tensor(last_idx)]
        logits = self.compute_logits(hl)
        out = []
        for b, s in enumerate(states):
            kw = dict(sampling[b]) if isinstance(sampling, (list, tuple)) else dict(sampling or {})
            g_b = bool(kw.pop("greedy", greedy))
            s.prompt_len = seq_after[b]
            seen = [d.codec_pad_id] * s.prompt_len if kw.get("rep_penalty", 1.0) != 1.0 else None
            tok = sample_row(logits[b], greedy=g_b, seen_ids=seen, step=0, **kw)
            s.seq_len = seq_after[b]
            s.last_id = tok
            s.last_hidden = hl[b]
            s.out_ids.append(tok)
            out.append(tok)
        return logits, torch.tensor(out), hl


# --------------------------------------------------------------------------
# Sparse MoE block of the Qwen3-Omni talker backbone.  The reference runs it on vLLM's FusedMoE
# (third party, absent: qwen3_omni_moe_talker.py via vllm Qwen3MoeSparseMoeBlock); the published algorithm restated
# here is HF transformers 5.15 Qwen3OmniMoeTalkerTextSparseMoeBlock (modeling_qwen3_omni_moe.py:2651-2733), bf16
# tensors with its rounding points: bf16 router logits -> fp32 softmax -> top-k (weights cast to bf16, optionally
# renormalised) -> per expert IN INDEX ORDER: gate_up, silu * up, down, * weight, bf16 index_add -> + sigmoid-gated
# shared expert.
# --------------------------------------------------------------------------
def moe_route(x: torch.Tensor, w_router: torch.Tensor, top_k: int, norm_topk_prob: bool = False):
    logits = linear(x, w_router)                                             # bf16 [T, E]
    probs = torch.softmax(logits.to(torch.float32), dim=-1)
    val, idx = torch.topk(probs, top_k, dim=-1)
    if norm_topk_prob:
        val = val / val.sum(dim=-1, keepdim=True)
    return logits, val.to(BF16), idx


def moe_block(x: torch.Tensor, w: dict, top_k: int, norm_topk_prob: bool = False) -> torch.Tensor:
    """x bf16 [T, H]; w: router [E, H], gate_up [E, 2I, H], down [E, H, I], shared_gate_up [2Is, H], shared_down [H, Is],
    shared_gate [1, H] -> bf16 [T, H]."""
    T, H = x.shape
    E, two_i = w["gate_up"].shape[0], w["gate_up"].shape[1]
    I = two_i // 2
    _, weights, idx = moe_route(x, w["router"], top_k, norm_topk_prob)
    out = torch.zeros_like(x)
    for e in range(E):
        kpos, tok = torch.where(idx.t() == e)                                # [k, T] mask like HF's expert_mask[e]
        if tok.numel() == 0:
            continue
        gu = linear(x[tok], w["gate_up"][e])
        act = silu_mul(gu[:, :I], gu[:, I:])
        y = linear(act, w["down"][e])
        y = y * weights[tok, kpos, None]
        out.index_add_(0, tok, y)                                            # bf16 accumulate, expert index order
    sgu = linear(x, w["shared_gate_up"])
    Is = w["shared_gate_up"].shape[0] // 2
    shared = linear(silu_mul(sgu[:, :Is], sgu[:, Is:]), w["shared_down"])
    shared = torch.sigmoid(linear(x, w["shared_gate"])) * shared
    return out + shared


# --------------------------------------------------------------------------
# SnakeBeta activation of the Code2Wav decoder (tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:602-700)
# --------------------------------------------------------------------------
def snake_beta(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """x [B, C, T]: x + 1 / (exp(beta) + 1e-9) * sin(x * exp(alpha))^2, per channel, fp32."""
    ea = torch.exp(alpha.to(torch.float32))[None, :, None]
    ib = (1.0 / (torch.exp(beta.to(torch.float32)) + 1e-9))[None, :, None]
    xf = x.to(torch.float32)
    return (xf + ib * torch.sin(xf * ea) ** 2).to(x.dtype)


# --------------------------------------------------------------------------
# KV extraction (kv_transfer_manager.py:224-301, kv_utils.py:13-85)
# --------------------------------------------------------------------------
def extract_kv(layer_kv: torch.Tensor, block_ids: list[int], seq_len: int) -> tuple[torch.Tensor, torch.Tensor]:
    if layer_kv.shape[0] == 2:
        kb, vb = layer_kv[0], layer_kv[1]
    else:
        kb, vb = layer_kv[:, 0], layer_kv[:, 1]
    mx = min(kb.shape[0], vb.shape[0]) - 1
    ids = [b for b in block_ids if 0 <= b <= mx]
    fk, fv = kb[ids].flatten(0, 1), vb[ids].flatten(0, 1)
    if seq_len < fk.shape[0]:
        fk, fv = fk[:seq_len], fv[:seq_len]
    return fk.contiguous(), fv.contiguous()


# --------------------------------------------------------------------------
# Qwen3-Omni talker prompt builder (row a11: qwen3_omni.py:586-606 talker_preprocess, 650-676 _get_tts_embed,
# 678-809 talker_preprocess_prefill, 834-905 _thinker_to_talker_prefill, 975-992 _get_talker_user_parts,
# 994-1060 _get_talker_assistant_parts, 907-973 decode-side text steps)
# --------------------------------------------------------------------------
def resize_mlp(x: torch.Tensor, w: dict) -> torch.Tensor:
    """HF Qwen3OmniMoeTalkerResizeMLP in bf16: linear_fc2(silu(linear_fc1(x))), every module output rounded to bf16."""
    h = linear(x, w["fc1_w"], w.get("fc1_b"))
    hf = h.to(torch.float32)
    h = (hf / (1.0 + torch.exp(-hf))).to(BF16)
    return linear(h, w["fc2_w"], w.get("fc2_b"))


def _last_row(x, width: int) -> torch.Tensor:
    """_get_tts_embed._ensure_1x1 (qwen3_omni.py:654-660) + the zero fallback (662-670) -> [1, width] bf16."""
    if not isinstance(x, torch.Tensor) or x.numel() == 0:
        return torch.zeros(1, width, dtype=BF16)
    if x.ndim == 3:
        return x[0, -1:, :].to(BF16)
    if x.ndim == 2:
        return x[-1:].to(BF16)
    return x.reshape(1, -1).to(BF16)


def omni_talker_prompt(thinker_embed: torch.Tensor, thinker_hidden: torch.Tensor, input_ids, result_ids, speaker_id: int,
                       tts_bos, tts_eos, tts_pad, w: dict, ids: dict):
    """_thinker_to_talker_prefill: (talker_input_ids int64 [P], talker_input_embeds bf16 [P, H], trailing_text_hidden bf16
    [n, H]).  thinker_embed / thinker_hidden bf16 [T, Ht]; input_ids = the thinker's chatml prompt ids, result_ids = prompt +
    generated ids [T]; w = {"text": resize-mlp weights, "hidden": resize-mlp weights, "codec_embed": [V, H]};
    ids = token ids (im_start, system, user, assistant, tts_pad_token, audio, image, video, codec_nothink, codec_think_bos,
    codec_think_eos, codec_pad, codec_bos)."""
    input_ids = torch.as_tensor(input_ids, dtype=torch.long).reshape(-1)
    result_ids = torch.as_tensor(result_ids, dtype=torch.long).reshape(-1)
    thinker_embed, thinker_hidden = thinker_embed.to(BF16), thinker_hidden.to(BF16)
    Ht = thinker_embed.shape[-1]
    H = w["codec_embed"].shape[1]
    starts = torch.nonzero(input_ids == ids["im_start"]).reshape(-1).tolist()
    if len(starts) < 2:   # the reference's torch.cat over a 0-d / empty index tensor raises here (qwen3_omni.py:851-857,903)
        raise ValueError("omni_talker_prompt: need at least two <|im_start|> segments")
    bounds = starts + [int(result_ids.shape[0])]
    mm = (result_ids == ids["audio"]) | (result_ids == ids["image"]) | (result_ids == ids["video"])
    bos, eos, pad = (resize_mlp(_last_row(t, Ht), w["text"]) for t in (tts_bos, tts_eos, tts_pad))
    emb, tid, trailing = [], [], None
    for i in range(len(bounds) - 1):
        s, e = bounds[i], bounds[i + 1]
        role = int(input_ids[s + 1])
        if role == ids["system"]:
            continue
        if role == ids["user"]:
            part = torch.zeros(e - s, H, dtype=BF16)
            m = mm[s:e]
            if bool(m.any()):
                part[m] = resize_mlp(thinker_hidden[s:e][m], w["hidden"])
            part[~m] = resize_mlp(thinker_embed[s:e][~m], w["text"])
            emb.append(part)
            tid.append(result_ids[s:e])
        elif role == ids["assistant"] and i == len(bounds) - 2:
            ah = resize_mlp(thinker_embed[s:e], w["text"])
            first = ah[3:4] if ah.shape[0] > 3 else torch.zeros(1, H, dtype=BF16)
            text = torch.cat([ah[:3], pad.expand(4, -1), bos, first], 0)
            codec_ids = torch.tensor([ids["codec_nothink"], ids["codec_think_bos"], ids["codec_think_eos"], int(speaker_id),
                                      ids["codec_pad"], ids["codec_bos"]], dtype=torch.long)
            codec = torch.cat([torch.zeros(3, H, dtype=BF16), w["codec_embed"][codec_ids].to(BF16)], 0)
            emb.append(text + codec)
            tid.append(torch.full((text.shape[0],), ids["tts_pad_token"], dtype=torch.long))
            trailing = torch.cat([ah[4:], eos], 0) if ah.shape[0] > 4 else eos
        elif role == ids["assistant"]:
            continue
        else:
            raise AssertionError("Expect role id after <|im_start|> (assistant, user, system)")
    if not emb:
        raise ValueError("omni_talker_prompt: no user / assistant segment")
    return torch.cat(tid, 0), torch.cat(emb, 0), trailing


def omni_text_step_pop(tail, tts_pad_proj: torch.Tensor):
    """talker_preprocess_decode, non-streaming branch (qwen3_omni.py:948-960): (text_step [1, H], new tail)."""
    if isinstance(tail, torch.Tensor) and tail.numel() > 0:
        return tail[0:1], (tail[1:] if tail.shape[0] > 1 else tts_pad_proj.reshape(1, -1))
    return tts_pad_proj.reshape(1, -1), tail


def omni_text_step_streaming(state: dict, n_thinker_output_ids: int, tts_eos_proj, tts_pad_proj, w_text: dict):
    """_thinker_decode_to_talker_decode (qwen3_omni.py:907-937), streaming (async_chunk) branch.  state: num_processed_tokens,
    finished_flag, cached [n, Ht] or None, fresh [m, Ht] or None; mutated like the reference's update_dict."""
    start = state.get("num_processed_tokens", 0)
    if start >= n_thinker_output_ids - 1:
        if state.get("finished_flag"):
            return tts_pad_proj
        state["finished_flag"] = True
        return tts_eos_proj
    cached, fresh = state.get("cached"), state.get("fresh")
    if cached is not None and start < cached.shape[0]:
        x = cached[start]
        if fresh is not None:
            state["cached"] = torch.cat([cached, fresh], 0)
    else:
        x = fresh
    state["fresh"] = None
    return resize_mlp(x.to(BF16), w_text)


# --------------------------------------------------------------------------
# Prompt-embedding builder of the Qwen3-TTS talker (reference:
# qwen3_tts_talker.py:1160-1209 _generate_icl_prompt, 1211-1567 _build_prompt_embeds).  Token ids in (the tokenizer and the
# speaker encoder run before this point), embeddings out.  Pinned by tests/golden/tts_prompt_builder.pt (the reference's own
# method on a stand-in self).  w = {"text_embedding" [Vt, Ht], "text_projection" {fc1_w, fc1_b, fc2_w, fc2_b},
# "codec_embed" [V, H], "cp_embed" [Q-1, Vc, H]}; ids = {tts_bos, tts_eos, tts_pad, codec_nothink, codec_think,
# codec_think_bos, codec_think_eos, codec_pad, codec_bos}.
# --------------------------------------------------------------------------
def tts_talker_prompt(w: dict, ids: dict, task_type: str, input_ids, *, language_id=None, speaker_id=None, speaker_embed=None,
                      instruct_ids=None, ref_ids=None, ref_code=None, in_context_mode: bool = False,
                      non_streaming_mode: bool | None = None):
    """-> (talker_prompt [P, H], trailing_text_hidden [T, H], tts_pad_embed [1, H], ref_code_len | None), all bf16."""
    if non_streaming_mode is None:
        non_streaming_mode = task_type in ("CustomVoice", "VoiceDesign")                 # talker.py:1226-1230
    input_ids = torch.as_tensor(input_ids, dtype=torch.long).reshape(-1)
    emb = lambda t: w["codec_embed"][torch.as_tensor(t, dtype=torch.long)]              # noqa: E731  embed_input_ids
    proj = lambda t: resize_mlp(w["text_embedding"][torch.as_tensor(t, dtype=torch.long)], w["text_projection"])   # noqa: E731
    tts_bos, tts_eos, tts_pad = (r[None] for r in proj([ids["tts_bos"], ids["tts_eos"], ids["tts_pad"]]))          # 1245-1252
    if language_id is None:                                                              # 1270-1286
        pre = [ids["codec_nothink"], ids["codec_think_bos"], ids["codec_think_eos"]]
    else:
        pre = [ids["codec_think"], ids["codec_think_bos"], int(language_id), ids["codec_think_eos"]]
    codec_input_0, codec_input_1 = emb(pre), emb([ids["codec_pad"], ids["codec_bos"]])
    if task_type == "Base":
        spk = speaker_embed.to(BF16).reshape(1, -1)
    elif task_type == "CustomVoice":
        spk = emb([int(speaker_id)])
    elif task_type == "VoiceDesign":
        spk = None
    else:
        raise ValueError(f"Unsupported task_type={task_type}")
    codec_input = torch.cat([codec_input_0] + ([spk] if spk is not None else []) + [codec_input_1], 0)
    role = proj(input_ids[:3])                                                           # <|im_start|>assistant\n
    n = codec_input.shape[0]
    codec_prefix = torch.cat([tts_pad.expand(n - 2, -1), tts_bos], 0) + codec_input[:-1]
    prompt = torch.cat([role, codec_prefix], 0)
    ref_code_len = None
    if task_type == "Base" and in_context_mode:                                          # _generate_icl_prompt
        ref_code = torch.as_tensor(ref_code, dtype=torch.long)
        ref_code_len = int(ref_code.shape[0])
        ref_ids = torch.as_tensor(ref_ids, dtype=torch.long).reshape(-1)
        text_embed = torch.cat([proj(torch.cat([ref_ids[3:-2], input_ids[3:-5]])), tts_eos], 0)
        Q = ref_code.shape[1]
        parts = [emb(ref_code[:, 0])[:, None]] + [w["cp_embed"][i - 1][ref_code[:, i]][:, None] for i in range(1, Q)]
        codec_sum = torch.cat([emb([ids["codec_bos"]]), torch.cat(parts, 1).sum(1)], 0)  # bf16 sum over the Q groups
        tl, cl = text_embed.shape[0], codec_sum.shape[0]
        if non_streaming_mode:
            icl = torch.cat([text_embed + emb([ids["codec_pad"]] * tl), codec_sum + tts_pad], 0)
            trailing = tts_pad
        elif tl > cl:
            icl, trailing = text_embed[:cl] + codec_sum, text_embed[cl:]
        else:
            icl = torch.cat([text_embed] + [tts_pad] * (cl - tl), 0) + codec_sum
            trailing = tts_pad
        prompt = torch.cat([prompt, icl], 0)
    elif non_streaming_mode:                                                             # 1424-1445
        text_all = torch.cat([proj(input_ids[3:-5]), tts_eos], 0)
        prompt = torch.cat([prompt, text_all + emb([ids["codec_pad"]] * text_all.shape[0]), tts_pad + emb([ids["codec_bos"]])], 0)
        trailing = tts_pad
    else:                                                                                # 1446-1455
        prompt = torch.cat([prompt, proj(input_ids[3:4]) + codec_input[-1:]], 0)
        trailing = torch.cat([proj(input_ids[4:-5]), tts_eos], 0)
    if instruct_ids is not None and len(instruct_ids):
        prompt = torch.cat([proj(torch.as_tensor(instruct_ids, dtype=torch.long).reshape(-1)), prompt], 0)
    return prompt, trailing, tts_pad, ref_code_len
