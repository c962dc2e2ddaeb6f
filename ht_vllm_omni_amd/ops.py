"""Tensor-level wrappers over the per-op C-ABI entry points (device memory via PyTorch-ROCm,
arithmetic in libomni_talker.so).  Every wrapper launches on torch's current stream."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L

BF16 = torch.bfloat16


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.OmniError("omni ops need device tensors (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise L.OmniError("omni ops need contiguous tensors")


def rmsnorm(x, w, eps, *, delta=None, residual=None):
    """out = w * bf16(v * rstd); v = residual(+delta, updated in place) or x."""
    src = residual if residual is not None else x
    _chk_dev(x, w, delta, residual)
    out = torch.empty_like(src)
    rows, hidden = src.shape
    L.check(L.load().omni_rmsnorm(L.ptr(x), L.ptr(delta), L.ptr(residual), L.ptr(w), L.ptr(out), rows, hidden,
                                  float(eps), L.current_stream()), "omni_rmsnorm")
    return out


def gemm(x, w, *, bias=None, epilogue=L.EPI_BF16, mask=None, layout=0, M=None):
    """out[M,N] = x[M,K] . w[N,K]^T ; silu_mul: w = [gate|up] rows -> N = rows/2.
    layout bits (LAYOUT_*_FRAG): w / x given, out produced in fragment-major order (x, out then have 16-row padding;
    pass the true M)."""
    _chk_dev(x, w, bias, mask)
    Mx, K = x.shape
    M = Mx if M is None else M
    N = w.shape[0] // 2 if epilogue in L.SILU_EPIS else w.shape[0]
    dt = BF16 if epilogue in (L.EPI_BF16,) + L.SILU_EPIS else torch.float32
    rows_out = (M + 15) // 16 * 16 if layout & L.LAYOUT_OUT_FRAG else M
    out = torch.zeros(rows_out, N, dtype=dt, device=x.device)
    L.check(L.load().omni_gemm_bf16_ex(L.ptr(x), x.stride(0), L.ptr(w), L.ptr(bias), L.ptr(out), M, N, K, epilogue,
                                       L.ptr(mask), layout, L.current_stream()), "omni_gemm_bf16")
    return out


def silu_mul(gate_up):
    """[T, 2I] = [gate | up] -> bf16(bf16(silu(gate)) * up) [T, I]."""
    _chk_dev(gate_up)
    T, two_i = gate_up.shape
    out = torch.empty(T, two_i // 2, dtype=BF16, device=gate_up.device)
    L.check(L.load().omni_silu_mul(L.ptr(gate_up), L.ptr(out), T, two_i // 2, L.current_stream()), "omni_silu_mul")
    return out


def silu(x):
    """bf16(silu(x)) elementwise."""
    _chk_dev(x)
    out = torch.empty_like(x)
    L.check(L.load().omni_silu(L.ptr(x), L.ptr(out), x.numel(), L.current_stream()), "omni_silu")
    return out


def resize_mlp(x, w):
    """HF Qwen3OmniMoeTalkerResizeMLP: linear_fc2(silu(linear_fc1(x))) over x bf16 [T, H_in]; w = {fc1_w [I, H_in], fc1_b [I],
    fc2_w [H_out, I], fc2_b [H_out]} (biases optional)."""
    _chk_dev(x, w["fc1_w"], w.get("fc1_b"), w["fc2_w"], w.get("fc2_b"))
    T, h_in = x.shape
    inter, h_out = w["fc1_w"].shape[0], w["fc2_w"].shape[0]
    out = torch.empty(T, h_out, dtype=BF16, device=x.device)
    ws = torch.empty(max(1, min(T, 64)), inter, dtype=BF16, device=x.device)
    L.check(L.load().omni_resize_mlp(L.ptr(x), L.ptr(w["fc1_w"]), L.ptr(w.get("fc1_b")), L.ptr(w["fc2_w"]), L.ptr(w.get("fc2_b")),
                                     L.ptr(ws), L.ptr(out), T, h_in, inter, h_out, L.current_stream()), "omni_resize_mlp")
    return out


def moe_route(logits, top_k, norm_topk_prob=False):
    """bf16 router logits [T, E] -> (topk_idx int32 [T, k], topk_w bf16 [T, k])."""
    _chk_dev(logits)
    T, E = logits.shape
    idx = torch.empty(T, top_k, dtype=torch.int32, device=logits.device)
    wts = torch.empty(T, top_k, dtype=BF16, device=logits.device)
    L.check(L.load().omni_moe_route(L.ptr(logits), T, E, top_k, int(norm_topk_prob), L.ptr(idx), L.ptr(wts), L.current_stream()),
            "omni_moe_route")
    return idx, wts


def moe_block(x, w, top_k, norm_topk_prob=False):
    """Sparse-MoE MLP of the Omni talker on device tensors.  w: router [E, H] (row-major), gate_up_f [E, 2I, H] and down_f
    [E, H, I] fragment-major per expert (engine.frag_shuffle), shared_gate_up [2Is, H], shared_down [H, Is], shared_gate
    [1, H] (row-major).  Returns bf16 [T, H]."""
    T, H = x.shape
    E, two_i = w["gate_up_f"].shape[0], w["gate_up_f"].shape[1]
    I = two_i // 2
    logits = gemm(x, w["router"])
    idx, wts = moe_route(logits, top_k, norm_topk_prob)
    shared = None
    if w.get("shared_gate_up") is not None:
        shared = gemm(gemm(x, w["shared_gate_up"], epilogue=L.EPI_SILU_MUL), w["shared_down"])
    act = torch.empty(T * top_k, I, dtype=BF16, device=x.device)
    y = torch.empty(T * top_k, H, dtype=BF16, device=x.device)
    out = torch.empty(T, H, dtype=BF16, device=x.device)
    L.check(L.load().omni_moe_experts(L.ptr(x), L.ptr(idx), L.ptr(wts), L.ptr(w["gate_up_f"]), L.ptr(w["down_f"]), L.ptr(shared),
                                      L.ptr(w.get("shared_gate")), L.ptr(act), L.ptr(y), L.ptr(out), T, H, I, E, top_k,
                                      L.current_stream()), "omni_moe_experts")
    return out, idx, wts


def snake_beta(x, exp_alpha, inv_beta):
    """SnakeBeta of the Code2Wav decoder: x [B, C, T] fp32 or bf16 -> x + inv_beta[c] * sin(x * exp_alpha[c])^2."""
    _chk_dev(x, exp_alpha, inv_beta)
    if x.ndim != 3 or x.dtype not in (torch.float32, BF16) or exp_alpha.dtype != torch.float32 or inv_beta.dtype != torch.float32:
        raise L.OmniError("snake_beta: x [B, C, T] fp32 / bf16, per-channel fp32 parameters")
    B, Cc, T = x.shape
    out = torch.empty_like(x)
    L.check(L.load().omni_snake_beta(L.ptr(x), L.ptr(exp_alpha), L.ptr(inv_beta), L.ptr(out), B, Cc, T, int(x.dtype == BF16),
                                     L.current_stream()), "omni_snake_beta")
    return out


def gemm_resid(x, w_frag, r_frag, partials, *, bias=None, accumulate=True, x_frag=True, M=None):
    """In place on the fragment-major residual stream: r = bf16((r if accumulate) + bf16(x . w^T + bias)); fills
    partials[N/16, 64] (per-row sum(r^2) slabs) and returns the slab count."""
    _chk_dev(x, w_frag, r_frag, partials, bias)
    Mx, K = x.shape
    M = Mx if M is None else M
    N = w_frag.shape[0]
    np_out = C.c_int(0)
    layout = L.LAYOUT_W_FRAG | (L.LAYOUT_X_FRAG if x_frag else 0)
    L.check(L.load().omni_gemm_resid(L.ptr(x), x.stride(0), L.ptr(w_frag), L.ptr(bias), L.ptr(r_frag), int(accumulate),
                                     L.ptr(partials), C.addressof(np_out), M, N, K, layout, L.current_stream()),
            "omni_gemm_resid")
    return np_out.value


def gemm_xnorm(r_frag, partials, nparts, norm_w, w_frag, eps, *, M, epilogue=L.EPI_BF16, mask=None, want_normed=False,
               out_frag=False):
    """out = epilogue((norm_w * bf16(r * rstd)) . w^T) on the fragment-major residual stream -> out [, normed rows]."""
    _chk_dev(r_frag, partials, norm_w, w_frag, mask)
    K = r_frag.shape[1]
    N = w_frag.shape[0] // 2 if epilogue in L.SILU_EPIS else w_frag.shape[0]
    dt = BF16 if epilogue in (L.EPI_BF16,) + L.SILU_EPIS else torch.float32
    rows_out = (M + 15) // 16 * 16 if out_frag else M
    out = torch.zeros(rows_out, N, dtype=dt, device=r_frag.device)
    normed = torch.empty(M, K, dtype=BF16, device=r_frag.device) if want_normed else None
    L.check(L.load().omni_gemm_xnorm(L.ptr(r_frag), L.ptr(partials), nparts, L.ptr(norm_w), float(eps), L.ptr(normed),
                                     L.ptr(w_frag), L.ptr(out), M, N, K, epilogue, L.ptr(mask), int(out_frag),
                                     L.current_stream()), "omni_gemm_xnorm")
    return (out, normed) if want_normed else out


def attn_decode_fused(qkv, qnorm_w, knorm_w, positions, cos_sin, k_cache, v_cache, block_table, seq_lens, *, q_heads,
                      kv_heads, head_dim, block_size, kv_dtype, eps, k_scale=1.0, v_scale=1.0, k_scales=None, v_scales=None,
                      max_seq_len=4096, split=True):
    """q/k-norm + RoPE + KV write of the new token + paged attention in one launch -> (out, slots)."""
    _chk_dev(qkv, qnorm_w, knorm_w, positions, cos_sin, k_cache, v_cache, block_table, seq_lens, k_scales, v_scales)
    B = qkv.shape[0]
    out = torch.empty(B, q_heads * head_dim, dtype=BF16, device=qkv.device)
    slots = torch.full((B,), -7, dtype=torch.int64, device=qkv.device)
    ws = None
    if split:
        nbytes = L.load().omni_paged_attn_workspace_bytes(B, q_heads, head_dim, max_seq_len)
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=qkv.device)      # (opens with the split merge's arrival counters: zero)
    L.check(L.load().omni_attn_decode_fused(
        L.ptr(qkv), L.ptr(qnorm_w), L.ptr(knorm_w), L.ptr(positions), L.ptr(cos_sin), float(eps), L.ptr(k_cache),
        L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), L.ptr(block_table), block_table.stride(0), L.ptr(seq_lens),
        L.ptr(slots), L.ptr(out), L.ptr(ws), B, q_heads, kv_heads, head_dim, block_size, kv_dtype, float(k_scale),
        float(v_scale), 1.0 / math.sqrt(head_dim), max_seq_len, L.current_stream()), "omni_attn_decode_fused")
    return out, slots


def slot_mapping(block_table, positions, block_size, B_padded=None):
    _chk_dev(block_table, positions)
    B = positions.shape[0]
    Bp = B if B_padded is None else B_padded
    out = torch.empty(Bp, dtype=torch.int64, device=positions.device)
    L.check(L.load().omni_slot_mapping(L.ptr(block_table), block_table.stride(0), L.ptr(positions), L.ptr(out), B, Bp,
                                       block_size, L.current_stream()), "omni_slot_mapping")
    return out


def mrope_axis_table(mrope_section, interleaved: bool) -> torch.Tensor:
    """uint8 [64]: the position-id axis (0 temporal, 1 height, 2 width) rotary pair p of a 128-wide head uses -- vLLM
    MRotaryEmbedding.forward: chunked sections (cos split by `mrope_section`, chunk i from row i) or `apply_interleaved_rope`
    (row 1 at p = 1, 4, ... < 3 * section[1], row 2 at p = 2, 5, ... < 3 * section[2], row 0 elsewhere; Qwen3-Omni)."""
    t, h, w = (int(v) for v in mrope_section)
    assert t + h + w == 64, mrope_section
    ax = torch.zeros(64, dtype=torch.uint8)
    if interleaved:
        ax[1:3 * h:3] = 1
        ax[2:3 * w:3] = 2
    else:
        ax[t:t + h] = 1
        ax[t + h:] = 2
    return ax


def qknorm_rope_kvwrite(qkv, qnorm_w, knorm_w, positions, cos_sin, slots, k_cache, v_cache, *, q_heads, kv_heads,
                        head_dim, eps, kv_dtype, k_scale=1.0, v_scale=1.0, k_scales=None, v_scales=None, mrope_axis=None):
    """positions int32 [T]; or, with mrope_axis (uint8 [64] on the device, `mrope_axis_table`), int32 [3, T] M-RoPE ids."""
    _chk_dev(qkv, qnorm_w, knorm_w, positions, cos_sin, slots, k_cache, v_cache, k_scales, v_scales, mrope_axis)
    T = qkv.shape[0]
    q = torch.empty(T, q_heads * head_dim, dtype=BF16, device=qkv.device)
    if mrope_axis is not None:
        assert positions.shape == (3, T) and positions.dtype == torch.int32 and positions.is_contiguous(), positions.shape
        assert mrope_axis.dtype == torch.uint8 and mrope_axis.numel() == 64
        L.check(L.load().omni_qknorm_mrope_kvwrite(
            L.ptr(qkv), L.ptr(qnorm_w), L.ptr(knorm_w), L.ptr(positions), L.ptr(mrope_axis), L.ptr(cos_sin), L.ptr(slots), L.ptr(q),
            L.ptr(k_cache), L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), T, q_heads, kv_heads, head_dim, float(eps),
            kv_dtype, float(k_scale), float(v_scale), L.current_stream()), "omni_qknorm_mrope_kvwrite")
        return q
    L.check(L.load().omni_qknorm_rope_kvwrite(
        L.ptr(qkv), L.ptr(qnorm_w), L.ptr(knorm_w), L.ptr(positions), L.ptr(cos_sin), L.ptr(slots), L.ptr(q),
        L.ptr(k_cache), L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), T, q_heads, kv_heads, head_dim, float(eps),
        kv_dtype, float(k_scale), float(v_scale), L.current_stream()), "omni_qknorm_rope_kvwrite")
    return q


def paged_attn_decode(q, k_cache, v_cache, block_table, seq_lens, *, q_heads, kv_heads, head_dim, block_size, kv_dtype,
                      k_scale=1.0, v_scale=1.0, k_scales=None, v_scales=None, max_seq_len=4096, split=True, workspace=None):
    """workspace: a caller-owned fp32 buffer of omni_paged_attn_workspace_bytes, zero-filled once (reused from call to call)."""
    _chk_dev(q, k_cache, v_cache, block_table, seq_lens, k_scales, v_scales)
    B = q.shape[0]
    out = torch.empty_like(q)
    ws = workspace
    if split and ws is None:
        nbytes = L.load().omni_paged_attn_workspace_bytes(B, q_heads, head_dim, max_seq_len)
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=q.device)
    L.check(L.load().omni_paged_attn_decode(
        L.ptr(q), L.ptr(k_cache), L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), L.ptr(block_table),
        block_table.stride(0), L.ptr(seq_lens), L.ptr(out), L.ptr(ws), B, q_heads, kv_heads, head_dim, block_size,
        kv_dtype, float(k_scale), float(v_scale), 1.0 / math.sqrt(head_dim), max_seq_len, L.current_stream()),
        "omni_paged_attn_decode")
    return out


def paged_attn_prefill(q, k_cache, v_cache, block_table, req_of_tok, positions, *, q_heads, kv_heads, head_dim,
                       block_size, kv_dtype, k_scale=1.0, v_scale=1.0, k_scales=None, v_scales=None):
    _chk_dev(q, k_cache, v_cache, block_table, req_of_tok, positions, k_scales, v_scales)
    T = q.shape[0]
    out = torch.empty_like(q)
    L.check(L.load().omni_paged_attn_prefill(
        L.ptr(q), L.ptr(k_cache), L.ptr(v_cache), L.ptr(k_scales), L.ptr(v_scales), L.ptr(block_table),
        block_table.stride(0), L.ptr(req_of_tok), L.ptr(positions), L.ptr(out), T, q_heads, kv_heads, head_dim,
        block_size, kv_dtype, float(k_scale), float(v_scale), 1.0 / math.sqrt(head_dim), L.current_stream()),
        "omni_paged_attn_prefill")
    return out


def embed(ids, table):
    _chk_dev(ids, table)
    T = ids.shape[0]
    out = torch.empty(T, table.shape[1], dtype=BF16, device=table.device)
    L.check(L.load().omni_embed(L.ptr(ids), L.ptr(table), L.ptr(out), T, table.shape[1], table.shape[0],
                                L.current_stream()), "omni_embed")
    return out


def sample(logits, *, greedy, temperature=1.0, top_k=0, top_p=1.0, rep_penalty=1.0, seen=None, seed=0, steps=None, step_mul=1,
           step_add=0, inc_steps=False):
    _chk_dev(logits, seen, steps)
    B, V = logits.shape
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    L.check(L.load().omni_sample(L.ptr(logits), logits.stride(0), B, V, int(greedy), float(temperature), int(top_k),
                                 float(top_p), float(rep_penalty), L.ptr(seen), int(seed) & 0xFFFFFFFF, L.ptr(steps), step_mul,
                                 step_add, int(inc_steps), L.ptr(out), L.current_stream()), "omni_sample")
    return out


def sample_rows(logits, rows: dict, *, seen=None, steps=None, step_mul=1, step_add=0, inc_steps=False):
    """Sampler with every parameter per row: rows = {greedy int32, temperature f32, top_k int32, top_p f32,
    rep_penalty f32, seed uint32-as-int32/int64} device tensors [B] (one SamplingParams per request)."""
    B, V = logits.shape
    ts = {k: rows[k] for k in ("greedy", "temperature", "top_k", "top_p", "rep_penalty", "seed")}
    _chk_dev(logits, seen, steps, *ts.values())
    rs = L.RowSampling(*[t.data_ptr() for t in ts.values()])
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    L.check(L.load().omni_sample_rows(L.ptr(logits), logits.stride(0), B, V, C.byref(rs), L.ptr(seen), L.ptr(steps), step_mul,
                                      step_add, int(inc_steps), L.ptr(out), L.current_stream()), "omni_sample_rows")
    return out


def rope_table(max_pos: int, head_dim: int, theta: float) -> torch.Tensor:
    """Host-built cos/sin table, bf16 [max_pos][2][head_dim/2]: fp32 cos/sin cast to bf16, the
    HF / reference numerics (qwen3_tts_code_predictor_vllm.py:80-93)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return torch.stack((freqs.cos(), freqs.sin()), dim=1).to(BF16).contiguous()


def gemm_tile(x: torch.Tensor, w_frag: torch.Tensor, *, M: int | None = None, bias=None, scale=None, act: int = L.TILE_ACT_NONE,
              resid=None, out=None, out_f32=None, out2=None, snake=None, taps: int = 1, dilation: int = 1,
              row_off: int | None = None, want: str = "b", tile_hint: int = 0, groups: int = 1, group_rows=None):
    """omni_gemm_tile: y[M, N] = act(A . W^T + bias) * scale (+ resid) on the matrix cores, A = x for taps == 1, else the
    causal conv window A[m] = [x[m + row_off], x[m + row_off + dilation], ...] (taps rows of x.shape[1] channels; row_off
    defaults to -(taps - 1) * dilation; rows outside x read as zero).  x bf16 [rows, C] (row stride = x.stride(0)); w_frag bf16
    [N, taps * C] fragment-major (engine.frag_shuffle); bias / scale fp32 [N]; resid fp32 [M, N].
    `want` names the outputs: "f" = fp32 y, "b" = bf16(y), "s" = bf16(snake(y)) with snake = (alpha, inv_beta) fp32 [N];
    given buffers (out_f32 / out / out2) are used, missing ones allocated.  Returns them in the order of `want` (a single
    tensor when one is wanted).
    groups > 1: the batched form -- x [groups * cap, C] (cap rows per group), w_frag [groups, N, K] (each matrix fragment-major),
    outputs [groups * cap, N]; group_rows int32 [groups] (device) = live rows per group (the rest is neither computed nor written)."""
    assert x.dtype == BF16 and w_frag.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    rows, Cin = x.shape
    if groups > 1:
        assert w_frag.dim() == 3 and w_frag.shape[0] == groups and w_frag.is_contiguous() and taps == 1 and rows % groups == 0
        N, K = w_frag.shape[1:]
        cap = rows // groups
    else:
        N, K = w_frag.shape
    assert K == taps * Cin, (K, taps, Cin)
    M = rows if M is None else M
    gu8 = act == L.TILE_ACT_SILU_MUL_GU8
    nout = N // 2 if gu8 else N
    g = L.TileGemm()
    g.x, g.x_rows, g.ldx = x.data_ptr(), rows, x.stride(0)
    g.seg_len, g.seg_rows = Cin, dilation
    g.row_off = (-(taps - 1) * dilation if row_off is None else row_off) if taps > 1 or row_off is not None else 0
    g.w, g.bias, g.scale, g.act = w_frag.data_ptr(), L.ptr(bias), L.ptr(scale), act
    if resid is not None:
        assert resid.dtype == torch.float32 and resid.stride(1) == 1
        g.resid, g.ldr = resid.data_ptr(), resid.stride(0)
    res = []
    for k in want:
        if k == "f":
            out_f32 = torch.empty(M, nout, dtype=torch.float32, device=x.device) if out_f32 is None else out_f32
            g.out_f32, g.ldf = out_f32.data_ptr(), out_f32.stride(0)
            res.append(out_f32)
        elif k == "b":
            out = torch.empty(M, nout, dtype=BF16, device=x.device) if out is None else out
            g.out, g.ldo = out.data_ptr(), out.stride(0)
            res.append(out)
        elif k == "s":
            out2 = torch.empty(M, nout, dtype=BF16, device=x.device) if out2 is None else out2
            g.out2, g.ldo2 = out2.data_ptr(), out2.stride(0)
            g.snake_alpha, g.snake_inv_beta = snake[0].data_ptr(), snake[1].data_ptr()
            res.append(out2)
        else:
            raise ValueError(want)
    g.M, g.N, g.K, g.tile_hint = M, N, K, tile_hint
    if groups > 1:
        assert M == rows, "a grouped launch covers every group's cap rows"
        g.M, g.groups, g.x_group_rows, g.w_group_elems, g.out_group_rows = cap, groups, cap, N * K, cap
        if group_rows is not None:
            assert group_rows.dtype == torch.int32 and group_rows.numel() == groups and group_rows.is_contiguous()
            g.group_rows = group_rows.data_ptr()
    L.check(L.load().omni_gemm_tile(C.byref(g), L.current_stream()), "omni_gemm_tile")
    return res[0] if len(res) == 1 else tuple(res)
