"""Synthetic talker weights (random-init, no checkpoints in this environment).

Layout mirrors what the reference's ``load_weights`` produces after its HF->vLLM
prefix mapping (qwen3_tts_talker.py:297-311,1569-1590): per layer a fused
``qkv_proj`` ([q rows | k rows | v rows], in-features contiguous) and a fused
``gate_up_proj`` ([gate rows | up rows]); ``torch.nn.Linear`` orientation [out, in].

Keys (all bf16, CPU):
  embed [V,H]  l{i}.ln1 [H]  l{i}.wqkv [(Hq+2Hkv)D,H]  l{i}.qnorm [D]  l{i}.knorm [D]
  l{i}.wo [H,Hq*D]  l{i}.ln2 [H]  l{i}.wgu [2I,H]  l{i}.wdown [H,I]  norm [H]  lm_head [V,H]
  MoE backbone (moe_experts > 0) instead of wgu / wdown: l{i}.moe_router [E,H]  moe_gate_up [E,2Im,H]  moe_down [E,H,Im]
  moe_shared_gate_up [2Is,H]  moe_shared_down [H,Is]  moe_shared_gate [1,H]
  cp.proj_w [Hc,H] cp.proj_b [Hc] (only when Hc != H)   cp.l{i}.* (same names, Hc dims)
  cp.norm [Hc]  cp.lm_head [Q-1,Vc,Hc]  cp.embed [Q-1,Vc,H]
"""
from __future__ import annotations

import torch

from .config import TalkerDims


def make_weights(d: TalkerDims, seed: int = 1234, std: float = 0.02, norm_noise: float = 0.0,
                 device: str = "cpu") -> dict[str, torch.Tensor]:
    """device='cpu' is the reproducible stream every parity test and the bench use; device='cuda' draws from the device
    generator (different values, same shapes) -- only for profiler runs that must avoid the 3.4 GB host-to-device copy."""
    g = torch.Generator(device=device).manual_seed(seed)

    def rnd(*shape):
        return (torch.randn(*shape, generator=g, dtype=torch.float32, device=device) * std).to(torch.bfloat16)

    def nrm(n):
        w = torch.ones(n, dtype=torch.float32, device=device)
        if norm_noise:
            w = w + torch.randn(n, generator=g, device=device) * norm_noise
        return w.to(torch.bfloat16)

    w: dict[str, torch.Tensor] = {}
    w["embed"] = rnd(d.vocab, d.hidden)
    for i in range(d.layers):
        p = f"l{i}."
        w[p + "ln1"] = nrm(d.hidden)
        w[p + "wqkv"] = rnd(d.qkv_out, d.hidden)
        w[p + "qnorm"] = nrm(d.head_dim)
        w[p + "knorm"] = nrm(d.head_dim)
        w[p + "wo"] = rnd(d.hidden, d.q_heads * d.head_dim)
        w[p + "ln2"] = nrm(d.hidden)
        if d.moe_experts > 0:
            w[p + "moe_router"] = (rnd(d.moe_experts, d.hidden).float() * 8).to(torch.bfloat16)     # spread the routing logits
            w[p + "moe_gate_up"] = rnd(d.moe_experts, 2 * d.moe_inter, d.hidden)
            w[p + "moe_down"] = rnd(d.moe_experts, d.hidden, d.moe_inter)
            w[p + "moe_shared_gate_up"] = rnd(2 * d.moe_shared_inter, d.hidden)
            w[p + "moe_shared_down"] = rnd(d.hidden, d.moe_shared_inter)
            w[p + "moe_shared_gate"] = rnd(1, d.hidden)
        else:
            w[p + "wgu"] = rnd(2 * d.inter, d.hidden)
            w[p + "wdown"] = rnd(d.hidden, d.inter)
    w["norm"] = nrm(d.hidden)
    w["lm_head"] = rnd(d.vocab, d.hidden)
    if d.has_cp_projection:
        w["cp.proj_w"] = rnd(d.cp_hidden, d.hidden)
        w["cp.proj_b"] = rnd(d.cp_hidden)
    for i in range(d.cp_layers):
        p = f"cp.l{i}."
        w[p + "ln1"] = nrm(d.cp_hidden)
        w[p + "wqkv"] = rnd(d.cp_qkv_out, d.cp_hidden)
        w[p + "qnorm"] = nrm(d.cp_head_dim)
        w[p + "knorm"] = nrm(d.cp_head_dim)
        w[p + "wo"] = rnd(d.cp_hidden, d.cp_q_heads * d.cp_head_dim)
        w[p + "ln2"] = nrm(d.cp_hidden)
        w[p + "wgu"] = rnd(2 * d.cp_inter, d.cp_hidden)
        w[p + "wdown"] = rnd(d.cp_hidden, d.cp_inter)
    w["cp.norm"] = nrm(d.cp_hidden)
    w["cp.lm_head"] = rnd(d.num_code_groups - 1, d.codebook, d.cp_hidden)
    w["cp.embed"] = rnd(d.num_code_groups - 1, d.codebook, d.hidden)
    return w


def peak_predictor_heads(d: TalkerDims, w: dict[str, torch.Tensor], gamma: float = 2.0, stream_gain: float = 20.0,
                         seed: int = 99) -> dict[str, torch.Tensor]:
    """A copy of `w` whose code-predictor heads have a CLEAR greedy winner for every input (N(0, std) heads give flat
    logits -- a quarter of all 15-group frames leave the oracle's greedy path at a <= 3-ulp tie, and a kernel bug that only moves
    low-margin picks would hide there).  Head g (the group-g logits, read at buffer position g) gets, on row perm_g[c], `gamma` times the unit
    direction of the projected INPUT embedding of code c at that position -- embed[c] for g = 1, cp.embed[g - 2][c] after that -- on top of
    its N(0, std) row, and the input projection is scaled by `stream_gain` so that the input's own direction is still a quarter of the
    residual stream behind the five layers (unscaled, the first sub-layer's output is 30 x the input and the direction is gone): the winner
    of group g is perm_g[code of group g - 1] with a margin of tens of bf16 ulps (measured on the oracle: >= 19 at 64 rows x 15 groups),
    still a function of the row's input, still through every stage of every pass, the layers still three quarters of what the head reads.
    Values stay bf16-exact (rounded once here); everything else is shared with `w`."""
    if not d.has_cp_projection:
        raise ValueError("peak_predictor_heads: built for predictors with an input projection (cp_hidden != hidden)")
    g = torch.Generator().manual_seed(seed)
    out = dict(w)
    out["cp.proj_w"] = (w["cp.proj_w"].float() * stream_gain).to(torch.bfloat16)
    head = w["cp.lm_head"].float().clone()
    for grp in range(1, d.num_code_groups):
        tab = (w["embed"][: d.codebook] if grp == 1 else w["cp.embed"][grp - 2]).float()
        # the code's OWN direction: the projection's bias and the table's mean are common to every code
        x = tab @ out["cp.proj_w"].float().T
        x = x - x.mean(dim=0, keepdim=True)
        x = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        perm = torch.randperm(d.codebook, generator=g)
        head[grp - 1, perm] += gamma * x
    out["cp.lm_head"] = head.to(torch.bfloat16)
    return out


def weight_bytes(d: TalkerDims) -> dict[str, int]:
    """Algorithmic weight bytes read once per talker step."""
    if d.moe_experts > 0:      # every expert counted once (T * k = 512 assignments over 128 experts hit ~98 % of them) + shared + router
        mlp = d.moe_experts * 3 * d.hidden * d.moe_inter + 3 * d.hidden * d.moe_shared_inter + d.moe_experts * d.hidden + d.hidden
    else:
        mlp = 3 * d.hidden * d.inter
    bb = d.layers * (d.hidden * d.qkv_out + d.q_heads * d.head_dim * d.hidden + mlp) * 2
    head = d.vocab * d.hidden * 2
    cp = d.cp_layers * (d.cp_hidden * d.cp_qkv_out + d.cp_q_heads * d.cp_head_dim * d.cp_hidden
                        + 3 * d.cp_hidden * d.cp_inter) * 2
    cp += (d.num_code_groups - 1) * d.codebook * d.cp_hidden * 2
    if d.has_cp_projection:
        cp += d.hidden * d.cp_hidden * 2
    return {"backbone": bb, "lm_head": head, "code_predictor": cp}
