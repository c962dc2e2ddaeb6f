"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np
import torch

BF16 = torch.bfloat16


def bf16_from_u16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(BF16)


def assert_bf16_close(got: torch.Tensor, ref: torch.Tensor, *, ulps: int = 1, max_mismatch: float = 0.02,
                      atol: float = 1e-3, what: str = "") -> None:
    """Both tensors hold bf16 values.  Same rounding points, different fp32 summation order:
    values agree to `atol` or `ulps` bf16 ulps, and all but `max_mismatch` of them bit-exactly."""
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    diff = (g - r).abs()
    tol = torch.maximum(torch.full_like(r, atol), r.abs() * (2.0 ** -7) * ulps * 1.01)
    bad = diff > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} values beyond {ulps} ulp; max diff {diff.max().item():.4g}"
    frac = (diff > 0).float().mean().item()
    assert frac <= max_mismatch, f"{what}: {frac:.4%} values differ (> {max_mismatch:.2%})"


def assert_f32_close(got: torch.Tensor, ref: torch.Tensor, *, atol: float = 1e-3, rtol: float = 0.0, what: str = "") -> None:
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    fin = torch.isfinite(r)
    assert torch.equal(torch.isfinite(g), fin), f"{what}: -inf pattern differs"
    diff = (g[fin] - r[fin]).abs()
    lim = atol + rtol * r[fin].abs()
    assert (diff <= lim).all(), f"{what}: max diff {diff.max().item():.4g} > {atol}"


# Round 6: the norm-fused GEMMs of the decode step (qkv, gate_up: launch path and chains alike) apply the row's rstd to the fp32 sums --
# rstd * sum(W * bf16(w r)) -- instead of normalising every operand element first -- sum(W * bf16(w * bf16(r rstd))): one bf16 rounding
# per element where the reference has two.  The HIP path is no longer a bit-faithful replay of the oracle's rounding points but a second
# bf16 pipeline of the same accuracy (the three-way statement of tests/test_gpu_parity_full.py is the gate: HIP no farther from
# fp32-activation arithmetic than the reference's own rounding is, x 1.3).  Against the bf16 oracle two such pipelines sit about one rounding
# apart: every end-to-end bound has a floor of 0.75 ulp (mean) / + 2 ulps (max) at the tensor's scale; callers' tighter figures still
# document what the bit-faithful arithmetic of rounds 1-5 measured.
E2E_MEAN_FLOOR_ULPS = 0.75
E2E_EXTRA_MAX_ULPS = 2.0


def assert_e2e_close(got: torch.Tensor, ref: torch.Tensor, *, mean_tol: float = 4e-3, max_ulps: float = 2.0,
                     what: str = "") -> None:
    """End-to-end (multi-layer) comparison of bf16 activations / logits.  A 1-ulp rounding flip
    upstream moves every downstream value by an absolute amount, so the bound is stated at the
    tensor's scale: max |diff| <= max_ulps (+ E2E_EXTRA_MAX_ULPS) bf16 ulps of the largest magnitude, mean |diff| <= mean_tol
    (or E2E_MEAN_FLOOR_ULPS ulps at that scale, whichever is larger; see the note above and DESIGN.md 'Parity')."""
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    fin = torch.isfinite(r)
    assert torch.equal(torch.isfinite(g), fin), f"{what}: -inf pattern differs"
    d = (g[fin] - r[fin]).abs()
    amax = r[fin].abs().max().item()
    ulp = 2.0 ** (int(np.floor(np.log2(max(amax, 1e-30)))) - 7)
    max_ulps = max_ulps + E2E_EXTRA_MAX_ULPS
    mean_tol = max(mean_tol, E2E_MEAN_FLOOR_ULPS * ulp)
    assert d.max().item() <= (max_ulps + 0.02) * ulp, \
        f"{what}: max diff {d.max().item():.4g} > {max_ulps} ulp ({max_ulps * ulp:.4g}) at scale {amax:.3g}"
    assert d.mean().item() <= mean_tol, f"{what}: mean diff {d.mean().item():.4g} > {mean_tol:.4g}"


def codes_on_the_oracles_frame(got: torch.Tensor, ref: torch.Tensor, cp_logits: torch.Tensor, *, tie_ulps: float = 4.0, what: str = "") -> torch.Tensor:
    """Greedy audio codes [B, Q] against the oracle's: bit-exact, except that a row may leave the oracle's greedy path at a group whose two
    best (bf16-rounded) predictor logits -- `cp_logits` [B, Q - 1, V], the ORACLE's -- are within `tie_ulps` bf16 ulps of each other: two bf16
    pipelines differ by about a rounding there -- each of the two logits by up to ~2 ulps behind five layers -- and an argmax over values that
    close is decided by it (4 ulps; 3 while the HIP path replayed the oracle's rounding points); what follows in that row reads
    another input and is not compared.  Returns the mask of the rows still on the oracle's frame; raises on any fork that is no near-tie.
    (Rounds 1-5 replayed the oracle's rounding points and tiny configurations agreed bit for bit; since round 6 -- rstd applied to the fp32
    sums, see above -- they fork where the full-size ones always did.)"""
    got, ref = got.detach().cpu().long(), ref.detach().cpu().long()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    on = torch.ones(got.shape[0], dtype=torch.bool)
    for b in (got != ref).any(1).nonzero().flatten().tolist():
        gfirst = int((got[b] != ref[b]).nonzero()[0])
        assert gfirst >= 1, f"{what}: row {b}: the layer-0 code differs"
        top = torch.topk(cp_logits[b, gfirst - 1].float(), 2).values
        tie = tie_ulps * 2.0 ** (int(np.floor(np.log2(max(float(top[0].abs()), 1e-30)))) - 7)
        assert float(top[0] - top[1]) <= tie, f"{what}: row {b}: code group {gfirst} differs without a near-tie (margin {float(top[0] - top[1]):.4g} > {tie:.4g})"
        on[b] = False
    return on


def make_moe_weights(H: int, E: int, I: int, Is: int, seed: int) -> dict:
    """Seeded bf16 weights of one sparse-MoE block (oracle.moe_block layout); the golden fixture stores only the seed,
    tests and tests/golden/make_fixtures.py both regenerate the tensors from it (CPU generator: reproducible)."""
    import torch
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)
    return {"router": rnd(E, H, scale=0.5), "gate_up": rnd(E, 2 * I, H, scale=0.08), "down": rnd(E, H, I, scale=0.08),
            "shared_gate_up": rnd(2 * Is, H, scale=0.08), "shared_down": rnd(H, Is, scale=0.08), "shared_gate": rnd(1, H, scale=0.3)}
