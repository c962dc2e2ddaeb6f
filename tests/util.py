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


def assert_e2e_close(got: torch.Tensor, ref: torch.Tensor, *, mean_tol: float = 4e-3, max_ulps: float = 2.0,
                     what: str = "") -> None:
    """End-to-end (multi-layer) comparison of bf16 activations / logits.  A 1-ulp rounding flip
    upstream moves every downstream value by an absolute amount, so the bound is stated at the
    tensor's scale: max |diff| <= 2 bf16 ulps of the largest magnitude, mean |diff| <= mean_tol
    (1e-3, the north_star tolerance, at the BASELINE weight scale; see DESIGN.md 'Parity')."""
    g, r = got.detach().float().cpu(), ref.detach().float().cpu()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    fin = torch.isfinite(r)
    assert torch.equal(torch.isfinite(g), fin), f"{what}: -inf pattern differs"
    d = (g[fin] - r[fin]).abs()
    amax = r[fin].abs().max().item()
    ulp = 2.0 ** (int(np.floor(np.log2(max(amax, 1e-30)))) - 7)
    assert d.max().item() <= (max_ulps + 0.02) * ulp, \
        f"{what}: max diff {d.max().item():.4g} > {max_ulps} ulp ({max_ulps * ulp:.4g}) at scale {amax:.3g}"
    assert d.mean().item() <= mean_tol, f"{what}: mean diff {d.mean().item():.4g} > {mean_tol}"


def make_moe_weights(H: int, E: int, I: int, Is: int, seed: int) -> dict:
    """Seeded bf16 weights of one sparse-MoE block (oracle.moe_block layout); the golden fixture stores only the seed,
    tests and tests/golden/make_fixtures.py both regenerate the tensors from it (CPU generator: reproducible)."""
    import torch
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)
    return {"router": rnd(E, H, scale=0.5), "gate_up": rnd(E, 2 * I, H, scale=0.08), "down": rnd(E, H, I, scale=0.08),
            "shared_gate_up": rnd(2 * Is, H, scale=0.08), "shared_down": rnd(H, Is, scale=0.08), "shared_gate": rnd(1, H, scale=0.3)}
