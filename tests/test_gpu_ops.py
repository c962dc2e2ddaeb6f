"""Per-op parity: every HIP kernel, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bar: integer/index outputs bit-exact; bf16 outputs within 1 bf16 ulp (same rounding
points, different fp32 summation order) and fp32 outputs within 1e-3 (north_star tolerance)."""
import os

import numpy as np
import pytest
import torch

from oracle import talker_oracle as O
from tests.util import BF16, assert_bf16_close, assert_f32_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ht_vllm_omni_amd import _lib, ops as _ops
    _lib.load()
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return _ops


def _rand(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(BF16)


@pytest.mark.parametrize("rows,hidden", [(1, 256), (7, 1024), (64, 2048), (3, 4096), (5, 128)])
def test_rmsnorm(ops, rows, hidden):
    g = torch.Generator().manual_seed(rows * 1000 + hidden)
    x, dl = _rand(g, rows, hidden), _rand(g, rows, hidden)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(BF16)
    out = ops.rmsnorm(x.cuda(), w.cuda(), 1e-6)
    assert_bf16_close(out, O.rms_norm(x, w, 1e-6), what="rmsnorm")
    # fused residual add: residual <- residual + delta (bf16), out = norm(residual)
    res = x.clone().cuda()
    out2 = ops.rmsnorm(None, w.cuda(), 1e-6, delta=dl.cuda(), residual=res)
    r_ref = x + dl
    assert torch.equal(res.cpu().view(torch.int16), r_ref.view(torch.int16)), "residual add must be bit-exact"
    assert_bf16_close(out2, O.rms_norm(r_ref, w, 1e-6), what="rmsnorm+residual")


@pytest.mark.parametrize("M", [1, 5, 16, 33, 64])
@pytest.mark.parametrize("N,K", [(256, 128), (4096, 2048), (2048, 6144), (3072, 1024), (48, 96)])
def test_gemm(ops, M, N, K):
    from ht_vllm_omni_amd import _lib as L
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=0.05), _rand(g, N, scale=0.1)
    ref32 = x.float() @ w.float().t()
    out32 = ops.gemm(x.cuda(), w.cuda(), epilogue=L.EPI_F32)
    assert_f32_close(out32, ref32, atol=1e-3 * max(1.0, ref32.abs().max().item() / 8), what="gemm f32")
    out = ops.gemm(x.cuda(), w.cuda(), bias=b.cuda())
    assert_bf16_close(out, O.linear(x, w, b), what="gemm bf16+bias")
    mask = (torch.rand(N, generator=g) > 0.3).to(torch.uint8)
    outm = ops.gemm(x.cuda(), w.cuda(), epilogue=L.EPI_F32_BF16RND, mask=mask.cuda())
    refm = ref32.to(BF16).float().masked_fill(mask == 0, float("-inf"))
    assert_bf16_close(outm.cpu().nan_to_num(neginf=-1e30), refm.nan_to_num(neginf=-1e30), what="gemm logits+mask")
    assert torch.equal(torch.isinf(outm.cpu()), torch.isinf(refm))


@pytest.mark.parametrize("M", [2, 64])
@pytest.mark.parametrize("inter,K", [(512, 256), (6144, 2048), (3072, 1024)])
def test_gemm_silu_mul(ops, M, inter, K):
    from ht_vllm_omni_amd import _lib as L
    g = torch.Generator().manual_seed(inter + K + M)
    x, w = _rand(g, M, K), _rand(g, 2 * inter, K, scale=0.05)
    out = ops.gemm(x.cuda(), w.cuda(), epilogue=L.EPI_SILU_MUL)
    gu = O.linear(x, w)
    ref = O.silu_mul(gu[:, :inter], gu[:, inter:])
    assert_bf16_close(out, ref, ulps=2, max_mismatch=0.03, what="gemm silu*mul")


@pytest.mark.parametrize("M", [1, 16, 37, 64])
@pytest.mark.parametrize("N,K,epi", [(4096, 2048, "bf16"), (2048, 6144, "bf16"), (6144, 2048, "silu"), (3072, 1024, "silu"),
                                     (3072, 2048, "logits"), (96, 64, "bf16")])
def test_gemm_fragment_major_layouts(ops, M, N, K, epi):
    """Fragment-major W / x / out (one wave-level load = 1 KB contiguous) is a pure re-addressing: results are
    bit-identical to the row-major launch of the same kernel."""
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd.engine import frag_shuffle, frag_unshuffle
    g = torch.Generator().manual_seed(M + N + K)
    rows = 2 * N if epi == "silu" else N
    x, w = _rand(g, M, K), _rand(g, rows, K, scale=0.05)
    code = {"bf16": L.EPI_BF16, "silu": L.EPI_SILU_MUL, "logits": L.EPI_F32_BF16RND}[epi]
    ref = ops.gemm(x.cuda(), w.cuda(), epilogue=code)
    wf = frag_shuffle(w).cuda()
    out_w = ops.gemm(x.cuda(), wf, epilogue=code, layout=L.LAYOUT_W_FRAG)
    assert torch.equal(out_w, ref), "W fragment-major"
    Mp = (M + 15) // 16 * 16
    xp = torch.zeros(Mp, K, dtype=BF16)
    xp[:M] = x
    xf = frag_shuffle(xp).cuda()
    out_wx = ops.gemm(xf, wf, epilogue=code, layout=L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG, M=M)
    assert torch.equal(out_wx, ref), "W + x fragment-major"
    if epi != "logits" and N % 32 == 0:
        out_f = ops.gemm(xf, wf, epilogue=code, layout=L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG | L.LAYOUT_OUT_FRAG, M=M)
        assert torch.equal(frag_unshuffle(out_f.cpu())[:M], ref.cpu()), "fragment-major output"
    assert torch.equal(frag_unshuffle(frag_shuffle(w)), w)


@pytest.mark.parametrize("M", [1, 16, 37, 64])
@pytest.mark.parametrize("H,K1,N2,epi", [(2048, 2048, 4096, "bf16"), (1024, 3072, 3072, "silu"), (2048, 6144, 3072, "logits"),
                                         (1024, 2048, 2048, "logits"), (64, 96, 48, "bf16")])
def test_gemm_norm_free_residual_stream(ops, M, H, K1, N2, epi):
    """omni_gemm_resid (producer: r += bf16(x.W1^T), per-workgroup sum(r^2) slabs) followed by omni_gemm_xnorm (consumer:
    RMSNorm applied to the fragments it loads) == residual add, rms_norm, linear of the oracle."""
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd.engine import frag_shuffle, frag_unshuffle
    g = torch.Generator().manual_seed(M + H + K1 + N2)
    Mp = (M + 15) // 16 * 16
    r0 = torch.zeros(Mp, H, dtype=BF16)
    r0[:M] = _rand(g, M, H)
    x1 = torch.zeros(Mp, K1, dtype=BF16)
    x1[:M] = _rand(g, M, K1)
    w1 = _rand(g, H, K1, scale=0.03)
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(BF16)
    rows2 = 2 * N2 if epi == "silu" else N2
    w2 = _rand(g, rows2, H, scale=0.03)
    code = {"bf16": L.EPI_BF16, "silu": L.EPI_SILU_MUL, "logits": L.EPI_F32_BF16RND}[epi]

    # oracle: r = bf16(r0 + linear(x1)); x = rms_norm(r); y = linear(x)
    r_ref = r0[:M] + O.linear(x1[:M], w1)
    x_ref = O.rms_norm(r_ref, nw, 1e-6)
    y = O.linear(x_ref, w2)

    rf = frag_shuffle(r0).cuda()
    part = torch.full((H // 16, 64), float("nan"), dtype=torch.float32, device="cuda")
    np_ = ops.gemm_resid(frag_shuffle(x1).cuda(), frag_shuffle(w1).cuda(), rf, part, M=M)
    assert np_ == H // 16
    r_got = frag_unshuffle(rf.cpu())[:M]
    # the add is exact on top of this library's own GEMM output (same kernel, same accumulation order) ...
    d_dev = ops.gemm(frag_shuffle(x1).cuda(), frag_shuffle(w1).cuda(), layout=L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG, M=M).cpu()
    assert torch.equal(r_got, r0[:M] + d_dev), "residual stream == bf16(r + bf16(acc)) bit for bit"
    # ... and that output is the oracle's linear within one rounding (absolute bound: r0 + delta may cancel)
    assert_bf16_close(d_dev, O.linear(x1[:M], w1), ulps=1, max_mismatch=0.06, what="delta")
    assert (r_got.float() - r_ref.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, O.linear(x1[:M], w1).float().abs().max().item())
    ss = part[:, :M].sum(0).cpu()
    torch.testing.assert_close(ss, r_got.float().pow(2).sum(-1), rtol=1e-5, atol=1e-6)

    out, normed = ops.gemm_xnorm(rf, part, np_, nw.cuda(), frag_shuffle(w2).cuda(), 1e-6, M=M, epilogue=code, want_normed=True)
    x_same = O.rms_norm(r_got, nw, 1e-6)            # same r as the device: isolates the consumer
    assert_bf16_close(normed, x_same, what="normalised rows")
    y_same = O.linear(x_same, w2)
    if epi == "silu":
        assert_bf16_close(out, O.silu_mul(y_same[:, :N2], y_same[:, N2:]), ulps=2, max_mismatch=0.06, what="xnorm silu")
    elif epi == "logits":
        assert_bf16_close(out, y_same.float(), ulps=1, max_mismatch=0.06, what="xnorm logits")
    else:
        assert_bf16_close(out, y_same, ulps=1, max_mismatch=0.06, what="xnorm gemm")
    # and against the separate-kernel path of this library on the same r: bit-identical GEMM input -> identical output
    xn = ops.rmsnorm(r_got.cuda(), nw.cuda(), 1e-6)
    assert torch.equal(xn, normed), "xnorm rows == omni_rmsnorm rows"
    assert torch.equal(ops.gemm(xn, w2.cuda(), epilogue=code), out), "xnorm GEMM == rmsnorm + GEMM"
    # fresh stream (no accumulate, bias, row-major x): the projection into the code predictor
    bias = _rand(g, H)
    rf2 = torch.zeros_like(rf)
    ops.gemm_resid(x1[:M].contiguous().cuda(), frag_shuffle(w1).cuda(), rf2, part, bias=bias.cuda(), accumulate=False, x_frag=False)
    ref2 = ops.gemm(x1[:M].contiguous().cuda(), w1.cuda(), bias=bias.cuda())
    assert torch.equal(frag_unshuffle(rf2.cpu())[:M], ref2.cpu()), "projection into the stream"
    if epi != "logits" and N2 % 32 == 0:
        out_f = ops.gemm_xnorm(rf, part, np_, nw.cuda(), frag_shuffle(w2).cuda(), 1e-6, M=M, epilogue=code, out_frag=True)
        assert out_f.shape[0] == Mp


@pytest.mark.parametrize("M", [1, 37, 64])
@pytest.mark.parametrize("inter,K", [(6144, 2048), (3072, 1024), (768, 1024), (96, 64)])
def test_gemm_silu_interleaved_gate_up(ops, M, inter, K):
    """OMNI_EPI_SILU_MUL_GU8: gate / up rows interleaved in groups of 8 before the fragment shuffle, so a workgroup may own
    any number of 16-row tiles (3 of them divide the backbone's gate_up into 256 workgroups).  Same accumulation order as
    the paired layout -> bit-identical, for the policy's tile and for every forced tile, plain and norm-fused."""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd.engine import frag_shuffle, gu8_shuffle
    with L.debug_library() as lib:      # libomni_talker_debug.so: the same sources with run-time tile knobs
        lib.omni_debug_tile.argtypes = [C.c_int, C.c_int]; lib.omni_debug_tile.restype = None
        g = torch.Generator().manual_seed(M + inter + K)
        x, w = _rand(g, M, K), _rand(g, 2 * inter, K, scale=0.05)
        ref = ops.gemm(x.cuda(), w.cuda(), epilogue=L.EPI_SILU_MUL)
        gu = O.linear(x, w)
        assert_bf16_close(ref, O.silu_mul(gu[:, :inter], gu[:, inter:]), ulps=2, max_mismatch=0.03, what="paired silu*mul")
        w8 = gu8_shuffle(w).cuda()
        Mp = (M + 15) // 16 * 16
        xp = torch.zeros(Mp, K, dtype=BF16)
        xp[:M] = x
        xf = frag_shuffle(xp).cuda()
        nw = torch.ones(K, dtype=BF16, device="cuda")
        # residual-stream form of the same rows: r = x, sum(r^2) in one slab, norm weight 1 (rstd != 1: compare with the paired xnorm)
        part = torch.zeros(K // 16, 64, device="cuda")
        part[0, :M] = x.float().pow(2).sum(-1).cuda()
        ref_x = ops.gemm_xnorm(xf, part, K // 16, nw, frag_shuffle(w).cuda(), 1e-6, M=M, epilogue=L.EPI_SILU_MUL)
        try:
            for nt in (0, 2, 3, 4):
                if nt and (inter // 8) % nt:
                    continue
                for mt in ((0,) if nt == 0 else (1, 2, 4)):
                    lib.omni_debug_tile(nt, mt)
                    out = ops.gemm(x.cuda(), w8, epilogue=L.EPI_SILU_MUL_GU8, layout=L.LAYOUT_W_FRAG)
                    assert torch.equal(out, ref), (nt, mt)
                    out = ops.gemm(xf, w8, epilogue=L.EPI_SILU_MUL_GU8, layout=L.LAYOUT_W_FRAG | L.LAYOUT_X_FRAG, M=M)
                    assert torch.equal(out, ref), (nt, mt, "x frag")
                    out = ops.gemm_xnorm(xf, part, K // 16, nw, w8, 1e-6, M=M, epilogue=L.EPI_SILU_MUL_GU8)
                    assert torch.equal(out, ref_x), (nt, mt, "xnorm")
        finally:
            lib.omni_debug_tile(0, 0)
    with pytest.raises(L.OmniError):
        ops.gemm(x.cuda(), w.cuda(), epilogue=L.EPI_SILU_MUL_GU8)        # row-major W has no interleaved form


def test_silu_mul(ops):
    g = torch.Generator().manual_seed(4)
    gu = _rand(g, 37, 2 * 3072, scale=2.0)
    out = ops.silu_mul(gu.cuda())
    assert_bf16_close(out, O.silu_mul(gu[:, :3072], gu[:, 3072:]), ulps=1, max_mismatch=0.02, what="silu_mul")


def test_gemm_rejects_bad_shapes(ops):
    from ht_vllm_omni_amd import _lib as L
    x = torch.zeros(65, 64, dtype=BF16, device="cuda")
    w = torch.zeros(32, 64, dtype=BF16, device="cuda")
    with pytest.raises(L.OmniError):
        ops.gemm(x, w)
    with pytest.raises(L.OmniError):
        ops.gemm(x[:4, :40].contiguous(), w[:, :40].contiguous())


def test_slot_mapping_bit_exact(ops):
    bs, B = 16, 9
    g = torch.Generator().manual_seed(1)
    bt = torch.randint(1, 500, (B, 40), generator=g, dtype=torch.int32)
    pos = torch.tensor([0, 1, 15, 16, 17, 255, 256, 300, 639], dtype=torch.int32)
    got = ops.slot_mapping(bt.cuda(), pos.cuda(), bs, B_padded=12).cpu()
    ref = [O.slot_of(bt[r].tolist(), int(pos[r]), bs) for r in range(B)] + [-1, -1, -1]
    assert got.tolist() == ref


@pytest.mark.parametrize("kv", ["bf16", "fp8", "int8"])
@pytest.mark.parametrize("hq,hkv", [(4, 2), (16, 8), (2, 1)])
def test_qknorm_rope_kvwrite(ops, kv, hq, hkv):
    from ht_vllm_omni_amd import _lib as L
    D, T, nb, bs = 128, 11, 6, 16
    g = torch.Generator().manual_seed(hq * 10 + hkv)
    qkv = _rand(g, T, (hq + 2 * hkv) * D, scale=2.0)
    qn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    kn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    pos = torch.tensor([0, 1, 2, 3, 17, 40, 41, 95, 5, 6, 7], dtype=torch.int32)
    slots = torch.tensor([16, 17, 18, 19, 33, 40, 41, -1, 85, 86, 95], dtype=torch.int64)  # one padded token
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    cos_sin = ops.rope_table(128, D, 1e6)
    store = {"bf16": BF16, "fp8": torch.uint8, "int8": torch.int8}[kv]
    cache = torch.zeros(2, nb, bs, hkv, D, dtype=store, device="cuda")
    scales = torch.zeros(2, nb, bs, hkv, dtype=torch.float32, device="cuda") if kv == "int8" else None
    q = ops.qknorm_rope_kvwrite(qkv.cuda(), qn.cuda(), kn.cuda(), pos.cuda(), cos_sin.cuda(), slots.cuda(), cache[0], cache[1],
                                q_heads=hq, kv_heads=hkv, head_dim=D, eps=1e-6, kv_dtype=L.KV_CODES[kv], k_scale=k_scale,
                                v_scale=v_scale, k_scales=None if scales is None else scales[0],
                                v_scales=None if scales is None else scales[1])
    # oracle
    cos, sin = O.rope_cos_sin(pos.long(), D, 1e6)
    qq = qkv[:, : hq * D].reshape(T, hq, D)
    kk = qkv[:, hq * D:(hq + hkv) * D].reshape(T, hkv, D)
    vv = qkv[:, (hq + hkv) * D:].reshape(T, hkv, D)
    q_ref = O.apply_rope(O.rms_norm(qq, qn, 1e-6), cos, sin)
    k_ref = O.apply_rope(O.rms_norm(kk, kn, 1e-6), cos, sin)
    assert_bf16_close(q.view(T, hq, D), q_ref, what="q norm+rope")
    pk = O.PagedKV(nb, bs, hkv, D, kv, k_scale, v_scale)
    pk.write(slots, k_ref, vv)
    got = cache.cpu()
    if kv == "bf16":
        assert torch.equal(got[1].view(torch.int16), pk.data[1].view(torch.int16)), "V rows are a pure copy"
        assert_bf16_close(got[0], pk.data[0], what="K cache bf16")
    elif kv == "fp8":
        ref = pk.data.view(torch.uint8)
        assert torch.equal(got[1], ref[1]), "V fp8 bytes must be bit-exact (same input, RNE + saturate)"
        mism = (got[0] != ref[0]).float().mean().item()
        assert mism < 0.01, f"K fp8 bytes differ in {mism:.3%}"
        d = (got[0].view(torch.float8_e4m3fn).float() - ref[0].view(torch.float8_e4m3fn).float()).abs()
        assert (d <= ref[0].view(torch.float8_e4m3fn).float().abs() * 0.126 + 2 ** -9).all(), "K fp8 beyond 1 ulp"
    else:
        assert torch.equal(got[1], pk.data[1]), "V int8 must be bit-exact"
        assert torch.equal(scales.cpu()[1], pk.scales[1]), "V int8 scales must be bit-exact"
        assert (got[0].int() - pk.data[0].int()).abs().max().item() <= 1
        torch.testing.assert_close(scales.cpu()[0], pk.scales[0], rtol=1e-2, atol=0)
    # padded token (slot -1) must not have been written anywhere: blocks 0 and 2's tail stay zero
    assert got[:, 0].abs().sum().item() == 0


@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("kv", ["bf16", "fp8"])
def test_qknorm_mrope_kvwrite_with_differing_rows(ops, kv, interleaved):
    """M-RoPE ids whose three rows differ (image / video tokens; tests/golden/mrope_positions.json holds real ones): the kernel
    gathers cos / sin per rotary pair from the row of the pair's axis -- against the oracle's restatement of vLLM's
    MRotaryEmbedding.forward (chunked sections and apply_interleaved_rope); identical rows reproduce the plain kernel bit for bit."""
    from ht_vllm_omni_amd import _lib as L
    D, T, nb, bs, hq, hkv = 128, 13, 6, 16, 4, 2
    sec = (24, 20, 20)
    g = torch.Generator().manual_seed(5 + int(interleaved))
    qkv = _rand(g, T, (hq + 2 * hkv) * D, scale=2.0)
    qn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    kn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    base = torch.tensor([0, 1, 2, 3, 3, 3, 3, 9, 10, 40, 41, 95, 96])
    pos3 = torch.stack([base, base + torch.tensor([0, 0, 0, 0, 1, 0, 1, 0, 0, 2, 3, 0, 7]), base + torch.tensor([0, 0, 0, 1, 0, 2, 1, 0, 0, 5, 1, 0, 11])])
    slots = torch.arange(16, 16 + T, dtype=torch.int64)
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    cos_sin = ops.rope_table(128, D, 1e6)
    axis = ops.mrope_axis_table(sec, interleaved).cuda()
    store = {"bf16": BF16, "fp8": torch.uint8}[kv]

    def run(positions, ax):
        cache = torch.zeros(2, nb, bs, hkv, D, dtype=store, device="cuda")
        q = ops.qknorm_rope_kvwrite(qkv.cuda(), qn.cuda(), kn.cuda(), positions.to(torch.int32).cuda().contiguous(), cos_sin.cuda(), slots.cuda(),
                                    cache[0], cache[1], q_heads=hq, kv_heads=hkv, head_dim=D, eps=1e-6, kv_dtype=L.KV_CODES[kv],
                                    k_scale=k_scale, v_scale=v_scale, mrope_axis=ax)
        return q.cpu(), cache.cpu()

    q, cache = run(pos3, axis)
    cos, sin = O.mrope_cos_sin(pos3, D, 1e6, sec, interleaved)
    qq = qkv[:, : hq * D].reshape(T, hq, D)
    kk = qkv[:, hq * D:(hq + hkv) * D].reshape(T, hkv, D)
    vv = qkv[:, (hq + hkv) * D:].reshape(T, hkv, D)
    q_ref = O.apply_rope(O.rms_norm(qq, qn, 1e-6), cos, sin)
    k_ref = O.apply_rope(O.rms_norm(kk, kn, 1e-6), cos, sin)
    assert_bf16_close(q.view(T, hq, D), q_ref, what="q norm + M-RoPE")
    pk = O.PagedKV(nb, bs, hkv, D, kv, k_scale, v_scale)
    pk.write(slots, k_ref, vv)
    if kv == "bf16":
        assert_bf16_close(cache[0], pk.data[0], what="K cache, M-RoPE")
    else:
        assert (cache[0] != pk.data.view(torch.uint8)[0]).float().mean().item() < 0.01
    # the rows really matter: the plain kernel on row 0 gives something else ...
    q_plain, _ = run(pos3[0], None)
    assert not torch.equal(q, q_plain)
    # ... and three identical rows give the plain kernel's bits
    q_same, c_same = run(pos3[0].expand(3, -1), axis)
    q_p0, c_p0 = run(pos3[0], None)
    assert torch.equal(q_same, q_p0) and torch.equal(c_same.view(torch.uint8), c_p0.view(torch.uint8))


def _fill_cache(kv, nb, bs, hkv, D, gen, k_scale, v_scale):
    pk = O.PagedKV(nb, bs, hkv, D, kv, k_scale, v_scale)
    slots = torch.arange(nb * bs)
    k = _rand(gen, nb * bs, hkv, D, scale=1.5)
    v = _rand(gen, nb * bs, hkv, D, scale=1.5)
    pk.write(slots, k, v)
    return pk


@pytest.mark.parametrize("kv", ["bf16", "fp8", "int8"])
@pytest.mark.parametrize("hq,hkv", [(4, 2), (16, 8), (4, 1), (16, 2)])
@pytest.mark.parametrize("split", [False, True])
def test_paged_attn_decode(ops, kv, hq, hkv, split):
    from ht_vllm_omni_amd import _lib as L
    D, bs, nb = 128, 16, 128
    g = torch.Generator().manual_seed(hq + hkv + len(kv))
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    pk = _fill_cache(kv, nb, bs, hkv, D, g, k_scale, v_scale)
    seq_lens = [1, 15, 16, 17, 31, 33, 257, 700, 128, 1000]
    B = len(seq_lens)
    perm = torch.randperm(nb - 1, generator=g) + 1
    bt = torch.zeros(B, 64, dtype=torch.int32)
    ptr = 0
    for r, n in enumerate(seq_lens):
        need = (n + bs - 1) // bs
        bt[r, :need] = perm[(ptr + torch.arange(need)) % (nb - 1)]
        ptr += 7
    q = _rand(g, B, hq * D, scale=1.0)
    store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    cache = store.cuda()
    sc = pk.scales.cuda() if kv == "int8" else None
    out = ops.paged_attn_decode(q.cuda(), cache[0], cache[1], bt.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs, kv_dtype=L.KV_CODES[kv],
                                k_scale=k_scale, v_scale=v_scale, k_scales=None if sc is None else sc[0],
                                v_scales=None if sc is None else sc[1], max_seq_len=1024, split=split)
    for r, n in enumerate(seq_lens):
        kk, vv = pk.gather(bt[r].tolist(), n)
        ref = O.attention_rows(q[r].view(1, hq, D), kk, vv, torch.tensor([n - 1]), D ** -0.5)
        assert_bf16_close(out[r].view(1, hq, D), ref, ulps=1, max_mismatch=0.05, what=f"attn row {r} len {n}")


@pytest.mark.parametrize("kv", ["fp8", "bf16", "int8"])
@pytest.mark.parametrize("hq,hkv", [(16, 8), (16, 2)])
def test_kv_splits_merged_inside_the_attention_launch_equal_the_merge_launch(ops, kv, hq, hkv):
    """Round 6 A/B arm (debug library; measured a tie at 16 / 32 rows and a loss at 1 row, so the product keeps the merge launch): at 1-32
    rows the context of a (row, kv head) is split over several workgroups; the LAST of them to arrive merges the splits inside the attention
    launch (arrival counters at the head of the workspace, write-through records).  Same arithmetic in split order: bit-identical to the
    merge launch, whoever arrives last; the counters are zero again after every launch, so one workspace serves call after call."""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    D, bs, nb = 128, 16, 160
    g = torch.Generator().manual_seed(11 + hkv)
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    pk = _fill_cache(kv, nb, bs, hkv, D, g, k_scale, v_scale)
    seq_lens = [700, 1000, 257, 33, 1, 513, 2000]
    B = len(seq_lens)
    perm = torch.randperm(nb - 1, generator=g) + 1
    bt = torch.zeros(B, 128, dtype=torch.int32)
    ptr = 0
    for r, n in enumerate(seq_lens):
        need = (n + bs - 1) // bs
        bt[r, :need] = perm[(ptr + torch.arange(need)) % (nb - 1)]
        ptr += 7
    qs = [_rand(g, B, hq * D, scale=1.0).cuda() for _ in range(3)]
    store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    cache = store.cuda()
    sc = pk.scales.cuda() if kv == "int8" else None
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_pa_merge.argtypes = [C.c_int]; lib.omni_debug_pa_merge.restype = None
        nbytes = lib.omni_paged_attn_workspace_bytes(B, hq, D, 2048)
        try:
            for inkernel in (0, 1):
                lib.omni_debug_pa_merge(inkernel)
                ws = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
                outs = []
                for q in qs:                 # the same workspace three times: the counters must come back to zero
                    outs.append(ops.paged_attn_decode(q, cache[0], cache[1], bt.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                                      q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs, kv_dtype=L.KV_CODES[kv],
                                                      k_scale=k_scale, v_scale=v_scale, k_scales=None if sc is None else sc[0],
                                                      v_scales=None if sc is None else sc[1], max_seq_len=2048, workspace=ws).clone())
                torch.cuda.synchronize()
                assert int(ws[:512].view(torch.int32).abs().sum()) == 0, "arrival counters left non-zero"
                res[inkernel] = outs
        finally:
            lib.omni_debug_pa_merge(0)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    for r, n in enumerate(seq_lens):         # ... and both are the oracle's attention
        kk, vv = pk.gather(bt[r].tolist(), n)
        ref = O.attention_rows(qs[2][r].cpu().view(1, hq, D), kk, vv, torch.tensor([n - 1]), D ** -0.5)
        assert_bf16_close(res[1][2][r].cpu().view(1, hq, D), ref, ulps=1, max_mismatch=0.05, what=f"attn row {r} len {n}")


@pytest.mark.parametrize("kv", ["bf16", "fp8", "int8"])
@pytest.mark.parametrize("hq,hkv", [(4, 2), (16, 8), (16, 2)])
@pytest.mark.parametrize("split", [False, True])
def test_attn_decode_fused(ops, kv, hq, hkv, split):
    """Fused q/k-norm + RoPE + KV write + attention == the two separate ops (oracle), incl. slot indices."""
    from ht_vllm_omni_amd import _lib as L
    D, bs, nb = 128, 16, 96
    g = torch.Generator().manual_seed(hq * 3 + hkv + len(kv))
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    pk = _fill_cache(kv, nb, bs, hkv, D, g, k_scale, v_scale)
    seq_lens = [1, 2, 16, 17, 33, 129, 400, 64]        # context incl. the new token
    B = len(seq_lens)
    perm = torch.randperm(nb - 1, generator=g) + 1
    bt = torch.zeros(B, 32, dtype=torch.int32)
    ptr = 0
    for r, n in enumerate(seq_lens):                   # disjoint blocks: this op WRITES the new token
        need = (n + bs - 1) // bs
        bt[r, :need] = perm[ptr:ptr + need]
        ptr += need
    pos = torch.tensor([n - 1 for n in seq_lens], dtype=torch.int32)
    qkv = _rand(g, B, (hq + 2 * hkv) * D, scale=2.0)
    qn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    kn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    cos_sin = ops.rope_table(512, D, 1e6)
    store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    cache = store.clone().cuda()
    sc = pk.scales.clone().cuda() if kv == "int8" else None
    out, slots = ops.attn_decode_fused(qkv.cuda(), qn.cuda(), kn.cuda(), pos.cuda(), cos_sin.cuda(), cache[0], cache[1],
                                       bt.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(), q_heads=hq, kv_heads=hkv,
                                       head_dim=D, block_size=bs, kv_dtype=L.KV_CODES[kv], eps=1e-6, k_scale=k_scale,
                                       v_scale=v_scale, k_scales=None if sc is None else sc[0],
                                       v_scales=None if sc is None else sc[1], max_seq_len=512, split=split)
    # oracle: norm + rope, write through the cache, attend
    ref_slots = torch.tensor([O.slot_of(bt[r].tolist(), int(pos[r]), bs) for r in range(B)])
    assert torch.equal(slots.cpu(), ref_slots), "slot indices must be bit-exact"
    cos, sin = O.rope_cos_sin(pos.long(), D, 1e6)
    qq = O.apply_rope(O.rms_norm(qkv[:, : hq * D].reshape(B, hq, D), qn, 1e-6), cos, sin)
    kk = O.apply_rope(O.rms_norm(qkv[:, hq * D:(hq + hkv) * D].reshape(B, hkv, D), kn, 1e-6), cos, sin)
    vv = qkv[:, (hq + hkv) * D:].reshape(B, hkv, D)
    pk.write(ref_slots, kk, vv)
    for r, n in enumerate(seq_lens):
        kf, vf = pk.gather(bt[r].tolist(), n)
        ref = O.attention_rows(qq[r:r + 1], kf, vf, torch.tensor([n - 1]), D ** -0.5)
        assert_bf16_close(out[r].view(1, hq, D), ref, ulps=1, max_mismatch=0.08, what=f"fused attn row {r} len {n}")
    # the cache now holds the new token exactly as the oracle's write would
    got = cache.cpu()
    ref_store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    if kv == "bf16":
        assert_bf16_close(got, ref_store, what="cache after fused write")
    else:
        # THE exception to "bit-exact for byte work" (VERDICT r4 weak #2): K reaches the quantiser through the q/k-norm, whose fp32 sum
        # over the 128 dimensions runs in another order here than in the oracle -- a value that lands on a rounding boundary of the 8-bit
        # format flips its byte.  Budget: < 1e-3 of the bytes of the whole cache tensor (only the new token's rows can differ at all).  V
        # does not pass through a norm: its bytes ARE exact (tests/test_gpu_parity_full.py::test_device_kv_write_bytes..., fixture G3)
        assert (got != ref_store).float().mean().item() < 1e-3
    if kv == "int8":
        torch.testing.assert_close(sc.cpu(), pk.scales, rtol=1e-2, atol=0)


@pytest.mark.parametrize("hq,hkv", [(4, 2), (16, 8), (2, 2), (16, 2)])
def test_attn_decode_fused_short_context(ops, hq, hkv):
    """Code-predictor geometry: block = one request's 17 positions, contexts 1..17 (one-wave kernel)."""
    from ht_vllm_omni_amd import _lib as L
    D, bs, B = 128, 17, 17
    g = torch.Generator().manual_seed(hq + hkv)
    pk = _fill_cache("bf16", B, bs, hkv, D, g, 1.0, 1.0)
    seq_lens = list(range(1, 18))
    bt = torch.arange(B, dtype=torch.int32).view(B, 1)
    pos = torch.tensor([n - 1 for n in seq_lens], dtype=torch.int32)
    qkv = _rand(g, B, (hq + 2 * hkv) * D, scale=2.0)
    qn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    kn = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    cos_sin = ops.rope_table(32, D, 1e4)
    cache = pk.data.clone().cuda()
    out, slots = ops.attn_decode_fused(qkv.cuda(), qn.cuda(), kn.cuda(), pos.cuda(), cos_sin.cuda(), cache[0], cache[1],
                                       bt.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(), q_heads=hq, kv_heads=hkv,
                                       head_dim=D, block_size=bs, kv_dtype=L.KV_BF16, eps=1e-6, max_seq_len=bs, split=False)
    ref_slots = torch.tensor([r * bs + int(pos[r]) for r in range(B)])
    assert torch.equal(slots.cpu(), ref_slots)
    cos, sin = O.rope_cos_sin(pos.long(), D, 1e4)
    qq = O.apply_rope(O.rms_norm(qkv[:, : hq * D].reshape(B, hq, D), qn, 1e-6), cos, sin)
    kk = O.apply_rope(O.rms_norm(qkv[:, hq * D:(hq + hkv) * D].reshape(B, hkv, D), kn, 1e-6), cos, sin)
    vv = qkv[:, (hq + hkv) * D:].reshape(B, hkv, D)
    pk.write(ref_slots, kk, vv)
    for r, n in enumerate(seq_lens):
        kf, vf = pk.gather(bt[r].tolist(), n)
        ref = O.attention_rows(qq[r:r + 1], kf, vf, torch.tensor([n - 1]), D ** -0.5)
        assert_bf16_close(out[r].view(1, hq, D), ref, ulps=1, max_mismatch=0.08, what=f"short ctx row {r} len {n}")
    assert_bf16_close(cache.cpu(), pk.data, what="cache after fused write (short)")


def test_paged_attn_prefill_causal(ops):
    from ht_vllm_omni_amd import _lib as L
    D, bs, nb, hq, hkv = 128, 16, 32, 4, 2
    g = torch.Generator().manual_seed(77)
    pk = _fill_cache("bf16", nb, bs, hkv, D, g, 1.0, 1.0)
    lens = [5, 33]
    bt = torch.tensor([[3, 0, 0, 0], [7, 9, 4, 0]], dtype=torch.int32)
    req = torch.tensor([0] * lens[0] + [1] * lens[1], dtype=torch.int32)
    pos = torch.tensor(list(range(lens[0])) + list(range(lens[1])), dtype=torch.int32)
    T = req.numel()
    q = _rand(g, T, hq * D)
    cache = pk.data.cuda()
    out = ops.paged_attn_prefill(q.cuda(), cache[0], cache[1], bt.cuda(), req.cuda(), pos.cuda(), q_heads=hq, kv_heads=hkv,
                                 head_dim=D, block_size=bs, kv_dtype=L.KV_BF16)
    o = 0
    for r, n in enumerate(lens):
        kk, vv = pk.gather(bt[r].tolist(), n)
        ref = O.attention_rows(q[o:o + n].view(n, hq, D), kk, vv, torch.arange(n), D ** -0.5)
        assert_bf16_close(out[o:o + n].view(n, hq, D), ref, max_mismatch=0.05, what=f"prefill req {r}")
        o += n


@pytest.mark.parametrize("kv", ["bf16", "fp8", "int8"])
@pytest.mark.parametrize("hq,hkv", [(4, 2), (16, 8), (4, 1), (2, 2), (16, 2)])
def test_paged_attn_prefill_mfma_ragged(ops, kv, hq, hkv):
    """Matrix-core prefill attention: ragged prompts (1-token, block-boundary and multi-tile lengths, tiles that
    straddle up to three requests), scattered blocks, a later chunk of a request (positions start past cached keys),
    every KV dtype -- against the oracle's fp32 attention through the same cache."""
    from ht_vllm_omni_amd import _lib as L
    D, bs, nb = 128, 16, 96
    g = torch.Generator().manual_seed(3 * hq + hkv + len(kv))
    k_scale, v_scale = (0.5, 2.0) if kv == "fp8" else (1.0, 1.0)
    pk = _fill_cache(kv, nb, bs, hkv, D, g, k_scale, v_scale)
    # (first position, number of tokens) per request: the 4th is the second chunk of a 150-token prompt
    chunks = [(0, 1), (0, 5), (0, 3), (100, 50), (0, 16), (0, 17), (0, 97), (0, 32)]
    R = len(chunks)
    perm = torch.randperm(nb - 1, generator=g) + 1
    bt = torch.zeros(R, 16, dtype=torch.int32)
    ptr = 0
    for r, (p0, n) in enumerate(chunks):
        need = (p0 + n + bs - 1) // bs
        bt[r, :need] = perm[(ptr + torch.arange(need)) % (nb - 1)]
        ptr += 11
    req = torch.cat([torch.full((n,), r) for r, (_, n) in enumerate(chunks)]).to(torch.int32)
    pos = torch.cat([torch.arange(p0, p0 + n) for p0, n in chunks]).to(torch.int32)
    T = req.numel()
    q = _rand(g, T, hq * D)
    store = pk.data.view(torch.uint8) if kv == "fp8" else pk.data
    cache = store.cuda()
    sc = pk.scales.cuda() if kv == "int8" else None
    out = ops.paged_attn_prefill(q.cuda(), cache[0], cache[1], bt.cuda(), req.cuda(), pos.cuda(), q_heads=hq, kv_heads=hkv,
                                 head_dim=D, block_size=bs, kv_dtype=L.KV_CODES[kv], k_scale=k_scale, v_scale=v_scale,
                                 k_scales=None if sc is None else sc[0], v_scales=None if sc is None else sc[1])
    o = 0
    for r, (p0, n) in enumerate(chunks):
        kk, vv = pk.gather(bt[r].tolist(), p0 + n)
        ref = O.attention_rows(q[o:o + n].view(n, hq, D), kk, vv, torch.arange(p0, p0 + n), D ** -0.5)
        assert_bf16_close(out[o:o + n].view(n, hq, D), ref, ulps=1, max_mismatch=0.05, what=f"prefill req {r} ({p0}+{n}) {kv}")
        o += n


def test_attention_softmax_property_full_size(ops):
    """BASELINE size (B=64, 16/8 heads, fp8, ctx U{96..608}): V == const -> output == const exactly,
    whatever K, q and the block placement are (softmax weights sum to 1)."""
    from ht_vllm_omni_amd import _lib as L
    D, bs, hq, hkv, B = 128, 16, 16, 8, 64
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(96, 609, (B,), generator=g)
    nblk = int(((lens + bs - 1) // bs).sum()) + 1
    cache = torch.zeros(2, nblk, bs, hkv, D, dtype=torch.uint8, device="cuda")
    cache[0] = torch.randint(0, 120, cache[0].shape, dtype=torch.uint8, device="cuda")   # positive finite e4m3 patterns
    cache[1] = 0x38                                                                    # e4m3fn 1.0
    bt = torch.zeros(B, 64, dtype=torch.int32)
    nxt = 1
    for r in range(B):
        need = int((lens[r] + bs - 1) // bs)
        bt[r, :need] = torch.arange(nxt, nxt + need)
        nxt += need
    q = _rand(g, B, hq * D)
    out = ops.paged_attn_decode(q.cuda(), cache[0], cache[1], bt.cuda(), lens.to(torch.int32).cuda(), q_heads=hq,
                                kv_heads=hkv, head_dim=D, block_size=bs, kv_dtype=L.KV_FP8, k_scale=1.0, v_scale=0.75,
                                max_seq_len=1024)
    assert (out.float().cpu() - 0.75).abs().max().item() <= 2 ** -8


def test_embed(ops):
    g = torch.Generator().manual_seed(2)
    tab = _rand(g, 50, 256)
    ids = torch.tensor([0, 49, 7, 7, 50, -1], dtype=torch.int32)
    out = ops.embed(ids.cuda(), tab.cuda()).cpu()
    assert torch.equal(out[:4].view(torch.int16), tab[ids[:4].long()].view(torch.int16))
    assert out[4:].abs().sum().item() == 0     # out-of-range ids -> zero rows


def test_sampler_greedy_and_penalty(ops):
    g = torch.Generator().manual_seed(3)
    B, V = 6, 3072
    logits = torch.randn(B, V, generator=g)
    logits[:, 2048:] = float("-inf")
    logits[2, 5] = logits[2, 9] = 50.0            # tie -> first index
    out = ops.sample(logits.cuda(), greedy=True).cpu()
    assert out.tolist() == [O.sample_row(logits[b], greedy=True) for b in range(B)]
    assert out[2].item() == 5
    seen = torch.zeros(B, V, dtype=torch.uint8)
    best = logits.argmax(-1)
    for b in range(B):
        seen[b, best[b]] = 1
    seen_d = seen.cuda()
    out2 = ops.sample(logits.cuda(), greedy=True, rep_penalty=50.0, seen=seen_d).cpu()
    ref2 = [O.sample_row(logits[b], greedy=True, rep_penalty=50.0, seen_ids=[int(best[b])]) for b in range(B)]
    assert out2.tolist() == ref2
    assert all(seen_d.cpu()[b, out2[b]] == 1 for b in range(B)), "sampled id must be marked seen"


@pytest.mark.parametrize("V,top_k", [(3072, 50), (2048, 50), (192, 0), (2048, 1)])
def test_sampler_topk_gumbel(ops, V, top_k):
    g = torch.Generator().manual_seed(V + top_k)
    B = 64
    logits = torch.randn(B, V, generator=g) * 3
    steps = torch.arange(B, dtype=torch.int32)
    kw = dict(temperature=0.9, top_k=top_k, seed=42)
    out = ops.sample(logits.cuda(), greedy=False, steps=steps.cuda(), step_mul=3, step_add=1, **kw).cpu()
    n_checked = 0
    for b in range(B):
        okw = dict(greedy=False, step=int(steps[b]) * 3 + 1, **kw)
        if O.sample_row_margin(logits[b], **okw) < 1e-4:
            continue                                     # near-tie: logf ulps may decide
        assert out[b].item() == O.sample_row(logits[b], **okw), b
        n_checked += 1
    assert n_checked >= B - 4
    if top_k:
        kth = torch.topk(logits, top_k, dim=-1).values[:, -1]
        assert (logits[torch.arange(B), out.long()] >= kth).all(), "sample outside the top-k set"


@pytest.mark.parametrize("V,top_k,top_p,scale", [(2048, 50, 0.8, 3.0), (2048, 50, 0.8, 0.3), (3072, 20, 0.5, 2.0), (192, 8, 0.95, 1.0),
                                                 (2048, 50, 0.05, 3.0)])
def test_sampler_top_p_after_top_k(ops, V, top_k, top_p, scale):
    """Nucleus cut of the Omni code predictor (qwen3_omni_moe_code_predictor_mtp.py:463-469): top-k, then keep the
    candidates whose preceding cumulative softmax mass is < top_p, then sample among them."""
    g = torch.Generator().manual_seed(V + top_k + int(100 * top_p))
    B = 64
    logits = (torch.randn(B, V, generator=g) * scale)
    logits[3, :40] = 1.5                                  # a block of exact ties inside the candidate set
    steps = torch.arange(B, dtype=torch.int32)
    kw = dict(temperature=1.0, top_k=top_k, top_p=top_p, seed=7)
    out = ops.sample(logits.cuda(), greedy=False, steps=steps.cuda(), step_mul=1, step_add=0, **kw).cpu()
    n_checked = 0
    for b in range(B):
        okw = dict(greedy=False, step=int(steps[b]), **kw)
        kept = torch.isfinite(O.top_p_filter(logits[b].masked_fill(logits[b] < torch.topk(logits[b], top_k).values[-1], float("-inf")), top_p))
        assert kept[out[b].item()], f"row {b}: sampled id outside the nucleus"
        if O.sample_row_margin(logits[b], **okw) < 1e-4:
            continue
        assert out[b].item() == O.sample_row(logits[b], **okw), b
        n_checked += 1
    assert n_checked >= B - 4
    # top_p without a usable top-k is refused, not silently ignored
    from ht_vllm_omni_amd import _lib as L
    with pytest.raises(L.OmniError):
        ops.sample(logits.cuda(), greedy=False, temperature=1.0, top_k=0, top_p=0.8)



@pytest.mark.parametrize("V,top_k,top_p", [(2048, 50, 1.0), (3072, 256, 1.0), (3072, 257, 1.0), (8192, 50, 0.9), (8192, 300, 0.9),
                                           (4096, 64, 1.0), (300, 299, 1.0)])
def test_sampler_candidate_filter_edges(ops, V, top_k, top_p):
    """The candidate-filter path of the sampler and its fallbacks: constant rows (every element is a candidate -> radix
    fallback), rows with fewer finite values than top_k, heavy ties at the threshold, masked (-inf) tails, top_k at / above
    the 256 limit of the fast path, the 12- and 32-elements-per-thread instantiations -- always the oracle's pick."""
    g = torch.Generator().manual_seed(V * 7 + top_k)
    B = 12
    logits = torch.randn(B, V, generator=g) * 2
    logits[0] = 0.25                                              # constant row
    logits[1] = float("-inf"); logits[1, 5:5 + min(top_k, 30) // 2] = torch.randn(min(top_k, 30) // 2, generator=g)   # < top_k finite
    logits[2, : V // 2] = 1.0                                     # half the row tied: threshold inside the tie block or above it
    logits[3, V // 3:] = float("-inf")                            # masked tail (allowed-codec mask)
    logits[4] = (logits[4] * 8).round() / 8                       # coarse grid: many duplicates everywhere
    logits[5] = torch.arange(V, dtype=torch.float32) * 1e-3       # strictly increasing: the top-k are the last indices
    logits[6, 7] = float("inf")                                   # +inf wins whatever the noise
    steps = torch.arange(B, dtype=torch.int32)
    kw = dict(temperature=0.7, top_k=top_k, top_p=top_p, seed=99)
    out = ops.sample(logits.cuda(), greedy=False, steps=steps.cuda(), step_mul=1, step_add=5, **kw).cpu()
    for b in range(B):
        okw = dict(greedy=False, step=int(steps[b]) + 5, **kw)
        x = logits[b] / 0.7
        kth = torch.topk(x, top_k).values[-1]
        kept = torch.isfinite(O.top_p_filter(x.masked_fill(x < kth, float("-inf")), top_p))
        if b == 6:
            assert out[b].item() == 7
            continue
        if b == 0 and top_p < 1.0 and V > 1024:
            continue        # documented limit (omni_talker.h, omni_sample): the nucleus cut sees at most 1024 candidates
        assert kept[out[b].item()], f"row {b}: sampled id outside the kept set"
        if O.sample_row_margin(logits[b], **okw) >= 1e-4:
            assert out[b].item() == O.sample_row(logits[b], **okw), b
    # greedy on the same rows (first index among ties; the all--inf row falls back to index 0 like torch.argmax)
    outg = ops.sample(logits.cuda(), greedy=True).cpu()
    for b in range(B):
        if b != 6:
            assert outg[b].item() == int(torch.argmax(logits[b])), b


@pytest.mark.parametrize("V,top_k,temperature", [(2048, 50, 0.9), (2048, 50, 1.3), (3072, 50, 0.9), (2048, 64, 1.0 / 128), (2048, 3, 0.9),
                                                 (1000, 50, 0.9), (2048, 130, 0.9), (3072, 0, 0.9), (2048, 2048, 0.7)])
def test_sampler_one_wave_pick_matches_the_four_wave_pick(V, top_k, temperature):
    """Round 6: rows without top-p are picked by ONE wave (sampler_body.cuh smp_pick_wave: lower bound by ballot bisection, LDS
    compaction, exact k-th key by bisection, Gumbel-max) -- the same function of the row as the 4-wave pick.  Both kernels on the same
    rows (debug library: omni_debug_sample_wave 0 / 1): identical ids on EVERY row -- bf16-exact logits (the code predictor's: temperature
    applied late, to the kept candidates only), fp32 logits (divided first), negative top-k sets, ties across the threshold, rows with
    fewer finite values than top_k, constant rows (> 128 candidates: the exact slow form), +inf, temperatures outside the late range,
    top_k > 128 (more kept values than list slots), no top-k; greedy too.  And against the oracle where its margin allows."""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd import ops as _ops
    g = torch.Generator().manual_seed(V * 31 + top_k)
    B = 24
    logits = torch.randn(B, V, generator=g) * 2
    logits[:12] = logits[:12].to(BF16).float()                     # the chain's head GEMM output: bf16-exact
    logits[1] = -logits[1].abs() - 3.0                             # every value negative (keys end in ones)
    logits[2] = 0.25                                               # constant row
    logits[3] = float("-inf"); logits[3, 5:25] = torch.randn(20, generator=g).to(BF16).float()     # fewer finite values than top_k = 50
    logits[4, : V // 2] = 1.0                                      # half the row tied
    logits[5] = (logits[5] * 4).round() / 4                        # coarse grid: duplicates everywhere, ties at the threshold
    logits[6, 7] = float("inf")
    logits[7, V // 3:] = float("-inf")                             # masked tail
    logits[13] = (logits[13] * 4).round() / 4
    logits[14] = float("-inf")                                     # nothing finite: index 0
    logits[15, ::2] = float("-inf")
    steps = torch.arange(B, dtype=torch.int32).cuda()
    seen = torch.zeros(B, V, dtype=torch.uint8)
    seen[16:, : V // 4] = 1                                        # repetition penalty on some rows of the fp32 half
    kw = dict(temperature=temperature, top_k=top_k, seed=1234)
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_sample_wave.argtypes = [C.c_int]; lib.omni_debug_sample_wave.restype = None
        for mode in (0, 1):
            lib.omni_debug_sample_wave(mode)
            sn = seen.clone().cuda()
            a = _ops.sample(logits.cuda(), greedy=False, steps=steps, step_mul=16, step_add=3, rep_penalty=1.3, seen=sn, **kw).cpu()
            gr = _ops.sample(logits.cuda(), greedy=True).cpu()
            res[mode] = (a, gr, sn.cpu())
        lib.omni_debug_sample_wave(1)
    assert torch.equal(res[0][0], res[1][0]), f"sampled ids differ between the 4-wave and the one-wave pick in rows {(res[0][0] != res[1][0]).nonzero().flatten().tolist()}"
    assert torch.equal(res[0][1], res[1][1]), "greedy ids differ between the two picks"
    assert torch.equal(res[0][2], res[1][2]), "seen marks differ"
    assert res[1][0][6].item() == 7 and res[1][0][14].item() == 0
    n_checked = 0
    for b in range(B):
        sl = seen[b].nonzero().flatten().tolist()
        okw = dict(greedy=False, step=b * 16 + 3, rep_penalty=1.3, seen_ids=sl, **kw)
        if b in (6, 14) or b >= 16 or O.sample_row_margin(logits[b], **okw) < 1e-4:       # (the margin helper knows no penalty: rows 16+ are
            continue                                                                        #  checked between the two kernels only)
        assert res[1][0][b].item() == O.sample_row(logits[b], **okw), b
        n_checked += 1
    assert n_checked >= 10


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_snake_beta(ops, golden_dir, dtype):
    """SnakeBeta of the Code2Wav decoder (the reference's Triton kernel -> HIP): golden vectors minted from the
    reference module, plus a decoder-sized seeded case against the oracle; odd T exercises the unaligned tail."""
    import numpy as np
    z = np.load(os.path.join(golden_dir, "snake_beta.npz"))
    cases = [tuple(torch.from_numpy(z[f"{k}{i}"]) for k in ("x", "alpha", "beta", "y")) for i in range(int(z["n"]))]
    g = torch.Generator().manual_seed(31)
    xa = torch.randn(2, 192, 4801, generator=g) * 3
    aa, ba = torch.randn(192, generator=g) * 0.3, torch.randn(192, generator=g) * 0.3
    cases.append((xa, aa, ba, O.snake_beta(xa, aa, ba)))
    for x, a, b, y in cases:
        ea, ib = torch.exp(a).cuda(), (1.0 / (torch.exp(b) + 1e-9)).cuda()
        if dtype == "fp32":
            got = ops.snake_beta(x.cuda(), ea, ib).cpu()
            torch.testing.assert_close(got, y, rtol=2e-6, atol=2e-6)
        else:
            xb = x.to(BF16)
            got = ops.snake_beta(xb.cuda(), ea, ib)
            assert_bf16_close(got, O.snake_beta(xb, a, b), ulps=1, max_mismatch=0.01, what="snake beta bf16")
    from ht_vllm_omni_amd import _lib as L
    with pytest.raises(L.OmniError):
        ops.snake_beta(torch.zeros(4, 4, device="cuda"), torch.ones(4, device="cuda"), torch.ones(4, device="cuda"))


def _moe_dev(w):
    from ht_vllm_omni_amd.engine import frag_shuffle
    d = {k: v.cuda() for k, v in w.items() if k not in ("gate_up", "down")}
    d["gate_up_f"], d["down_f"] = frag_shuffle(w["gate_up"]).cuda(), frag_shuffle(w["down"]).cuda()
    return d


@pytest.mark.parametrize("H,E,K,I,Is,T,norm", [(64, 16, 4, 32, 64, 9, False), (128, 128, 8, 96, 64, 37, False), (64, 16, 2, 32, 32, 5, True),
                                               (1024, 128, 8, 384, 768, 64, False), (1024, 128, 8, 384, 768, 1, False)])
def test_moe_block(ops, H, E, K, I, Is, T, norm):
    """Sparse-MoE MLP of the Omni talker (row a11): routing (indices exact up to exact ties, weights within one bf16 ulp),
    then the expert path + bf16 accumulation in expert order + gated shared expert against the oracle (which is pinned to
    HF's module by tests/test_oracle_golden.py); the last two shapes are the real talker's (H 1024, 128 experts, top-8)."""
    from tests.util import make_moe_weights
    w = make_moe_weights(H, E, I, Is, seed=7 * E + K)
    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, H, generator=g).to(BF16)
    out, idx, wts = ops.moe_block(x.cuda(), _moe_dev(w), K, norm)
    logits, rw, ri = O.moe_route(x, w["router"], K, norm)
    # routing: the device picks the same expert set unless two probabilities tie within fp32 rounding
    same_rows = (idx.cpu().sort(-1).values == ri.to(torch.int32).sort(-1).values).all(-1)
    probs = torch.softmax(logits.float(), -1).sort(-1, descending=True).values
    tie = (probs[:, K - 1] - probs[:, K]).abs() <= 1e-6 * probs[:, K - 1]      # bf16 logits: the k-th and (k+1)-th can be EQUAL
    assert bool((same_rows | tie).all()), "routing sets differ without a tie at the cut"
    assert (idx.cpu()[same_rows] == ri.to(torch.int32)[same_rows]).float().mean().item() >= 0.9, "routing order (ties inside the set may swap)"
    assert_bf16_close(wts.cpu()[same_rows], rw[same_rows], ulps=1, max_mismatch=0.05, what="routing weights")
    ref = O.moe_block(x, w, K, norm)
    ok = same_rows.nonzero().flatten()
    # eight bf16 contributions per element may cancel, and a 1-ulp routing weight moves one of them: bound at tensor scale
    from tests.util import assert_e2e_close
    assert_e2e_close(out.cpu()[ok], ref[ok], mean_tol=2e-3 * max(1.0, ref.float().abs().max().item()), max_ulps=2, what="moe block output")
    # exactness of the expert path given the device's own routing: oracle block with the routing forced
    ref2 = _moe_oracle_with_routing(x, w, idx.cpu().long(), wts.cpu())
    scale = max(1.0, ref2.float().abs().max().item())
    assert_e2e_close(out.cpu(), ref2, mean_tol=5e-4 * scale, max_ulps=2, what="moe experts given routing")
    assert (out.cpu().view(torch.int16) != ref2.view(torch.int16)).float().mean().item() < 0.1, "bit-identical for > 90 % of the outputs"


def _moe_oracle_with_routing(x, w, idx, wts):
    I = w["gate_up"].shape[1] // 2
    out = torch.zeros_like(x)
    for e in range(w["gate_up"].shape[0]):
        kpos, tok = torch.where(idx.t() == e)
        if tok.numel() == 0:
            continue
        gu = O.linear(x[tok], w["gate_up"][e])
        y = O.linear(O.silu_mul(gu[:, :I], gu[:, I:]), w["down"][e]) * wts[tok, kpos, None]
        out.index_add_(0, tok, y)
    Is = w["shared_gate_up"].shape[0] // 2
    sgu = O.linear(x, w["shared_gate_up"])
    shared = torch.sigmoid(O.linear(x, w["shared_gate"])) * O.linear(O.silu_mul(sgu[:, :Is], sgu[:, Is:]), w["shared_down"])
    return out + shared


@pytest.mark.parametrize("H,E,K,I,Is,T,ep", [(64, 16, 4, 32, 64, 9, 2), (1024, 128, 8, 384, 768, 64, 8)])
def test_moe_experts_expert_parallel_and_fp8_weights(ops, H, E, K, I, Is, T, ep):
    """omni_moe_experts_ex (BASELINE configs #4 / #5): (a) expert parallel -- every rank runs the block on its E / ep experts,
    slots of foreign experts count as zero, and the ranks' partial outputs add up to the single-rank block; (b) fp8 e4m3fn
    expert weights with per-row scales, dequantised in registers: bit-comparable to the bf16 kernel (and within the oracle
    bound) on the weight matrix bf16(fp8 * scale)."""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd.engine import fp8_dequant_rows, fp8_quant_rows, frag_shuffle
    from tests.util import assert_e2e_close, make_moe_weights
    lib = L.load()
    w = make_moe_weights(H, E, I, Is, seed=3 * E + ep)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, H, generator=g).to(BF16).cuda()
    logits = ops.gemm(x, w["router"].cuda())
    idx, wts = ops.moe_route(logits, K, False)

    def run(gu, dn, sgu, sdn, e0, El, with_shared):
        act = torch.empty(T * K, I, dtype=BF16, device="cuda")
        y = torch.zeros(T * K, H, dtype=BF16, device="cuda")
        out = torch.empty(T, H, dtype=BF16, device="cuda")
        shared = None
        if with_shared:
            shared = ops.gemm(ops.gemm(x, w["shared_gate_up"].cuda(), epilogue=L.EPI_SILU_MUL), w["shared_down"].cuda())
        L.check(lib.omni_moe_experts_ex(L.ptr(x), L.ptr(idx), L.ptr(wts), L.ptr(gu), L.ptr(sgu), L.ptr(dn), L.ptr(sdn), L.ptr(shared),
                                        L.ptr(w["shared_gate"].cuda()) if with_shared else None, L.ptr(act), L.ptr(y), L.ptr(out), T, H, I,
                                        El, e0, K, L.current_stream()), "omni_moe_experts_ex")
        return out

    full = run(frag_shuffle(w["gate_up"]).cuda(), frag_shuffle(w["down"]).cuda(), None, None, 0, E, True)
    El = E // ep
    parts = [run(frag_shuffle(w["gate_up"][r * El:(r + 1) * El]).cuda(), frag_shuffle(w["down"][r * El:(r + 1) * El]).cuda(), None, None,
                 r * El, El, r == 0) for r in range(ep)]             # the shared expert rides on rank 0 in this test
    tot = sum(p.float() for p in parts).to(BF16)
    scale = max(1.0, float(full.float().abs().max()))
    assert_e2e_close(tot.cpu(), full.cpu(), mean_tol=2e-3 * scale, max_ulps=3, what="sum of expert-parallel partials")
    # fp8 weights: same kernel arithmetic on bf16(fp8 * scale)
    q_gu, s_gu = fp8_quant_rows(w["gate_up"])
    q_dn, s_dn = fp8_quant_rows(w["down"])
    deq_gu, deq_dn = fp8_dequant_rows(q_gu, s_gu), fp8_dequant_rows(q_dn, s_dn)
    got8 = run(frag_shuffle(q_gu).cuda(), frag_shuffle(q_dn).cuda(), s_gu.cuda(), s_dn.cuda(), 0, E, True)
    ref8 = run(frag_shuffle(deq_gu).cuda(), frag_shuffle(deq_dn).cuda(), None, None, 0, E, True)
    assert torch.equal(got8, ref8), "in-register dequantisation must reproduce the dequantised bf16 matrix bit for bit"
    w8 = dict(w, gate_up=deq_gu, down=deq_dn)
    ref_o = _moe_oracle_with_routing(x.cpu(), w8, idx.cpu().long(), wts.cpu())
    assert_e2e_close(got8.cpu(), ref_o, mean_tol=5e-4 * max(1.0, float(ref_o.float().abs().max())), max_ulps=2, what="fp8-weight experts vs oracle")
    assert float((got8.float() - full.float()).abs().mean()) > 0          # the quantisation is really in effect


def test_torch_library_ops_run_the_same_kernels_as_the_ctypes_wrappers():
    """torch.ops.mi355x_omni.* (torch_ops.py) against ops.* on the same inputs: bit-identical (same C entry points)."""
    import ht_vllm_omni_amd.torch_ops  # noqa: F401  (registers the ops)
    from ht_vllm_omni_amd import _lib as LL, ops as OP
    t = torch.ops.mi355x_omni
    g = torch.Generator().manual_seed(0)
    M, H, I = 24, 512, 768
    x = torch.randn(M, H, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(H, generator=g) * 0.1 + 1).to(torch.bfloat16).cuda()
    res = torch.randn(M, H, generator=g).to(torch.bfloat16).cuda()
    # rmsnorm_residual_: out-variant, residual updated in place
    r1, r2 = res.clone(), res.clone()
    out = torch.empty_like(x)
    t.rmsnorm_residual_(None, x, r1, w, out, 1e-6)
    ref = OP.rmsnorm(None, w, 1e-6, delta=x, residual=r2)
    assert torch.equal(out, ref) and torch.equal(r1, r2)
    # skinny_gemm / lmhead_mask / silu_mul
    W = (torch.randn(2 * I, H, generator=g) * 0.05).to(torch.bfloat16).cuda()
    y = t.skinny_gemm(x, W, None, None, LL.EPI_BF16, 0)
    assert torch.equal(y, OP.gemm(x, W))
    assert torch.equal(t.silu_mul(y), OP.silu_mul(y))
    mask = (torch.arange(2 * I) % 3 != 0).to(torch.uint8).cuda()
    lg = t.lmhead_mask(x, W, mask, True)
    assert torch.equal(lg, OP.gemm(x, W, epilogue=LL.EPI_F32_BF16RND, mask=mask)) and torch.isinf(lg[:, 0]).all()
    # topk_sample
    steps = torch.full((M,), 3, dtype=torch.int32).cuda()
    a = t.topk_sample(lg, None, steps.clone(), False, 0.9, 50, 1.0, 1.0, 42, 1, 0, False)
    b = OP.sample(lg, greedy=False, temperature=0.9, top_k=50, seed=42, steps=steps.clone())
    assert torch.equal(a, b)
    # qknorm_rope_kvwrite_ + paged_attn_decode over a small fp8 cache
    hq, hkv, D, bs, nb = 4, 2, 128, 16, 8
    T_ = 20
    qkv = torch.randn(T_, (hq + 2 * hkv) * D, generator=g).to(torch.bfloat16).cuda()
    qn = torch.ones(D, dtype=torch.bfloat16).cuda(); kn = torch.ones(D, dtype=torch.bfloat16).cuda()
    pos = torch.arange(T_, dtype=torch.int32).cuda()
    cs = OP.rope_table(64, D, 1e6).cuda()
    slots = (torch.arange(T_) + bs).to(torch.int64).cuda()
    caches = [[torch.zeros(nb, bs, hkv, D, dtype=torch.uint8).cuda() for _ in range(2)] for _ in range(2)]
    q1 = torch.empty(T_, hq * D, dtype=torch.bfloat16).cuda()
    t.qknorm_rope_kvwrite_(qkv, qn, kn, pos, cs, slots, q1, caches[0][0], caches[0][1], None, None, hq, hkv, D, 1e-6, LL.KV_FP8, 1.0, 1.0)
    q2 = OP.qknorm_rope_kvwrite(qkv, qn, kn, pos, cs, slots, caches[1][0], caches[1][1], q_heads=hq, kv_heads=hkv, head_dim=D, eps=1e-6,
                                kv_dtype=LL.KV_FP8)
    assert torch.equal(q1, q2) and torch.equal(caches[0][0], caches[1][0]) and torch.equal(caches[0][1], caches[1][1])
    bt = torch.tensor([[1, 2, 0, 0]], dtype=torch.int32).cuda()
    sl = torch.tensor([T_], dtype=torch.int32).cuda()
    qd = q1[-1:].contiguous()
    o1 = t.paged_attn_decode(qd, caches[0][0], caches[0][1], bt, sl, None, None, hq, hkv, D, bs, LL.KV_FP8, 1.0, 1.0, 64)
    o2 = OP.paged_attn_decode(qd, caches[0][0], caches[0][1], bt, sl, q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs, kv_dtype=LL.KV_FP8,
                              max_seq_len=64)
    assert torch.equal(o1, o2)
