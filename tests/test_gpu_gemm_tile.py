"""omni_gemm_tile (csrc/gemm_prefill.hip): the large-M MFMA GEMM / conv-as-GEMM kernel against fp64 references built from
the SAME bf16 operands -- plain GEMMs at every tile configuration (N = 256 k, 192 k, 128 k, 96 k, ragged), odd k-step
counts, M tails; causal dilated conv1d windows and the transposed-conv mapping against torch.nn.functional on fp64; every
epilogue (bias, GELU, scale, residual, snake second output, interleaved SiLU-mul).  Bound: one bf16 rounding of an fp32
accumulation -- |got - ref| <= 2^-8 |ref| + tiny (half an ulp) + the accumulation-order slack 2^-16 * sum|a_k b_k|."""
import pytest
import torch

from tests.util import BF16

pytestmark = pytest.mark.gpu


def _ops():
    from ht_vllm_omni_amd import ops
    from ht_vllm_omni_amd import _lib as L
    from ht_vllm_omni_amd.engine import frag_shuffle, gu8_shuffle
    return ops, L, frag_shuffle, gu8_shuffle


def _close(got, ref, mag, what):
    """got bf16, ref fp64 exact value of the pre-rounding quantity, mag = sum of |terms| (accumulation-order slack)."""
    g = got.double().cpu()
    tol = ref.abs() * 2.0 ** -8 + mag * 2.0 ** -16 + 1e-30
    bad = (g - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} off; worst {((g - ref).abs() / tol).max().item():.2f} x bound"


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 512, 2048), (257, 192, 96), (513, 384, 672), (640, 128, 64),
                                   (1111, 96, 672), (255, 288, 96), (70, 48, 32), (900, 272, 160), (6400, 2048, 2048),
                                   (700, 256, 64), (1300, 768, 96), (515, 256, 32), (2000, 1024, 6144)])
def test_plain_gemm_every_tile_configuration(M, N, K):
    """tile_hint 1 = the large tiles (256 / 192 / 128 / 96 columns by N; the lockstep kernel), 2 = the 128 x 64 small-M tile, 0 = the
    dispatch rule (plain GEMMs with N % 256 == 0: the two-group kernel)."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16)
    ref = x.double() @ w.double().T
    mag = x.double().abs() @ w.double().abs().T
    outs = []
    for hint in (1, 2, 0):
        outs.append(ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), tile_hint=hint))
        _close(outs[-1], ref, mag, f"gemm {M}x{N}x{K} hint {hint}")
    assert torch.equal(outs[0], outs[1])      # same k order per output element in both geometries: bit-identical
    assert torch.equal(outs[0], outs[2])
    if N % 256 == 0:                           # the 224- and 192-row tiles of the 256-column geometry; 5 / 6 / 7 = the two-group tile at 256 / 224 / 192 rows
        for hint in (3, 4, 5, 6, 7):
            assert torch.equal(outs[0], ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), tile_hint=hint)), hint


@pytest.mark.parametrize("Cin,Cout,taps,dil,T", [(96, 96, 7, 1, 700), (96, 96, 7, 9, 531), (192, 192, 7, 3, 400), (512, 1024, 3, 1, 77),
                                                  (768, 768, 7, 9, 290), (96, 96, 1, 1, 333)])
def test_causal_dilated_conv1d_as_windowed_gemm(Cin, Cout, taps, dil, T):
    """y[t, co] = b[co] + sum_{j, ci} w[co, ci, j] x[t - (taps - 1 - j) dil, ci]  -- Qwen3TTSTokenizerV2CausalConvNet
    (modeling_qwen3_tts_tokenizer_v2.py:174-207: left padding (k - 1) d, stride 1) on time-major activations."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(Cin + taps * 13 + dil)
    x = torch.randn(T, Cin, generator=g).to(BF16)
    w = (torch.randn(Cout, Cin, taps, generator=g) * 0.05).to(BF16)
    b = torch.randn(Cout, generator=g)
    wk = w.permute(0, 2, 1).reshape(Cout, taps * Cin).contiguous()            # K index = tap * Cin + ci
    out = ops.gemm_tile(x.cuda(), frag_shuffle(wk).cuda(), bias=b.cuda(), taps=taps, dilation=dil, tile_hint=1)
    assert torch.equal(out, ops.gemm_tile(x.cuda(), frag_shuffle(wk).cuda(), bias=b.cuda(), taps=taps, dilation=dil, tile_hint=2))
    xp = torch.nn.functional.pad(x.double().T[None], ((taps - 1) * dil, 0))
    ref = torch.nn.functional.conv1d(xp, w.double(), b.double(), dilation=dil)[0].T
    mag = torch.nn.functional.conv1d(xp.abs(), w.double().abs(), b.double().abs(), dilation=dil)[0].T
    _close(out, ref, mag, f"conv k{taps} d{dil} {Cin}->{Cout}")


@pytest.mark.parametrize("Cin,Cout,s,T", [(192, 96, 3, 300), (384, 192, 4, 161), (1536, 768, 8, 40), (768, 384, 5, 97)])
def test_transposed_conv_stride_s_kernel_2s_as_one_gemm(Cin, Cout, s, T):
    """Qwen3TTSTokenizerV2CausalTransConvNet (…:210-224): ConvTranspose1d(k = 2 s, stride s), last s samples dropped.
    y[s i + r] = x[i] w[:, :, r] + x[i - 1] w[:, :, r + s]  ->  A[i] = [x[i - 1] | x[i]], W rows n = r * Cout + co: the row-major
    [T, s * Cout] output is the [T * s, Cout] signal."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(Cin + s)
    x = torch.randn(T, Cin, generator=g).to(BF16)
    w = (torch.randn(Cin, Cout, 2 * s, generator=g) * 0.05).to(BF16)         # ConvTranspose1d weight [Cin, Cout, k]
    b = torch.randn(Cout, generator=g)
    wk = torch.cat([w[:, :, s:], w[:, :, :s]], dim=0)                         # [2 Cin (x[i-1] | x[i]), Cout, s (phase r)]
    wk = wk.permute(2, 1, 0).reshape(s * Cout, 2 * Cin).contiguous()
    out = ops.gemm_tile(x.cuda(), frag_shuffle(wk).cuda(), bias=b.repeat(s).cuda(), taps=2, dilation=1, tile_hint=1)
    assert torch.equal(out, ops.gemm_tile(x.cuda(), frag_shuffle(wk).cuda(), bias=b.repeat(s).cuda(), taps=2, dilation=1, tile_hint=2))
    ref = torch.nn.functional.conv_transpose1d(x.double().T[None], w.double(), b.double(), stride=s)[0, :, : T * s].T
    mag = torch.nn.functional.conv_transpose1d(x.double().abs().T[None], w.double().abs(), b.double().abs(), stride=s)[0, :, : T * s].T
    _close(out.view(T * s, Cout), ref, mag, f"transconv s{s} {Cin}->{Cout}")


def test_epilogues_gelu_scale_fp32_residual_stream_and_snake_output():
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 700, 192, 192
    x = torch.randn(M, K, generator=g).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.08).to(BF16)
    b, sc = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    r = torch.randn(M, N, generator=g) * 4
    al, ib = torch.rand(N, generator=g) + 0.5, torch.rand(N, generator=g) + 0.5
    acc = x.double() @ w.double().T + b.double()
    mag = x.double().abs() @ w.double().abs().T + b.double().abs()
    # GELU * scale -> bf16
    out = ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), bias=b.cuda(), scale=sc.cuda(), act=L.TILE_ACT_GELU)
    _close(out, torch.nn.functional.gelu(acc) * sc.double(), mag * sc.double() * 1.2 + 1e-3, "gelu * scale")
    # y = acc * scale + resid: fp32 stream (in place on the residual buffer), bf16 copy, snake of the fp32 value
    stream = r.clone().cuda()
    f, o, s2 = ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), bias=b.cuda(), scale=sc.cuda(), resid=stream, out_f32=stream,
                             snake=(al.cuda(), ib.cuda()), want="fbs")
    assert f.data_ptr() == stream.data_ptr()
    ref = acc * sc.double() + r.double()
    tol = mag * sc.double() * 2.0 ** -16 + ref.abs() * 2.0 ** -22 + 1e-6
    assert ((f.double().cpu() - ref).abs() <= tol).all()
    assert torch.equal(o.cpu(), f.cpu().to(BF16))                                  # one rounding of the same fp32 value
    z = f.double().cpu()
    ref2 = z + ib.double() * torch.sin(z * al.double()) ** 2
    assert ((s2.double().cpu() - ref2).abs() <= ref2.abs() * 2.0 ** -8 + 1e-4 + 4e-6 * z.abs()).all()   # __sinf argument error
    for hint in (1, 2):
        only = ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), bias=b.cuda(), scale=sc.cuda(), resid=r.cuda(), snake=(al.cuda(), ib.cuda()),
                             want="s", tile_hint=hint)
        assert torch.equal(only, s2), hint
        f2 = ops.gemm_tile(x.cuda(), frag_shuffle(w).cuda(), bias=b.cuda(), scale=sc.cuda(), resid=r.cuda(), want="f", tile_hint=hint)
        assert torch.equal(f2, f), hint


def test_two_group_tile_with_the_prefills_residual_epilogue():
    """tile_hints 5 / 6 / 7 (the two-group tile at 256 / 224 / 192 rows) on the o_proj / down_proj form (fp32 residual in, bf16 out): bit-identical to the 256 x 256 tile of hint 1."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(11)
    M, N, K = 1234, 512, 1088
    x = torch.randn(M, K, generator=g).to(BF16).cuda()
    w = frag_shuffle((torch.randn(N, K, generator=g) * 0.05).to(BF16)).cuda()
    r = (torch.randn(M, N, generator=g) * 3).cuda()
    b = torch.randn(N, generator=g).cuda()
    for want in ("b", "f"):
        a1 = ops.gemm_tile(x, w, bias=b, resid=r, want=want, tile_hint=1)
        for hint in (5, 6, 7):
            assert torch.equal(a1, ops.gemm_tile(x, w, bias=b, resid=r, want=want, tile_hint=hint)), (want, hint)


@pytest.mark.parametrize("M,I,K", [(333, 3072, 1024), (6400, 6144, 2048), (100, 64, 64)])
def test_interleaved_gate_up_with_fused_silu_mul(M, I, K):
    """The decode step's gate_up weight as it lies in HBM (gate / up rows interleaved by 8, fragment-major) serves the
    prefill GEMM: act = bf16(bf16(SiLU(bf16 gate)) * bf16 up) -- ops.silu_mul's (= torch bf16) rounding points."""
    ops, L, frag_shuffle, gu8_shuffle = _ops()
    g = torch.Generator().manual_seed(I)
    x = torch.randn(M, K, generator=g).to(BF16)
    w = (torch.randn(2 * I, K, generator=g) * 0.03).to(BF16)
    out = ops.gemm_tile(x.cuda(), gu8_shuffle(w).cuda(), act=L.TILE_ACT_SILU_MUL_GU8, tile_hint=1)
    assert out.shape == (M, I)
    assert torch.equal(out, ops.gemm_tile(x.cuda(), gu8_shuffle(w).cuda(), act=L.TILE_ACT_SILU_MUL_GU8, tile_hint=2))
    if (2 * I) % 256 == 0:
        for hint in (3, 4, 5, 6, 7):
            assert torch.equal(out, ops.gemm_tile(x.cuda(), gu8_shuffle(w).cuda(), act=L.TILE_ACT_SILU_MUL_GU8, tile_hint=hint)), hint
    gu = (x.double() @ w.double().T)
    ga, up = gu[:, :I].float().to(BF16), gu[:, I:].float().to(BF16)
    ref = torch.nn.functional.silu(ga) * up                       # torch bf16 semantics: silu rounds, the product rounds
    same = (out.cpu() == ref).float().mean().item()
    assert same >= 0.995, same                                    # the rest: 1-ulp flips of gate / up at fp32 accumulation-order ties
    assert ((out.cpu().float() - ref.float()).abs() <= ref.float().abs() * 2.0 ** -6 + 1e-3).all()
    from ht_vllm_omni_amd.engine import frag_shuffle as _fs
    plain = ops.gemm_tile(x.cuda(), _fs(w).cuda())                # the same GEMM without the epilogue, through omni_silu_mul
    assert torch.equal(out, ops.silu_mul(plain))


@pytest.mark.parametrize("M,N,K,taps,dil", [(6400, 2048, 2048, 1, 1), (3000, 96, 96, 7, 9), (5000, 192, 192, 7, 3), (700, 1024, 1024, 1, 1),
                                              (2600, 768, 768, 7, 1), (6438, 2048, 6144, 1, 1), (6438, 4096, 2048, 1, 1)])
def test_race_screen_repeated_launches_are_bit_identical(M, N, K, taps, dil):
    """The ring's synchronisation (LDS-DMA retired by counted vmcnt + a barrier one phase before the read; buffers restaged only
    after the barrier that follows their last read) screened the way a hand-scheduled pipeline has to be: 150 back-to-back
    launches per shape and geometry, other kernels' traffic in between, every result bit-identical to the first and to the other
    geometry (a read that overtakes its DMA shows up as a rare wrong tile, not as a failed reference check)."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(N + taps)
    x = torch.randn(M, K, generator=g).to(BF16).cuda()
    w = frag_shuffle((torch.randn(N, taps * K, generator=g) * 0.05).to(BF16)).cuda()
    noise = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    first = None
    for hint in (1, 2, 5, 6, 7) if (taps == 1 and N % 256 == 0) else (1, 2):  # 5 / 6 / 7: the two-group tile, a synchronisation structure of its own
        outs = []
        for it in range(150):
            outs.append(ops.gemm_tile(x, w, taps=taps, dilation=dil, tile_hint=hint))
            if it % 7 == 0:
                noise.add_(1)                                  # evict / perturb between launches
        ref = outs[0] if first is None else first
        first = ref
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
        assert not bad, (hint, bad[:5])


def test_rejects_bad_shapes():
    ops, L, frag_shuffle, _ = _ops()
    from ht_vllm_omni_amd._lib import OmniError
    x = torch.zeros(64, 48, dtype=BF16, device="cuda")
    with pytest.raises(OmniError):
        ops.gemm_tile(x, torch.zeros(32, 48, dtype=BF16, device="cuda"))            # K % 32
    with pytest.raises(OmniError):
        ops.gemm_tile(torch.zeros(64, 64, dtype=BF16, device="cuda"), torch.zeros(24, 64, dtype=BF16, device="cuda"))   # N % 16


@pytest.mark.parametrize("E,cap,N,K", [(16, 128, 128, 256), (128, 448, 768, 1024), (128, 448, 1024, 384), (5, 320, 192, 96)])
def test_grouped_launch_over_an_expert_sorted_batch(E, cap, N, K):
    """groups > 1: group g multiplies its rows [g cap, g cap + rows[g]) by ITS matrix; rows past the live count are neither read
    nor written (the output buffer keeps its sentinel there), empty groups cost nothing -- the MoE prefill's [E, cap, H] batch
    (engine._moe_mlp_tile).  The same rows through E separate plain launches give the same bits."""
    ops, L, frag_shuffle, _ = _ops()
    g = torch.Generator().manual_seed(E * 31 + N)
    x = torch.randn(E * cap, K, generator=g).to(BF16)
    w = (torch.randn(E, N, K, generator=g) * 0.05).to(BF16)
    rows = torch.randint(0, cap + 1, (E,), generator=g).to(torch.int32)
    rows[0], rows[-1] = cap, 0                                               # a full group and an empty one
    if E > 2:
        rows[1] = 1
    wf = frag_shuffle(w).cuda()
    out = torch.full((E * cap, N), 7.0, dtype=BF16, device="cuda")
    # poison the dead rows of x: they must read as zero, i.e. not leak into live rows (they cannot: rows are independent) nor fault
    xp = x.clone()
    for e in range(E):
        xp[e * cap + int(rows[e]): (e + 1) * cap] = float("nan")
    ops.gemm_tile(xp.cuda(), wf, out=out, groups=E, group_rows=rows.cuda())
    out = out.cpu()
    for e in range(E):
        r = int(rows[e])
        live, dead = out[e * cap: e * cap + r], out[e * cap + r: (e + 1) * cap]
        assert torch.equal(dead, torch.full_like(dead, 7.0)), f"group {e}: rows past the live count were written"
        if r:
            xe = x[e * cap: e * cap + r]
            _close(live, xe.double() @ w[e].double().T, xe.double().abs() @ w[e].double().abs().T, f"group {e}")
            if e < 3:
                assert torch.equal(live, ops.gemm_tile(xe.cuda(), wf[e].contiguous()).cpu())
    # without group_rows: every group computes its cap rows
    full = ops.gemm_tile(x.cuda(), wf, groups=E).cpu()
    assert torch.equal(full[:cap], ops.gemm_tile(x[:cap].cuda(), wf[0].contiguous()).cpu())
    assert torch.equal(full[-cap:], ops.gemm_tile(x[-cap:].cuda(), wf[-1].contiguous()).cpu())
