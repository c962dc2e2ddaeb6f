"""End-to-end parity of the native talker step (through the C-ABI engine) against the CPU oracle
and the golden vectors minted from the reference.  Bar (BASELINE.json north_star): KV block/slot
indices bit-exact, sampled ids / codes bit-exact, logits within 1e-3 (pre-rounding fp32) resp. one
bf16 ulp (the reference's bf16 logits)."""
import os

import numpy as np
import pytest
import torch

from ht_vllm_omni_amd import _lib as L
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.sched import BlockPool
from ht_vllm_omni_amd.weights import make_weights
from oracle import talker_oracle as O
from tests.util import codes_on_the_oracles_frame, assert_f32_close, BF16, assert_e2e_close, bf16_from_u16

pytestmark = pytest.mark.gpu


def _engine(d, w, **kw):
    from ht_vllm_omni_amd.engine import TalkerEngine
    return TalkerEngine(d, w, **kw)


def test_code_predictor_matches_reference_golden(golden_dir):
    """Reference's own code predictor (qwen3_tts_code_predictor_vllm.py) greedy codes: bit-exact."""
    z = np.load(os.path.join(golden_dir, "code_predictor_tiny.npz"))
    d = get_dims("tiny")
    w = make_weights(d, seed=int(z["seed"]), std=float(z["std"]), norm_noise=float(z["norm_noise"]))
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=8)
    code0 = torch.from_numpy(z["layer0_code"]).reshape(-1).to(torch.int32)
    B = code0.shape[0]
    e0 = bf16_from_u16(z["layer0_embed"]).reshape(B, -1)
    lh = bf16_from_u16(z["last_talker_hidden"]).reshape(B, -1)
    codes = eng.code_predictor(code0.cuda(), e0.cuda(), lh.cuda(), greedy=True)
    assert torch.equal(codes.cpu(), torch.from_numpy(z["all_codes"]))


@pytest.mark.parametrize("B", [1, 5, 16])
def test_code_predictor_matches_oracle(B):
    d = get_dims("tiny")
    w = make_weights(d, seed=3, std=0.08, norm_noise=0.1)
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16)
    orc = O.TalkerOracle(d, w)
    g = torch.Generator().manual_seed(B)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    e0 = w["embed"][code0]
    lh = torch.randn(B, d.hidden, generator=g).to(BF16)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    on = codes_on_the_oracles_frame(codes, ref_codes, ref_lg, what="code predictor")
    assert int(on.sum()) >= (B + 1) // 2
    assert_e2e_close(lg.cpu()[on], ref_lg[on], mean_tol=1e-3, what="code predictor logits")
    # sampled mode: same hash RNG on both sides (reference uses the global torch generator)
    steps = torch.full((B,), 7, dtype=torch.int32)
    codes_s = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=False, temperature=0.9,
                                 top_k=50, seed=42, steps=steps.cuda())
    ref_s = orc.code_predictor(code0, e0, lh, do_sample=True, temperature=0.9, top_k=50, seed=42, step=7)
    agree = (codes_s.cpu() == ref_s).float().mean().item()
    assert agree >= 0.9, f"sampled codes agree only {agree:.2%}"   # a near-tie flips the rest of that row


def test_code_predictor_16_groups_dense_cache_path():
    """Q = 16 code groups (the real talker): the predictor's private KV cache has 17-token blocks and the attention
    kernel runs its index-free dense mode (position known at launch, both history groups prefetched) -- all 15 passes
    against the oracle's re-prefill formulation, 1.7B code-predictor layer shapes, 2 layers."""
    d = get_dims("tts-1.7b").with_(layers=1, cp_layers=2, max_model_len=256)
    assert d.num_code_groups == 16
    w = make_weights(d, seed=11, std=0.02)
    B = 9
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16)
    orc = O.TalkerOracle(d, w)
    g = torch.Generator().manual_seed(B)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    e0 = w["embed"][code0]
    lh = torch.randn(B, d.hidden, generator=g).to(BF16)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    # a greedy pick may differ only where the oracle's own top-2 logits are within a bf16 ulp; everything after such a
    # flip in that row follows a different prefix and is excluded from the logits comparison
    keep = torch.ones(B, d.num_code_groups - 1, dtype=torch.bool)
    for b in range(B):
        bad = (codes[b].cpu() != ref_codes[b]).nonzero().flatten().tolist()
        if bad:
            q = bad[0]
            top = torch.topk(ref_lg[b, q - 1], 2).values
            assert (top[0] - top[1]).item() <= 2 ** -6, f"row {b} group {q}: code differs without a near-tie"
            keep[b, q:] = False
    assert keep.float().mean().item() > 0.8
    # logits here are O(1) (N(0,1) hidden rows in): ~0.15 bf16 ulp at the first group, growing with the number of cached
    # positions whose K/V each carry their own rounding; the separate-norm and row-major paths show the same figures
    assert_e2e_close(lg.cpu()[:, 0], ref_lg[:, 0], mean_tol=1.5e-3, what="16-group code predictor logits, group 1")
    assert_e2e_close(lg.cpu()[keep], ref_lg[keep], mean_tol=3e-3, max_ulps=3, what="16-group code predictor logits")


@pytest.mark.parametrize("model,B", [("tts-1.7b", 37), ("tts-1.7b", 64), ("tts-0.6b", 5)])
def test_code_predictor_pair_pass_matches_sequential_and_oracle(model, B):
    """Positions 0 and 1 of the code predictor as one two-block pass (rows [0, B) and [Bp, Bp + B), 128-row slabs, the pair
    attention kernel) against the one-position-per-pass form of the same library and against the oracle: the position-0 /
    position-1 K and V rows in the predictor's cache are bit-identical, the first group's logits are within the same bound."""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    d = get_dims(model).with_(layers=1, cp_layers=3, max_model_len=256)
    w = make_weights(d, seed=21, std=0.02)
    g = torch.Generator().manual_seed(B)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    e0 = w["embed"][code0]
    lh = torch.randn(B, d.hidden, generator=g).to(BF16)
    orc = O.TalkerOracle(d, w)
    ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    res = {}
    with L.debug_library() as lib:      # run-time schedule knob: libomni_talker_debug.so only
        lib.omni_debug_cp_pair01.argtypes = [C.c_int]; lib.omni_debug_cp_pair01.restype = None
        try:
            for on in (0, 1):
                lib.omni_debug_cp_pair01(on)
                eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=64)
                for ids in (code0.to(torch.int32).cuda(), None):          # folded e0 table (step path) / explicit embedding (parity entry)
                    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
                    res[on] = (codes.cpu(), lg.cpu())
                assert_e2e_close(res[on][1][:, 0], ref_lg[:, 0], what=f"pair={on} logits, group 1")      # 3 layers: ~2.5e-3 either way
                assert (res[on][0][:, 1] == ref_codes[:, 1]).float().mean().item() >= 0.9
        finally:
            lib.omni_debug_cp_pair01(1)
    # same library, two schedules: group-1 logits agree to rounding (the pair kernel sums the two scores in another order)
    assert_e2e_close(res[1][1][:, 0], res[0][1][:, 0], what="pair vs sequential logits, group 1")
    err = [(res[on][1][:, 0].float() - ref_lg[:, 0].float()).abs().mean().item() for on in (0, 1)]
    assert abs(err[1] - err[0]) <= 0.25 * max(err), f"the two schedules are not equally close to the oracle: {err}"
    assert (res[1][0] == res[0][0]).all(dim=1).float().mean().item() >= 0.8


def test_code_predictor_omni_style_no_projection_top_p():
    """The Omni talker's predictor (qwen3_omni_moe_code_predictor_mtp.py:405-482): no small_to_mtp projection
    (predictor width == talker width, as in the 0.6B TTS config too), growing sequence (== the KV-cached form), T = 1,
    top-k 50 then top-p 0.8.  Greedy logits against the oracle, sampled codes with the shared hash RNG."""
    d = get_dims("tts-0.6b").with_(layers=1, cp_layers=2, max_model_len=256)
    assert not d.has_cp_projection and d.num_code_groups == 16
    w = make_weights(d, seed=13, std=0.02)
    B = 12
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16)
    orc = O.TalkerOracle(d, w)
    g = torch.Generator().manual_seed(B)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    e0 = w["embed"][code0]
    lh = torch.randn(B, d.hidden, generator=g).to(BF16)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    assert_e2e_close(lg.cpu()[:, 0], ref_lg[:, 0], mean_tol=2e-3, what="no-projection predictor logits, group 1")   # 1.2-1.5e-3 over seeds, pair pass or not (scripts/diag_pair01.py)
    assert (codes.cpu()[:, 1] == ref_codes[:, 1]).float().mean().item() >= 0.9
    steps = torch.full((B,), 3, dtype=torch.int32)
    kw = dict(temperature=1.0, top_k=50, top_p=0.8, seed=11)
    got = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=False, steps=steps.cuda(), **kw).cpu()
    ref = orc.code_predictor(code0, e0, lh, do_sample=True, step=3, **kw)
    # a flipped pick changes the rest of that row: compare the first sampled group, then whole-row agreement
    assert (got[:, 1] == ref[:, 1]).float().mean().item() >= 0.9, "first sampled group"
    assert (got == ref).all(dim=1).float().mean().item() >= 0.5, "whole rows"


def _scenario(d, w, kv, prompt_lens, n_steps, *, B_pad=None, sampling=None, num_blocks=64, graph=False, seed=0,
              mean_tol=4e-3, engine_kw=None, max_ulps=2.0, oracle_kw=None, before_engine_prefill=None):
    """Prefill + n_steps decode steps on GPU engine and oracle; returns per-step records."""
    bs = 16
    B = len(prompt_lens)
    eng = _engine(d, w, kv_dtype=kv, num_blocks=num_blocks, block_size=bs, max_batch=B_pad or B, **(engine_kw or {}))
    orc = O.TalkerOracle(d, w, kv_dtype=kv, num_blocks=num_blocks, block_size=bs, **(oracle_kw or {}))
    pool = BlockPool(num_blocks, bs)
    g = torch.Generator().manual_seed(seed)
    prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in prompt_lens]
    tails = [[torch.randn(d.hidden, generator=g).to(BF16) for _ in range(k)] for k in ([2, 0, 5, 1] * 16)[:B]]
    pads = [torch.randn(d.hidden, generator=g).to(BF16) for _ in range(B)]
    for r, n in enumerate(prompt_lens):
        pool.allocate(f"r{r}", n + n_steps + 1)
    bts = [pool.block_ids(f"r{r}") for r in range(B)]
    # ---- oracle
    states = [O.OracleState(tail_text=list(tails[r]), tts_pad=pads[r]) for r in range(B)]
    samp = dict(sampling or {})
    greedy = not samp
    o_logits, o_ids, o_h = orc.prefill(states, prompts, bts, greedy=greedy, sampling=samp)
    # ---- engine prefill
    bt = torch.zeros(eng.max_batch, eng.bt_stride, dtype=torch.int32)
    for r in range(B):
        bt[r, :len(bts[r])] = torch.tensor(bts[r])
    eng.block_table.copy_(bt)
    x = torch.cat(prompts, 0)
    pos = torch.cat([torch.arange(n) for n in prompt_lens]).to(torch.int32)
    req = torch.cat([torch.full((n,), r) for r, n in enumerate(prompt_lens)]).to(torch.int32)
    slots = torch.tensor([bts[int(req[t])][int(pos[t]) // bs] * bs + int(pos[t]) % bs for t in range(x.shape[0])])
    assert torch.equal(slots, orc.last_slots)
    if before_engine_prefill is not None:
        before_engine_prefill(eng, orc)
    hid = eng.prefill(x.cuda(), pos.cuda(), req.cuda(), slots.cuda())
    last = torch.tensor(np.cumsum(prompt_lens) - 1)
    hl = hid[last.cuda()]
    assert_e2e_close(hl, o_h, mean_tol=mean_tol, max_ulps=max_ulps, what="prefill hidden")
    lg = eng.compute_logits(hl)
    rec = {"prefill_logits": (lg.cpu(), o_logits)}
    # hand the ORACLE's first token / hidden to the engine so later steps compare like for like
    eng.input_ids[:B] = o_ids.to(torch.int32).cuda()
    eng.last_hidden[:B] = o_h.cuda()
    eng.positions[:B] = torch.tensor(prompt_lens, dtype=torch.int32).cuda()
    eng.seq_lens[:B] = (torch.tensor(prompt_lens, dtype=torch.int32) + 1).cuda()
    eng.steps[:B] = 1
    if samp:
        eng.set_sampling(greedy=0, temperature=samp["temperature"], top_k=samp["top_k"], rep_penalty=samp.get("rep_penalty", 1.0),
                         seed=samp["seed"])
        eng.seen.zero_()
        eng.seen[:B, d.codec_pad_id] = 1
        for r in range(B):
            eng.seen[r, int(o_ids[r])] = 1
    steps = []
    gr = None
    for s in range(n_steps):
        text = torch.stack([(tails[r][s] if s < len(tails[r]) else pads[r]) for r in range(B)])
        eng.text_step[:B] = text.cuda()
        if graph:
            if gr is None:
                # warm up on a copy of the state, then capture (hipGraph via torch.cuda.CUDAGraph)
                keep = {n: getattr(eng, n).clone() for n in ("input_ids", "positions", "seq_lens", "last_hidden", "steps", "seen")}
                kvk = [c.clone() for c in eng.kv_caches]
                eng.decode_step(B)
                torch.cuda.synchronize()
                for n, v in keep.items():
                    getattr(eng, n).copy_(v)
                for c, v in zip(eng.kv_caches, kvk):
                    c.copy_(v)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    eng.decode_step(B)
                for n, v in keep.items():
                    getattr(eng, n).copy_(v)
                for c, v in zip(eng.kv_caches, kvk):
                    c.copy_(v)
            gr.replay()
        else:
            eng.decode_step(B)
        torch.cuda.synchronize()
        cp_kw = dict(do_sample=False)
        ol, oi, oh, oc, osl = orc.decode_step(states, bts, greedy=greedy, sampling=samp, cp_kw=cp_kw)
        steps.append(dict(
            slots=(eng.slot_mapping[:B].cpu(), osl), codes=(eng.audio_codes[:B].cpu(), oc),
            x=(eng.inputs_embeds[:B].cpu(), None), logits=(eng.logits[:B].cpu(), ol), ids=(eng.input_ids[:B].cpu(), oi),
            hidden=(eng.last_hidden[:B].cpu(), oh)))
        # keep both sides on the oracle's trajectory (a 1-ulp logit flip must not fork the run)
        eng.input_ids[:B] = oi.to(torch.int32).cuda()
        eng.last_hidden[:B] = oh.cuda()
    rec["steps"] = steps
    rec["engine"], rec["oracle"], rec["weights"] = eng, orc, w
    rec["first"] = (o_ids, o_h)
    return rec


def _codes_equal_up_to_near_ties(rec, i, w, tie_ulps=4.0):      # (3.0 while the step replayed the oracle's rounding points: tests/util.py)
    """Audio codes of step i: bit-exact, except that a row may leave the oracle's greedy path at a group whose two best
    (bf16-rounded) code-predictor logits are within `tie_ulps` bf16 ulps of each other -- after 5 predictor layers the two
    pipelines' logits differ by 1-2 ulp, so an argmax over values that close is decided by the summation order; everything
    after that group in the row follows a different input and is not compared."""
    got, ref = rec["steps"][i]["codes"]
    if torch.equal(got, ref):
        return []
    orc = rec["oracle"]
    ids_prev, h_prev = rec["first"] if i == 0 else (rec["steps"][i - 1]["ids"][1], rec["steps"][i - 1]["hidden"][1])
    bad = (got != ref).any(1).nonzero().flatten().tolist()
    _, lg = orc.code_predictor(ids_prev[bad], w["embed"][ids_prev[bad]], h_prev[bad], do_sample=False, return_logits=True)
    for j, b in enumerate(bad):
        gfirst = int((got[b] != ref[b]).nonzero()[0])
        assert gfirst >= 1, f"step {i} row {b}: layer-0 code differs"
        top = torch.topk(lg[j, gfirst - 1].float(), 2).values
        tie = tie_ulps * 2.0 ** (int(np.floor(np.log2(max(float(top[0].abs()), 1e-30)))) - 7)
        assert float(top[0] - top[1]) <= tie, f"step {i} row {b}: code group {gfirst} differs without a near-tie (margin {float(top[0] - top[1]):.4g} > {tie:.4g})"
    assert len(bad) <= max(1, got.shape[0] // 2), f"step {i}: {len(bad)} rows left the greedy path"
    return bad


def _check(rec, *, mean_tol=4e-3, max_ulps=2.0, weights=None):
    lg, ol = rec["prefill_logits"]
    assert_e2e_close(lg, ol, mean_tol=mean_tol, max_ulps=max_ulps, what="prefill logits")
    for i, st in enumerate(rec["steps"]):
        assert torch.equal(st["slots"][0], st["slots"][1]), f"step {i}: slot mapping must be bit-exact"
        keep = torch.ones(st["codes"][1].shape[0], dtype=torch.bool)
        # a row that left the greedy path at a verified near-tie feeds the backbone another frame: not comparable further
        keep[_codes_equal_up_to_near_ties(rec, i, weights if weights is not None else rec["weights"])] = False
        st["rows_compared"] = keep
        g, o = st["logits"]
        assert torch.equal(torch.isinf(g), torch.isinf(o)), f"step {i}: codec mask pattern"
        assert_e2e_close(g[keep], o[keep], mean_tol=mean_tol, max_ulps=max_ulps, what=f"step {i} logits")
        assert_e2e_close(st["hidden"][0][keep], st["hidden"][1][keep], mean_tol=mean_tol, max_ulps=max_ulps, what=f"step {i} hidden")
        ids_g, ids_o = st["ids"]
        for b in range(ids_o.shape[0]):
            if keep[b] and ids_g[b] != ids_o[b]:
                top = torch.topk(o[b], 2).values
                assert (top[0] - top[1]).item() <= 2 ** -6, f"step {i} row {b}: sampled id differs without a near-tie"


@pytest.mark.parametrize("kv", ["bf16", "fp8", "int8"])
def test_decode_steps_match_oracle_tiny(kv):
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.06, norm_noise=0.1)
    ulps = 4.0 if kv == "int8" else 2.0      # (int8: the per-token quantisation step on top of the rounding distance; 5-5.75 ulps measured in round 6)
    rec = _scenario(d, w, kv, prompt_lens=[5, 17, 33, 16], n_steps=6, mean_tol=6e-3, max_ulps=ulps)
    _check(rec, mean_tol=6e-3, max_ulps=ulps)      # tiny model uses 3x the BASELINE weight scale -> 3x the logit sensitivity
    # the KV cache bytes the runner exposes for KV transfer: same layout as the oracle's
    eng, orc = rec["engine"], rec["oracle"]
    for li in range(d.layers):
        got = eng.kv_caches[li].cpu()
        ref = orc.kv[li].data.view(torch.uint8) if kv == "fp8" else orc.kv[li].data
        if kv == "bf16":
            assert_e2e_close(got, ref, what=f"kv layer {li}")
        else:
            assert (got != ref).float().mean().item() < 0.15, f"kv bytes layer {li}"


@pytest.mark.parametrize("mode", ["engine-calibrates", "oracle-scales-eager", "oracle-scales-graph"])
def test_fp8_kv_scales_calibrated_by_the_first_prefill_match_the_oracle(mode):
    """cache_config.calculate_kv_scales (VERDICT r3 missing #3): the reference runs its FIRST forward eager so that vLLM's attention
    layers set k_scale = max|k| / 200, v_scale = max|v| / 100 per layer from that pass (V/worker/gpu_ar_model_runner.py:122,269-275;
    rule restated in the oracle's header).  Here the first prefill pass of the engine does it -- the same qknorm+RoPE kernel into
    a bf16 scratch, amax, fp32 division -- before that pass's own cache write; the decode step's attention reads the scales from a
    device table.  Against the oracle with calculate_kv_scales=True (NON-unit scales: |k| of a few units / 200, different per layer):
      * engine-calibrates: the engine's own scales within one bf16 ulp of the oracle's (max|k| of two bf16 pipelines may differ by an
        ulp of the one extreme element); a scale that differs at all re-rounds every fp8 byte of that layer, so the end-to-end bound
        is the fp8 noise level (6 ulps), not the accumulation-order level;
      * oracle-scales: the engine is GIVEN the oracle's scales (set_kv_scales) -- same scales, so cache bytes and every step meet the
        tight bound of the unit-scale tests: the per-layer scale plumbing (prefill arguments, device table of the decode step)."""
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.06, norm_noise=0.1)
    own = mode == "engine-calibrates"
    give = None if own else (lambda eng, orc: eng.set_kv_scales([kv.k_scale for kv in orc.kv], [kv.v_scale for kv in orc.kv]))
    tol = dict(mean_tol=1.2e-2, max_ulps=6.0) if own else dict(mean_tol=6e-3)
    rec = _scenario(d, w, "fp8", prompt_lens=[5, 17, 33, 16], n_steps=4, graph=mode.endswith("graph"), before_engine_prefill=give,
                    engine_kw=dict(calculate_kv_scales=own), oracle_kw=dict(calculate_kv_scales=True), **tol)
    eng, orc = rec["engine"], rec["oracle"]
    assert not eng.calibrate_pending and not orc.calculate_kv_scales
    ks, vs = [kv.k_scale for kv in orc.kv], [kv.v_scale for kv in orc.kv]
    assert len(set(ks)) == d.layers and all(abs(k - 1.0) > 0.5 for k in ks), f"the test needs non-unit, per-layer scales: {ks}"
    for l in range(d.layers):
        assert abs(eng.k_scale_l[l] / ks[l] - 1) <= 2 ** -7 and abs(eng.v_scale_l[l] / vs[l] - 1) <= 2 ** -7, (l, eng.k_scale_l[l], ks[l])
    _check(rec, weights=w, **tol)
    if not own:
        for li in range(d.layers):
            got, ref = eng.kv_caches[li].cpu(), orc.kv[li].data.view(torch.uint8)
            assert (got != ref).float().mean().item() < 0.15, f"kv bytes layer {li}"


def test_graph_captured_before_the_calibration_reads_the_calibrated_scales():
    """The decode graphs are captured at warm-up, before any request: the attention launches must read the scales the FIRST prefill
    sets later (omni_talker_set_kv_scales writes the device table they read) -- a graph captured before the calibration and an
    eager step after it produce the same bits."""
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.06, norm_noise=0.1)
    lens, B, bs = [7, 19, 12], 3, 16
    outs = []
    for early in (True, False):
        eng = _engine(d, w, kv_dtype="fp8", num_blocks=32, block_size=bs, max_batch=B, calculate_kv_scales=True)
        gr = None
        if early:
            eng.decode_step(B, advance=False)           # code objects
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                eng.decode_step(B)
            for c in eng.kv_caches:
                c.zero_()
        g = torch.Generator().manual_seed(2)
        prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
        bts = [[1 + 3 * r, 2 + 3 * r, 3 + 3 * r] for r in range(B)]
        for r in range(B):
            eng.block_table[r, :3] = torch.tensor(bts[r], dtype=torch.int32)
        x = torch.cat(prompts, 0)
        pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32)
        req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32)
        slots = torch.tensor([bts[int(req[t])][int(pos[t]) // bs] * bs + int(pos[t]) % bs for t in range(x.shape[0])])
        hid = eng.prefill(x.cuda(), pos.cuda(), req.cuda(), slots.cuda())
        assert not eng.calibrate_pending and abs(eng.k_scale_l[0] - 1.0) > 0.3
        last = torch.tensor(np.cumsum(lens) - 1)
        eng.input_ids[:B] = eng.compute_logits(hid[last.cuda()]).argmax(-1).to(torch.int32)
        eng.last_hidden[:B] = hid[last.cuda()]
        eng.positions[:B] = torch.tensor(lens, dtype=torch.int32).cuda()
        eng.seq_lens[:B] = (torch.tensor(lens, dtype=torch.int32) + 1).cuda()
        eng.steps[:B] = 1
        eng.seen.zero_()
        eng.text_step[:B] = 0
        frames = []
        for _ in range(3):
            gr.replay() if gr is not None else eng.decode_step(B)
            frames.append((eng.logits[:B].clone(), eng.audio_codes[:B].clone(), eng.last_hidden[:B].clone()))
        torch.cuda.synchronize()
        outs.append((frames, [c.view(torch.uint8).clone() for c in eng.kv_caches]))
    for s, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        for x, y in zip(a, b):
            assert torch.equal(x, y), f"step {s}: the early-captured graph and the eager step differ"
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("engine_kw", [dict(fused_norm=False), dict(frag_layout=False)],
                         ids=["separate-norms", "row-major"])
def test_decode_steps_alternate_layout_paths(engine_kw):
    """The default step keeps the residual stream fragment-major with every RMSNorm folded into its GEMMs; the
    separate-norm path (what tensor-parallel ranks run) and the row-major path must meet the same oracle bar."""
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.06, norm_noise=0.1)
    rec = _scenario(d, w, "fp8", prompt_lens=[5, 17, 33, 16], n_steps=4, mean_tol=6e-3, engine_kw=engine_kw)
    assert rec["engine"].fused_norm is False
    _check(rec, mean_tol=6e-3)


def test_default_engine_runs_the_norm_free_stream():
    d = get_dims("tiny")
    eng = _engine(d, make_weights(d, seed=5), kv_dtype="fp8", num_blocks=8, block_size=16, max_batch=4)
    assert eng.frag_layout and eng.fused_norm


def test_decode_step_hipgraph_replay_matches_eager():
    d = get_dims("tiny")
    w = make_weights(d, seed=6, std=0.06, norm_noise=0.1)
    eager = _scenario(d, w, "fp8", prompt_lens=[9, 20, 31], n_steps=4, B_pad=4)
    graph = _scenario(d, w, "fp8", prompt_lens=[9, 20, 31], n_steps=4, B_pad=4, graph=True)
    _check(graph)
    for a, b in zip(eager["steps"], graph["steps"]):
        for k in ("slots", "codes", "logits", "ids", "hidden"):
            assert torch.equal(a[k][0], b[k][0]), f"graph replay differs from eager on {k}"


def test_decode_steps_sampled_topk():
    d = get_dims("tiny")
    w = make_weights(d, seed=7, std=0.06, norm_noise=0.1)
    samp = dict(temperature=0.9, top_k=50, rep_penalty=1.05, seed=42)
    rec = _scenario(d, w, "fp8", prompt_lens=[7, 12, 30, 18, 5], n_steps=5, sampling=samp)
    agree = tot = 0
    c_agree = c_tot = 0
    for st in rec["steps"]:
        assert torch.equal(st["slots"][0], st["slots"][1])
        # greedy argmax over bf16-rounded code-predictor logits: a 1-ulp difference can flip an exact tie
        c_agree += int((st["codes"][0] == st["codes"][1]).sum())
        c_tot += st["codes"][1].numel()
        agree += int((st["ids"][0] == st["ids"][1]).sum())
        tot += st["ids"][1].numel()
    assert agree >= tot - 2, f"sampled ids agree {agree}/{tot}"
    assert c_agree >= c_tot - 3, f"codes agree {c_agree}/{c_tot}"


def test_real_dims_one_layer():
    """1.7B layer shapes (H=2048, 16/8 heads, I=6144, V=3072, cp 1024) with 1 backbone / 1 cp layer, B=64."""
    d = get_dims("tts-1.7b").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
    w = make_weights(d, seed=8, std=0.02)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(4, 40, (64,), generator=g).tolist()
    rec = _scenario(d, w, "fp8", prompt_lens=lens, n_steps=2, num_blocks=300, mean_tol=1e-3)
    _check(rec, mean_tol=1e-3)      # BASELINE weight scale: logits within 1e-3 in the mean, 1-2 bf16 ulp max


def test_unreleased_width_runs_both_chains_and_matches_oracle():
    """a width NO released checkpoint has -- hidden 1536, intermediate 4608, 12 q / 6 kv heads (the dimensions are the
    checkpoint's to decide: configuration_qwen3_tts.py:192-216) -- takes the persistent chains too (csrc/bb_chain.hip BB_SHAPES: the stage
    set is instantiated from the shape), reported by the native step (chains_ran == 3), against the oracle: prefill + 2 decode steps at 64
    rows, 2 backbone layers, the full 5-layer / 16-group predictor."""
    d = get_dims("tts-1.7b").with_(layers=2, max_model_len=256, hidden=1536, inter=4608, q_heads=12, kv_heads=6)
    w = make_weights(d, seed=19, std=0.02)
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(4, 40, (64,), generator=g).tolist()
    rec = _scenario(d, w, "fp8", prompt_lens=lens, n_steps=2, num_blocks=300, mean_tol=2e-3, max_ulps=3)
    _check(rec, mean_tol=2e-3, max_ulps=3)
    eng = rec["engine"]
    assert eng.persistent_chains and eng.chains_ran() == 3 and eng.chain_error() == 0, eng.chains_ran()


def test_prefill_wide_paths_match_chunked_native_and_oracle():
    """All-tokens-per-layer prefill with omni_gemm_tile on the fragment-major decode weights == the same pass with hipBLASLt
    GEMMs on row-major copies == native skinny-GEMM chunks == oracle."""
    d = get_dims("tiny")
    w = make_weights(d, seed=12, std=0.06, norm_noise=0.1)
    bs, nb = 16, 64
    lens = [5, 40, 17, 64]
    for kv in ("bf16", "fp8"):
        engs = [_engine(d, w, kv_dtype=kv, num_blocks=nb, block_size=bs, max_batch=8, prefill_gemm="both") for _ in range(3)]
        orc = O.TalkerOracle(d, w, kv_dtype=kv, num_blocks=nb, block_size=bs)
        pool = BlockPool(nb, bs)
        g = torch.Generator().manual_seed(1)
        prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
        for r, n in enumerate(lens):
            pool.allocate(f"r{r}", n)
        bts = [pool.block_ids(f"r{r}") for r in range(len(lens))]
        states = [O.OracleState() for _ in lens]
        _, _, o_h = orc.prefill(states, prompts, bts)
        x = torch.cat(prompts).cuda()
        pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32).cuda()
        req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32).cuda()
        outs = []
        for e, (wide, gemm) in zip(engs, ((False, None), (True, "tile"), (True, "blas"))):
            for r in range(len(lens)):
                e.block_table[r, :len(bts[r])] = torch.tensor(bts[r], dtype=torch.int32)
            outs.append(e.prefill(x, pos, req, orc.last_slots.cuda(), use_blas=wide, gemm=gemm))
        last = torch.tensor(np.cumsum(lens) - 1)
        for h in outs:
            assert_e2e_close(h[last.cuda()], o_h, mean_tol=6e-3, max_ulps=3, what=f"prefill hidden {kv}")
        assert_e2e_close(outs[0], outs[1], mean_tol=9e-3, max_ulps=4, what="tile vs chunked prefill")      # two implementations, each within 6e-3 of the oracle
        assert_e2e_close(outs[1], outs[2], mean_tol=9e-3, max_ulps=4, what="tile vs blas prefill")
        for li in range(d.layers):
            a, b = engs[0].kv_caches[li].cpu(), engs[1].kv_caches[li].cpu()
            assert (a != b).float().mean().item() < 0.1, "KV written by both prefill paths"
    # the default engine keeps ONE copy of the dense GEMM weights and refuses the path it has no weights for
    e = _engine(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs, max_batch=8)
    assert e.prefill_gemm == "tile" and "wqkv" not in e.layer_w[0] and "wqkv_f" in e.layer_w[0]
    with pytest.raises(L.OmniError):
        e.prefill(x, pos, req, orc.last_slots.cuda(), use_blas=True, gemm="blas")


@pytest.mark.parametrize("fp8w", [False, True])
def test_moe_prefill_grouped_tile_matches_batched_blas_and_oracle(fp8w):
    """The Omni talker's MoE prefill at its real expert shapes (128 experts top-8 of width 384, shared 768; 2 layers): router,
    grouped expert GEMMs and shared expert on omni_gemm_tile (fragment-major decode weights) == the batched hipBLASLt arm ==
    oracle at the last prompt rows; bf16 and fp8 expert weights (the latter: everyone sees bf16(fp8 * scale))."""
    from ht_vllm_omni_amd.engine import fp8_quant_rows, fp8_dequant_rows
    d = get_dims("omni-talker").with_(layers=2, cp_layers=1, num_code_groups=3, max_model_len=1024)
    w = make_weights(d, seed=23, std=0.02)
    ow = w
    if fp8w:
        ow = dict(w)
        for li in range(d.layers):
            for n in ("moe_gate_up", "moe_down"):
                q8, sc = fp8_quant_rows(w[f"l{li}.{n}"])
                ow[f"l{li}.{n}"] = fp8_dequant_rows(q8, sc)
    bs, nb = 16, 80
    lens = [300, 257, 140, 9]
    engs = [_engine(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs, max_batch=8, moe_fp8=fp8w, prefill_gemm="both") for _ in range(2)]
    orc = O.TalkerOracle(d, ow, kv_dtype="bf16", num_blocks=nb, block_size=bs)
    pool = BlockPool(nb, bs)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
    for r, n in enumerate(lens):
        pool.allocate(f"r{r}", n)
    bts = [pool.block_ids(f"r{r}") for r in range(len(lens))]
    states = [O.OracleState() for _ in lens]
    _, _, o_h = orc.prefill(states, prompts, bts)
    x = torch.cat(prompts).cuda()
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32).cuda()
    req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32).cuda()
    outs = []
    for e, gemm in zip(engs, ("tile", "blas")):
        for r in range(len(lens)):
            e.block_table[r, :len(bts[r])] = torch.tensor(bts[r], dtype=torch.int32)
        outs.append(e.prefill(x, pos, req, orc.last_slots.cuda(), use_blas=True, gemm=gemm))
    last = torch.tensor(np.cumsum(lens) - 1)
    # a routing near-tie may send a token to a different 8th expert in either arm: compare the bulk, bound the outliers
    diff = (outs[0].float() - outs[1].float()).abs().mean(-1) / outs[1].float().abs().mean(-1)
    assert (diff > 0.02).float().mean().item() < 0.02, f"{int((diff > 0.02).sum())} of {diff.numel()} rows differ between the arms"
    for h in outs:
        dr = (h[last.cuda()].float().cpu() - o_h.float()).abs().mean(-1) / o_h.float().abs().mean(-1)
        assert dr.median().item() < 8e-3 and (dr < 0.03).sum().item() >= len(lens) - 1, dr
    # the tile arm holds no [E, K, N] copies unless asked for both
    e = _engine(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs, max_batch=8, moe_fp8=fp8w)
    assert "moe_gate_up_t" not in e.layer_w[0] and "moe_gate_up_f" in e.layer_w[0] and "moe_gate_up" not in e.layer_w[0]


def test_tp_collective_path_captured_in_hipgraph():
    """The tensor-parallel step (phase calls + RCCL all-reduce after o_proj and down_proj) on a 1-rank nccl group:
    eager and hipGraph replay agree bit for bit and reproduce the single-call step to rounding.  (N > 1 needs the driver's 8-GPU node.)"""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        d = get_dims("tiny")
        w = make_weights(d, seed=13, std=0.06, norm_noise=0.1)
        outs = []
        for mode in ("single", "tp_eager", "tp_graph"):
            eng = _engine(d, w, kv_dtype="fp8", num_blocks=32, max_batch=4, tp_force=(mode != "single"))
            g = torch.Generator().manual_seed(2)
            B = 3
            eng.block_table[:B, :2] = torch.tensor([[1, 2], [3, 4], [5, 6]], dtype=torch.int32)
            eng.positions[:B] = torch.tensor([3, 9, 17], dtype=torch.int32)
            eng.seq_lens[:B] = eng.positions[:B] + 1
            eng.input_ids[:B] = torch.tensor([5, 9, 77], dtype=torch.int32)
            eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
            eng.text_step[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
            for c in eng.kv_caches:
                c.copy_(torch.randint(0, 100, c.shape, generator=g, dtype=torch.uint8))
            if mode == "tp_graph":
                keep = {n: getattr(eng, n).clone() for n in ("input_ids", "positions", "seq_lens", "last_hidden", "steps")}
                kvk = [c.clone() for c in eng.kv_caches]
                eng.decode_step(B)                       # warm-up (RCCL communicator, code objects)
                torch.cuda.synchronize()
                for n, v in keep.items():
                    getattr(eng, n).copy_(v)
                for c, v in zip(eng.kv_caches, kvk):
                    c.copy_(v)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    eng.decode_step(B)
                gr.replay()
            else:
                eng.decode_step(B)
            torch.cuda.synchronize()
            outs.append((eng.logits[:B].cpu(), eng.audio_codes[:B].cpu(), eng.input_ids[:B].cpu(), eng.slot_mapping[:B].cpu()))
        # eager and hipGraph replay of the collective path: bit for bit
        for a, b in zip(outs[1], outs[2]):
            assert torch.equal(a, b)
        # ... and against the single-call step: the collective path runs the separate RMSNorm kernels (the reference's rounding points), the
        # single-rank step the norm-fused GEMMs with the rstd applied to the fp32 sums (round 6) -- two bf16 pipelines one rounding apart
        # (bit-identical until round 5): same slots, logits to rounding on the rows whose code frames agree
        assert torch.equal(outs[0][3], outs[1][3])
        same = (outs[0][1] == outs[1][1]).all(-1)
        assert int(same.sum()) >= 2
        assert_e2e_close(outs[1][0][same], outs[0][0][same], mean_tol=6e-3, max_ulps=3, what="collective path vs single-call step: logits")
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("fused", [True, False], ids=["norm-free-stream", "separate-norms"])
def test_moe_backbone_decode_steps_match_oracle(fused):
    """Row a11 end to end: the talker with a sparse-MoE MLP in every backbone layer (Omni talker in miniature: 16 experts,
    top-4, gated shared expert, no code-predictor projection) -- prefill + decode steps through the native engine against
    the oracle whose MoE block is pinned to HF's module.  Default (round 3): the MoE layers on the norm-free residual stream --
    router and shared expert with the fused-norm prologue, the combine adding into the fragment-major residual and writing the
    sum-of-squares slabs (omni_moe_experts_resid); the separate-norm path stays for RCCL-reduced tensor-parallel ranks."""
    d = get_dims("omni-moe-tiny")
    assert d.moe_experts == 16 and not d.has_cp_projection
    w = make_weights(d, seed=15, std=0.06, norm_noise=0.1)
    rec = _scenario(d, w, "fp8", prompt_lens=[5, 17, 33, 16, 9], n_steps=4, mean_tol=8e-3, engine_kw=None if fused else dict(fused_norm=False))
    assert rec["engine"].fused_norm is fused
    lg, ol = rec["prefill_logits"]
    assert_e2e_close(lg, ol, mean_tol=8e-3, max_ulps=3, what="MoE prefill logits")
    bad_rows = 0
    for i, st in enumerate(rec["steps"]):
        assert torch.equal(st["slots"][0], st["slots"][1]), f"step {i}: slot mapping must be bit-exact"
        same = (st["codes"][0] == st["codes"][1]).all(-1)
        bad_rows += int((~same).sum())
        g, o = st["logits"]
        assert_e2e_close(g[same], o[same], mean_tol=8e-3, max_ulps=3, what=f"MoE step {i} logits")
    # a routing near-tie (bf16 router logits) may send a token to a different 4th expert than the oracle's torch.topk
    assert bad_rows <= 2, f"{bad_rows} rows diverged"


def test_mrope_prompt_with_differing_rows_and_decode_offset_match_oracle():
    """A prompt whose M-RoPE ids differ between rows (an image block in the middle: ids from positions.omni_input_positions,
    pinned to the reference by tests/golden/mrope_positions.json) through prefill and three decode steps: the prefill kernel
    rotates every pair by its axis' id, the decode steps by index + mrope_position_delta (engine.rope_delta, read inside the
    captured step), cache slots and attention context by the plain index.  Against the oracle's restatement of vLLM's
    MRotaryEmbedding; a second request in the same batch keeps plain ids (delta 0)."""
    import types
    from ht_vllm_omni_amd import positions as P
    d = get_dims("omni-moe-tiny")
    assert d.mrope_section == (24, 20, 20) and d.mrope_interleaved
    w = make_weights(d, seed=31, std=0.06, norm_noise=0.1)
    bs, nb, n_steps = 16, 32, 3
    ids = dict(audio_token_index=901, image_token_index=902, video_token_index=903, audio_start_token_id=904, audio_end_token_id=905,
               vision_start_token_id=906, vision_end_token_id=907, seconds_per_chunk=2)
    cfg = types.SimpleNamespace(thinker_config=types.SimpleNamespace(vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=25), **ids))
    toks = [11, 12, 906] + [902] * 24 + [907, 13, 14, 15]
    mp, delta = P.get_input_positions_tensor(toks, cfg, [[1, 8, 12]], [], [])
    assert delta == -18 and not torch.equal(mp[1], mp[2])
    lens = [len(toks), 9]
    rope = torch.cat([mp, torch.arange(lens[1]).expand(3, -1)], 1)
    deltas = [delta, 0]
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs, max_batch=4)
    orc = O.TalkerOracle(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs)
    pool = BlockPool(nb, bs)
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in lens]
    pads = [torch.randn(d.hidden, generator=g).to(BF16) for _ in lens]
    for r, n in enumerate(lens):
        pool.allocate(f"r{r}", n + n_steps + 1)
    bts = [pool.block_ids(f"r{r}") for r in range(2)]
    states = [O.OracleState(tts_pad=pads[r], rope_delta=deltas[r]) for r in range(2)]
    o_logits, o_ids, o_h = orc.prefill(states, prompts, bts, rope_positions=rope)
    plain_h = O.TalkerOracle(d, w, kv_dtype="bf16", num_blocks=nb, block_size=bs).prefill([O.OracleState(tts_pad=pads[r]) for r in range(2)], prompts, bts)[2]
    assert (plain_h[0].float() - o_h[0].float()).abs().mean() > 10 * (plain_h[1].float() - o_h[1].float()).abs().mean() + 1e-3   # the ids matter
    for r in range(2):
        eng.block_table[r, :len(bts[r])] = torch.tensor(bts[r], dtype=torch.int32)
    x = torch.cat(prompts).cuda()
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32).cuda()
    req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32).cuda()
    hid = eng.prefill(x, pos, req, orc.last_slots.cuda(), rope_positions=rope.to(torch.int32).cuda())
    last = torch.tensor(np.cumsum(lens) - 1)
    assert_e2e_close(hid[last.cuda()], o_h, mean_tol=8e-3, max_ulps=3, what="M-RoPE prefill hidden")
    eng.input_ids[:2] = o_ids.to(torch.int32).cuda()
    eng.last_hidden[:2] = o_h.cuda()
    eng.positions[:2] = torch.tensor(lens, dtype=torch.int32).cuda()
    eng.seq_lens[:2] = (torch.tensor(lens, dtype=torch.int32) + 1).cuda()
    eng.rope_delta[:2] = torch.tensor(deltas, dtype=torch.int32).cuda()
    eng.steps[:2] = 1
    compared = 0
    for s in range(n_steps):
        eng.text_step[:2] = torch.stack(pads).cuda()
        eng.decode_step(2)
        lg, ids_o, h_o, codes_o, slots_o = orc.decode_step(states, bts)
        assert torch.equal(eng.slot_mapping[:2].cpu(), slots_o), "cache slots follow the plain index, not the rotary id"
        # (a row whose greedy codes left the oracle's at a near-tie decodes another input embedding: compared while on the oracle's frame)
        on = (eng.audio_codes[:2].cpu() == codes_o).all(-1)
        compared += int(on.sum())
        if on.any():
            assert_e2e_close(eng.last_hidden[:2].cpu()[on], h_o[on], mean_tol=8e-3, max_ulps=3, what=f"M-RoPE decode step {s} hidden")
        eng.input_ids[:2] = ids_o.to(torch.int32).cuda()        # keep both on the oracle's token (a routing near-tie may flip one)
        eng.last_hidden[:2] = h_o.cuda()
    assert compared >= n_steps, f"only {compared} of {2 * n_steps} (row, step) pairs stayed on the oracle's code frames"
    # the offset matters: the same step without it lands elsewhere
    assert eng.rope_delta[:2].tolist() == deltas


def test_omni_finite_suppression_value():
    """The Omni talker's compute_logits writes -1e9 for suppressed tokens instead of -inf (qwen3_omni.py:1143-1149): same
    picks, finite logits; desc.masked_logit carries the value into the lm_head epilogue of the step and of omni_talker_logits."""
    d = get_dims("tiny")
    w = make_weights(d, seed=5, std=0.08, norm_noise=0.1)
    out = {}
    for fill in (0.0, -1e9):
        eng = _engine(d, w, kv_dtype="bf16", num_blocks=16, max_batch=4, masked_logit=fill)
        orc = O.TalkerOracle(d, w, masked_logit=float("-inf") if fill == 0.0 else fill)
        h = torch.randn(4, d.hidden, generator=torch.Generator().manual_seed(2)).to(BF16)
        lg = eng.compute_logits(h.cuda()).cpu()
        ref = orc.compute_logits(h)
        masked = ~orc.allowed
        assert masked.any() and (~masked).any()
        if fill == 0.0:
            assert torch.isinf(lg[:, masked]).all() and (lg[:, masked] < 0).all()
        else:
            assert (lg[:, masked] == -1e9).all()
        assert_f32_close(lg[:, ~masked], ref[:, ~masked], atol=0.02, what="unmasked logits")
        out[fill] = (lg, ops_sample(eng, lg))
    assert torch.equal(out[0.0][1], out[-1e9][1]), "greedy picks do not depend on the suppression value"


def ops_sample(eng, lg):
    from ht_vllm_omni_amd import ops
    return ops.sample(lg.cuda(), greedy=True).cpu()


def test_omni_talker_real_dims_one_layer():
    """The Omni talker's real layer shapes (H 1024, 16 q / 2 kv heads = four virtual groups of 4 per kv head, 128 experts
    top-8 of width 384 + shared 768) with one backbone / one predictor layer, B = 64, int8 KV (BASELINE config #4)."""
    d = get_dims("omni-talker").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
    w = make_weights(d, seed=21, std=0.02)
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(4, 40, (64,), generator=g).tolist()
    rec = _scenario(d, w, "int8", prompt_lens=lens, n_steps=2, num_blocks=300, mean_tol=2e-3)
    lg, ol = rec["prefill_logits"]
    assert_e2e_close(lg, ol, mean_tol=2e-3, max_ulps=3, what="omni talker prefill logits")
    diverged = 0
    for i, st in enumerate(rec["steps"]):
        assert torch.equal(st["slots"][0], st["slots"][1]), f"step {i}: slot mapping must be bit-exact"
        same = (st["codes"][0] == st["codes"][1]).all(-1)
        diverged += int((~same).sum())
        assert_e2e_close(st["logits"][0][same], st["logits"][1][same], mean_tol=2e-3, max_ulps=3, what=f"omni talker step {i} logits")
    assert diverged <= 16, f"{diverged} of 128 rows took a different expert / code at a near-tie"      # (6 until round 6: see tests/util.py)


@pytest.mark.parametrize("model,tp", [("tiny", 2), ("tts-1.7b-1layer", 8), ("tts-1.7b-1layer", 4), ("omni-talker-1layer", 2),
                                      ("omni-talker-1layer-fp8w", 8)])
def test_tp_sharded_engines_in_lockstep_match_oracle(model, tp):
    """Tensor parallel on ONE GPU: all rank engines of a TP group (sharded qkv / o / gate_up / down, one
    KV head each, replicated code predictor) are driven phase by phase, the two all-reduces of every layer replaced by
    adding the two partial buffers -- the sharded kernels, the phase API and the KV-head split of the cache against the
    unsharded oracle.  (The RCCL all-reduce itself: test_tp_collective_path_captured_in_hipgraph, 1-rank group.)"""
    import ctypes as C
    from ht_vllm_omni_amd import _lib as L
    kv, moe_fp8, ow = "bf16", False, None
    if model == "tiny":
        d = get_dims("tiny")
        w = make_weights(d, seed=17, std=0.06, norm_noise=0.1)
    elif model.startswith("omni-talker"):
        # BASELINE config #4: the Omni talker's real layer shape (sparse MoE: 128 experts top-8 of width 384 + shared 768,
        # 16 q / 2 kv heads) under TP = 2 with int8 KV -- experts split over their intermediate dimension (192 columns per
        # rank); config #5's regime: 8 ranks = expert parallel (16 experts per rank), fp8 e4m3fn expert weights, fp8 KV
        from ht_vllm_omni_amd.engine import fp8_dequant_rows, fp8_quant_rows, moe_parallel_mode
        d = get_dims("omni-talker").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
        w = make_weights(d, seed=17, std=0.02)
        moe_fp8 = model.endswith("fp8w")
        kv = "fp8" if moe_fp8 else "int8"
        assert moe_parallel_mode(d, tp) == ("ep" if tp == 8 else "tp")
        if moe_fp8:                                   # the oracle computes on the dequantised matrices the engine's kernels rebuild
            ow = dict(w)
            for n in ("l0.moe_gate_up", "l0.moe_down"):
                ow[n] = fp8_dequant_rows(*fp8_quant_rows(w[n]))
    else:                                             # real layer shapes: TP = 8 -> 2 q heads / 1 kv head / 768 columns per rank
        d = get_dims("tts-1.7b").with_(layers=1, cp_layers=1, num_code_groups=3, max_model_len=256)
        w = make_weights(d, seed=17, std=0.02)
    bs, nb, n_steps = 16, 32, 3
    prompt_lens = [5, 17, 33]
    B = len(prompt_lens)
    orc = O.TalkerOracle(d, ow or w, kv_dtype=kv, num_blocks=nb, block_size=bs)
    pool = BlockPool(nb, bs)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randn(n, d.hidden, generator=g).to(BF16) for n in prompt_lens]
    pads = [torch.randn(d.hidden, generator=g).to(BF16) for _ in range(B)]
    for r, n in enumerate(prompt_lens):
        pool.allocate(f"r{r}", n + n_steps + 1)
    bts = [pool.block_ids(f"r{r}") for r in range(B)]
    states = [O.OracleState(tail_text=[], tts_pad=pads[r]) for r in range(B)]
    _, o_ids, o_h = orc.prefill(states, prompts, bts, greedy=True, sampling={})
    engs = [_engine(d, w, kv_dtype=kv, num_blocks=nb, block_size=bs, max_batch=B, tp_rank=r, tp_size=tp, moe_fp8=moe_fp8) for r in range(tp)]
    lib = engs[0].lib
    moe = d.moe_experts > 0

    def kv_head_slice(e, r):                             # more ranks than KV heads: heads replicate (vLLM QKVParallelLinear)
        h0 = r * e.hkv_l if tp <= d.kv_heads else r // (tp // d.kv_heads)
        return slice(h0, h0 + e.hkv_l)

    for r, e in enumerate(engs):
        assert e.tp_path and e.hq_l == d.q_heads // tp and e.hkv_l == max(d.kv_heads // tp, 1) and not e.fused_norm and e.cp_fused_norm
        for li in range(d.layers):                       # this rank's KV head of the prefilled cache
            src = orc.kv[li].data.view(torch.uint8) if kv == "fp8" else orc.kv[li].data
            e.kv_caches[li].copy_(src[:, :, :, kv_head_slice(e, r)].cuda())
            if kv == "int8":
                e.kv_scales[li].copy_(orc.kv[li].scales[:, :, :, kv_head_slice(e, r)].cuda())
        bt = torch.zeros(e.max_batch, e.bt_stride, dtype=torch.int32)
        for q in range(B):
            bt[q, :len(bts[q])] = torch.tensor(bts[q])
        e.block_table.copy_(bt)
        e.input_ids[:B] = o_ids.to(torch.int32).cuda()
        e.last_hidden[:B] = o_h.cuda()
        e.positions[:B] = torch.tensor(prompt_lens, dtype=torch.int32).cuda()
        e.seq_lens[:B] = (torch.tensor(prompt_lens, dtype=torch.int32) + 1).cuda()
        e.steps[:B] = 1
        e.text_step[:B] = torch.stack(pads).cuda()
    st = L.current_stream()

    def all_reduce(name):
        tot = sum(getattr(e, name)[:B].float() for e in engs)       # fp32 sum of the bf16 partials, one rounding
        for e in engs:
            getattr(e, name)[:B] = tot.to(BF16)

    for s in range(n_steps):
        ios = [e._io(B, True) for e in engs]
        for e, io in zip(engs, ios):
            L.check(lib.omni_talker_mtp(e.handle, C.byref(io), st), "mtp")
        for li in range(d.layers):
            for e, io in zip(engs, ios):
                L.check(lib.omni_talker_layer_attn(e.handle, C.byref(io), li, st), "layer_attn")
            all_reduce("_attn_out")
            for e, io in zip(engs, ios):
                L.check(lib.omni_talker_layer_mlp(e.handle, C.byref(io), li, st), "layer_mlp")
            all_reduce("_mlp_out")
        for e, io in zip(engs, ios):
            L.check(lib.omni_talker_finish(e.handle, C.byref(io), st), "finish")
        torch.cuda.synchronize()
        ol, oi, oh, oc, osl = orc.decode_step(states, bts, greedy=True, sampling={}, cp_kw=dict(do_sample=False))
        for r, e in enumerate(engs):
            assert torch.equal(e.slot_mapping[:B].cpu(), osl), f"step {s} rank {r}: slots"
            on = codes_on_the_oracles_frame(e.audio_codes[:B], oc, orc.last_cp_logits, what=f"step {s} rank {r}")
            # sparse MoE: a routing near-tie may send a row to another k-th expert than torch.topk's (cf. the single-rank MoE test)
            rows = on if not moe else (on & ((e.logits[:B].cpu().nan_to_num(neginf=0) - ol.nan_to_num(neginf=0)).abs().amax(1) < 0.25))
            assert int(rows.sum()) >= B - 1 - int((~on).sum()) and int(on.sum()) >= (B + 1) // 2
            assert_e2e_close(e.logits[:B].cpu()[rows], ol[rows], mean_tol=6e-3, max_ulps=3, what=f"step {s} rank {r} logits")
            assert_e2e_close(e.last_hidden[:B].cpu()[rows], oh[rows], mean_tol=6e-3, max_ulps=3, what=f"step {s} rank {r} hidden")
        assert torch.equal(engs[0].logits[:B], engs[1].logits[:B]), "ranks must agree bit for bit (replicated tail)"
        for e in engs:                                   # stay on the oracle's trajectory
            e.input_ids[:B] = oi.to(torch.int32).cuda()
            e.last_hidden[:B] = oh.cuda()
    # each rank's cache holds its own KV head of the new tokens
    for li in range(d.layers):
        for r, e in enumerate(engs):
            if kv == "bf16":
                assert_e2e_close(e.kv_caches[li].cpu(), orc.kv[li].data[:, :, :, kv_head_slice(e, r)], what=f"kv layer {li} rank {r}")
            else:
                ref = (orc.kv[li].data.view(torch.uint8) if kv == "fp8" else orc.kv[li].data)[:, :, :, kv_head_slice(e, r)]
                assert (e.kv_caches[li].cpu() != ref).float().mean().item() < 0.15, f"kv bytes layer {li} rank {r}"


def test_decode_is_run_to_run_deterministic():
    """No float atomics anywhere on the path (sum-of-squares slabs, split-KV merges and MoE accumulation all run in a fixed
    order): two engines fed the same state produce bit-identical logits, ids, codes and KV bytes over sampled steps."""
    d = get_dims("tts-1.7b").with_(layers=2, cp_layers=2, num_code_groups=5, max_model_len=256)
    w = make_weights(d, seed=3, std=0.02)
    samp = dict(temperature=0.9, top_k=50, rep_penalty=1.05, seed=42)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(4, 60, (48,), generator=g).tolist()
    runs = []
    for _ in range(2):
        rec = _scenario(d, w, "fp8", prompt_lens=lens, n_steps=3, num_blocks=400, sampling=samp, graph=True, mean_tol=8e-3)
        runs.append(rec)
    for a, b in zip(runs[0]["steps"], runs[1]["steps"]):
        for k in ("slots", "codes", "logits", "ids", "hidden"):
            assert torch.equal(a[k][0], b[k][0]), f"{k} differs between two identical runs"
    for ca, cb in zip(runs[0]["engine"].kv_caches, runs[1]["engine"].kv_caches):
        assert torch.equal(ca, cb)
