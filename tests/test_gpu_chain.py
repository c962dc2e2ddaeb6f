"""The code predictor's layer stack as persistent launches (csrc/cp_chain.hip) against the launch-per-op path of the SAME
library: the chain keeps tiles, k-step ownership, accumulation order and rounding points, so codes AND logits must be
bit-identical; the flag waits must never time out (error word 0).  Reference function: the decoder loop of
qwen3_tts_code_predictor_vllm.py:480-561 (parity of either schedule with the oracle: tests/test_gpu_engine.py)."""
import ctypes as C

import pytest
import torch

from ht_vllm_omni_amd import _lib as L
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights
from oracle import talker_oracle as O
from tests.util import BF16, assert_e2e_close

pytestmark = pytest.mark.gpu


def _engine(d, w, **kw):
    from ht_vllm_omni_amd.engine import TalkerEngine
    return TalkerEngine(d, w, **kw)


def _inputs(d, w, B, seed):
    g = torch.Generator().manual_seed(seed)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    return code0, w["embed"][code0], torch.randn(B, d.hidden, generator=g).to(BF16)


@pytest.mark.parametrize("mode", [(2, 7, 1, 1), (1, 7, 1, 1), (2, 8, 1, 4), (2, 6, 0, 1)],
                         ids=["all-passes", "per-pass", "all-passes-all-flags-nap4", "all-passes-wide-gate_up"])
@pytest.mark.parametrize("model,B", [("tts-1.7b", 64), ("tts-1.7b", 37), ("tts-1.7b", 5), ("tts-0.6b", 16), ("tts-1.7b", 32), ("tts-1.7b", 48), ("tts-1.7b", 17)])
def test_chain_is_bit_identical_to_the_launch_chain(model, B, mode):
    """mode = (span, flag domain, gate_up tile, poll pause): span 1 = one persistent launch per pass (layer stack only), 2 = ONE
    launch for every pass with its head GEMM and sampler.  The 48-column gate_up tile sums the RMSNorm statistics in the
    16-row order, which is the launch path's order only up to 32 rows: above that the comparison is on rounding distance."""
    d = get_dims(model).with_(layers=1, max_model_len=256)          # the released predictor: 5 layers, 16 groups
    w = make_weights(d, seed=33, std=0.02)
    code0, e0, lh = _inputs(d, w, B, B)
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_cp_chain.argtypes = [C.c_int]; lib.omni_debug_cp_chain.restype = None
        lib.omni_debug_chain_mode.argtypes = [C.c_int, C.c_int, C.c_int]; lib.omni_debug_chain_mode.restype = None
        try:
            for on in (0, 1):
                lib.omni_debug_cp_chain(mode[0] if on else 0)
                lib.omni_debug_chain_mode(*mode[1:])
                eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=64)
                for rep in range(3):                                 # flags / epochs carry over from call to call
                    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
                    steps = torch.full((B,), 3 + rep, dtype=torch.int32)
                    sampled = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=False, temperature=0.9,
                                                 top_k=50, seed=7, steps=steps.cuda())
                assert eng.chain_error() == 0
                res[on] = (codes.cpu(), lg.cpu(), sampled.cpu())
        finally:
            lib.omni_debug_cp_chain(2)
            lib.omni_debug_chain_mode(7, 1, 1)
    if mode[2] == 0 and B > 32:          # other summation order of the norm statistics: equal to rounding, not to the bit
        # (rows whose greedy codes forked at a near-tie read other inputs from that group on: the logits are compared on the rows that stayed
        #  together -- until round 6 no row of this seed forked and the comparison silently covered all of them)
        same = (res[1][0] == res[0][0]).all(dim=1)
        assert same.float().mean().item() >= 0.8
        assert_e2e_close(res[1][1][same], res[0][1][same], mean_tol=2e-3, max_ulps=3, what="wide gate_up tile vs launch chain")
        return
    bad = (res[1][0] != res[0][0]).any(dim=1).nonzero().flatten().tolist()
    assert torch.equal(res[1][0], res[0][0]), f"greedy codes differ between the persistent chain and the launch chain in rows {bad}: first differing group per row {[int((res[1][0][b] != res[0][0][b]).nonzero()[0]) for b in bad]}"
    assert torch.equal(res[1][1], res[0][1]), "logits differ between the persistent chain and the launch chain"
    assert torch.equal(res[1][2], res[0][2]), "sampled codes differ between the persistent chain and the launch chain"


def test_chain_matches_oracle_at_full_predictor_depth():
    """Product library, chain on (the default): greedy codes / logits of the 5-layer, 16-group predictor vs the oracle."""
    d = get_dims("tts-1.7b").with_(layers=1, max_model_len=256)
    w = make_weights(d, seed=21, std=0.02)
    B = 48
    code0, e0, lh = _inputs(d, w, B, 5)
    eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=64)
    codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
    assert eng.chain_error() == 0
    ref_codes, ref_lg = O.TalkerOracle(d, w).code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    assert_e2e_close(lg.cpu()[:, 0], ref_lg[:, 0], mean_tol=6e-3, max_ulps=3, what="chain: code predictor logits, group 1 (5 layers)")
    assert (codes.cpu()[:, 1] == ref_codes[:, 1]).float().mean().item() >= 0.9


def _decode_engine(d, w, B, kv="fp8"):
    eng = _engine(d, w, kv_dtype=kv, num_blocks=256, max_batch=64)
    g = torch.Generator().manual_seed(9)
    eng.input_ids[:B] = torch.randint(1, d.codebook, (B,), generator=g).to(torch.int32).cuda()
    eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
    eng.text_step[:B] = (torch.randn(B, d.hidden, generator=g) * 0.02).to(BF16).cuda()
    eng.positions[:B] = 17
    eng.seq_lens[:B] = 18
    for b in range(B):
        eng.block_table[b, :3] = torch.tensor([1 + 3 * b, 2 + 3 * b, 3 + 3 * b], dtype=torch.int32)
    for c in eng.kv_caches:        # a history to attend to (same bytes in every engine built from this seed)
        c.view(torch.uint8).copy_(torch.randint(0, 120, c.view(torch.uint8).shape, generator=g, dtype=torch.uint8).cuda())
    eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
    return eng


# the released 1.7B width, and one NO released checkpoint has (round 6, VERDICT r5 item 4: the stage sets are instantiated from the shape --
# csrc/bb_chain.hip BB_SHAPES -- not from two literals): hidden 1536, intermediate 4608, 12 q / 6 kv heads (qkv 3072 wide)
UNRELEASED = dict(hidden=1536, inter=4608, q_heads=12, kv_heads=6)


@pytest.mark.parametrize("mode,B,kv,shape", [("engine", 64, "fp8", None), ("plain-chain", 64, "fp8", None), ("deep-rings", 64, "fp8", None),
                                             ("engine", 49, "bf16", None), ("plain-chain", 49, "bf16", None), ("deep-rings", 49, "bf16", None),
                                             ("plain-chain", 64, "fp8", UNRELEASED), ("plain-chain", 41, "int8", UNRELEASED)])
def test_backbone_segment_chain_against_the_launch_path(B, kv, mode, shape):
    """o_proj -> gate_up -> down_proj -> next qkv as one persistent launch per layer against the launch-per-op backbone of the
    same library at the 1.7B shape.  The plain chain (csrc/bb_chain.hip) and the loader / consumer engine (bb_engine.hip) keep
    the launch path's tiles and summation order: logits, hidden state, sampled ids, codes and every KV byte of three decode
    steps are identical.  (Rounds 3's two-group and split-role arms -- bb_pp.hip, bb_xw.hip: both lost -- left the tree in round 6, when
    the deferred rstd became the product arithmetic; NOTEBOOK "Round 3" keeps their measurements.)  No flag wait times out in any mode."""
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256, **(shape or {}))
    w = make_weights(d, seed=8, std=0.02)
    res = {}
    with L.debug_library() as lib:
        for fn in (lib.omni_debug_bb_chain, lib.omni_debug_bb_engine, lib.omni_debug_bb_deep):
            fn.argtypes = [C.c_int]; fn.restype = None
        try:
            for on in (0, 1):
                lib.omni_debug_bb_chain(on)
                lib.omni_debug_bb_engine(int(mode == "engine"))
                lib.omni_debug_bb_deep(int(mode == "deep-rings"))
                eng = _decode_engine(d, w, B, kv)
                outs = []
                for _ in range(3):
                    eng.decode_step(B)
                    outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.input_ids[:B].clone(), eng.audio_codes[:B].clone()))
                torch.cuda.synchronize()
                assert eng.chain_error() == 0
                assert eng.chains_ran() == (3 if on else 1), (on, eng.chains_ran(), "the shape must run BOTH persistent chains")
                res[on] = (outs, [c.view(torch.uint8).clone() for c in eng.kv_caches])
        finally:
            lib.omni_debug_bb_chain(1)
            lib.omni_debug_bb_engine(0)
            lib.omni_debug_bb_deep(0)
    for s, (a, b) in enumerate(zip(res[1][0], res[0][0])):
        for name, x, y in zip(("logits", "hidden", "ids", "codes"), a, b):
            assert torch.equal(x, y), f"step {s}: {name} differ between the backbone chain and the launch path"
    for l, (x, y) in enumerate(zip(res[1][1], res[0][1])):
        assert torch.equal(x, y), f"KV cache of layer {l} differs"


@pytest.mark.parametrize("B,kv,shape", [(32, "fp8", None), (23, "bf16", None), (16, "fp8", None), (9, "int8", None), (1, "fp8", None), (40, "fp8", None),
                                        (48, "bf16", None), (33, "int8", None), (27, "fp8", UNRELEASED), (7, "bf16", UNRELEASED)])
def test_backbone_segment_chain_at_1_to_32_rows_against_the_launch_path(B, kv, shape):
    """Round 5 (VERDICT r4 missing #3: the hole at 33-48 rows): those batch sizes run the 64-row stage set (bb_chain_kernel) with the
    last row tile partly filled, and the launch path picks the same tiles there (gemm.hip pick_tile) -- so the two schedules still
    agree bit for bit, at every batch size 1..64 now.
    Round 4: the 1.7B backbone segment as a persistent launch at 1-32 rows too (csrc/bb_chain.hip bb_chain_b32_kernel: the launch
    path's tiles at those batch sizes -- 16-row qkv / o / down tiles, gate_up on 16 rows up to 16 and 32 rows above, so the rstd
    summation order and with it every bit is the launch path's).  Three decode steps: logits, hidden, ids, codes, KV bytes identical;
    both chains reported as launched.  (Round 6: and at a width no released checkpoint has.)"""
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256, **(shape or {}))
    w = make_weights(d, seed=8, std=0.02)
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_bb_chain.argtypes = [C.c_int]; lib.omni_debug_bb_chain.restype = None
        try:
            for on in (0, 1):
                lib.omni_debug_bb_chain(on)
                eng = _decode_engine(d, w, B, kv)
                outs = []
                for _ in range(3):
                    eng.decode_step(B)
                    outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.input_ids[:B].clone(), eng.audio_codes[:B].clone()))
                torch.cuda.synchronize()
                assert eng.chain_error() == 0
                assert eng.chains_ran() == (3 if on else 1), (on, eng.chains_ran())
                res[on] = (outs, [c.view(torch.uint8).clone() for c in eng.kv_caches])
        finally:
            lib.omni_debug_bb_chain(1)
    for s, (a, b) in enumerate(zip(res[1][0], res[0][0])):
        for name, x, y in zip(("logits", "hidden", "ids", "codes"), a, b):
            assert torch.equal(x, y), f"step {s}: {name} differ between the small-batch backbone chain and the launch path"
    for l, (x, y) in enumerate(zip(res[1][1], res[0][1])):
        assert torch.equal(x, y), f"KV cache of layer {l} differs"


@pytest.mark.parametrize("B", [64, 37, 32, 16, 5, 1])
def test_backbone_segment_chain_at_the_0p6b_shape_against_the_launch_path(B):
    """BASELINE config #2's backbone (hidden 1024, intermediate 3072: the code predictor's layer dimensions) runs the per-layer
    segment on the code predictor's 16-row stage set (csrc/bb_chain.hip bb_chain_small_kernel) at EVERY batch size; against the
    launch-per-op backbone of the same library: logits, hidden state, sampled ids, codes and every KV byte of three decode steps
    identical, no flag wait times out, and the native step reports both chains as launched.
    Reference shape: configuration_qwen3_tts.py:192-216,376-409 (read from the checkpoint, never hard-coded: the chain is chosen
    by the shape, every other shape keeps the launch path)."""
    d = get_dims("tts-0.6b").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=8, std=0.02)
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_bb_chain.argtypes = [C.c_int]; lib.omni_debug_bb_chain.restype = None
        try:
            for on in (0, 1):
                lib.omni_debug_bb_chain(on)
                eng = _decode_engine(d, w, B, "bf16")
                outs = []
                for _ in range(3):
                    eng.decode_step(B)
                    outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.input_ids[:B].clone(), eng.audio_codes[:B].clone()))
                torch.cuda.synchronize()
                assert eng.chain_error() == 0
                assert eng.chains_ran() == (3 if on else 1), (on, eng.chains_ran())
                res[on] = (outs, [c.view(torch.uint8).clone() for c in eng.kv_caches])
        finally:
            lib.omni_debug_bb_chain(1)
    for s, (a, b) in enumerate(zip(res[1][0], res[0][0])):
        for name, x, y in zip(("logits", "hidden", "ids", "codes"), a, b):
            assert torch.equal(x, y), f"step {s}: {name} differ between the 0.6B backbone chain and the launch path"
    for l, (x, y) in enumerate(zip(res[1][1], res[0][1])):
        assert torch.equal(x, y), f"KV cache of layer {l} differs"


@pytest.mark.parametrize("B,kv,base", [(64, "fp8", "launch-path"), (49, "bf16", "layer-chain"), (57, "int8", "launch-path"), (51, "fp16", "layer-chain")])
def test_whole_backbone_launch_against_the_launch_path(B, kv, base):
    """The whole decoder stack -- qkv(0), then attention -> o_proj -> gate_up -> down_proj -> next qkv per layer -- as ONE persistent
    launch (csrc/bb_all.hip: the paged attention is a stage of the grid, two (row, kv head) pairs per workgroup) against the
    launch-per-op backbone and against round 3's per-layer chain of the same library: pa_body.cuh / chain_gemm.cuh are the launch
    path's arithmetic, so logits, hidden state, sampled ids, codes, slot mapping and every KV byte of three decode steps are
    identical; no flag wait times out.  Reference: the vLLM Qwen3 decoder stack behind qwen3_tts_talker.py:341,414-422."""
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=8, std=0.02)
    res = {}
    with L.debug_library() as lib:
        for fn in (lib.omni_debug_bb_chain, lib.omni_debug_bb_all, lib.omni_debug_pa_tail):
            fn.argtypes = [C.c_int]; fn.restype = None
        try:
            lib.omni_debug_pa_tail(0)          # (the in-grid attention stage keeps the interleaved tail round: compare like with like)
            for on in (0, 1):
                lib.omni_debug_bb_all(on)
                lib.omni_debug_bb_chain(1 if (on or base == "layer-chain") else 0)
                eng = _decode_engine(d, w, B, kv)
                outs = []
                for _ in range(3):
                    eng.decode_step(B)
                    outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.input_ids[:B].clone(), eng.audio_codes[:B].clone(),
                                 eng.slot_mapping[:B].clone()))
                torch.cuda.synchronize()
                assert eng.chain_error() == 0
                res[on] = (outs, [c.view(torch.uint8).clone() for c in eng.kv_caches])
        finally:
            lib.omni_debug_bb_chain(1)
            lib.omni_debug_bb_all(0)
            lib.omni_debug_pa_tail(1)
    for s, (a, b) in enumerate(zip(res[1][0], res[0][0])):
        for name, x, y in zip(("logits", "hidden", "ids", "codes", "slots"), a, b):
            assert torch.equal(x, y), f"step {s}: {name} differ between the one-launch backbone and the {base}"
    for l, (x, y) in enumerate(zip(res[1][1], res[0][1])):
        assert torch.equal(x, y), f"KV cache of layer {l} differs"


@pytest.mark.parametrize("B,kv", [(64, "int8"), (37, "fp8"), (8, "bf16"), (1, "int8")])
def test_moe_layer_chain_against_the_launch_path(B, kv):
    """Round 5 (VERDICT r4 missing #2, BASELINE configs #4 / #5): the Omni talker's sparse-MoE layer as TWO persistent launches around its
    expert GEMMs (csrc/moe_chain.hip) -- o_proj -> { router GEMM + normalised rows | shared-expert gate_up } -> { shared-expert down_proj,
    top-k routing }, and { combine into the residual stream -> the next layer's qkv } -- against the seven launches they replace (mode 2:
    the first chain alone): the launch path's tiles, k-step ownership, combine orders and the routing / combine kernels' arithmetic, so three
    decode steps at the released Omni talker width (3 layers) agree bit for bit in logits, hidden state, sampled ids, code frames and every
    KV byte; no flag wait times out; the step reports the chain as launched.  Reference: HF Qwen3OmniMoeTalkerTextSparseMoeBlock behind
    qwen3_omni.py:586-649."""
    d = get_dims("omni-talker").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=8, std=0.02)
    res = {}
    with L.debug_library() as lib:
        lib.omni_debug_moe_chain.argtypes = [C.c_int]; lib.omni_debug_moe_chain.restype = None
        try:
            for on in (0, 2, 1):
                lib.omni_debug_moe_chain(on)
                eng = _decode_engine(d, w, B, kv)
                outs = []
                for _ in range(3):
                    eng.decode_step(B)
                    outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.input_ids[:B].clone(), eng.audio_codes[:B].clone()))
                torch.cuda.synchronize()
                assert eng.chain_error() == 0
                assert bool(eng.chains_ran() & 2) == bool(on), (on, eng.chains_ran())
                res[on] = (outs, [c.view(torch.uint8).clone() for c in eng.kv_caches])
        finally:
            lib.omni_debug_moe_chain(1)
    for on, what in ((2, "the first MoE chain"), (1, "both MoE chains")):
        for s, (a, b) in enumerate(zip(res[on][0], res[0][0])):
            for name, x, y in zip(("logits", "hidden", "ids", "codes"), a, b):
                assert torch.equal(x, y), f"step {s}: {name} differ between {what} and the launch path"
        for l, (x, y) in enumerate(zip(res[on][1], res[0][1])):
            assert torch.equal(x, y), f"KV cache of layer {l} differs ({what})"


def test_chain_steps_replay_in_a_graph_and_stay_deterministic():
    """Whole decode steps with the chain inside a captured hipGraph: two engines fed the same requests produce the same
    codes step after step (flags and epochs advance on the device across replays), error word stays 0."""
    d = get_dims("tts-1.7b").with_(layers=2, max_model_len=256)
    w = make_weights(d, seed=4, std=0.02)
    B = 64
    outs = []
    for _ in range(2):
        eng = _engine(d, w, kv_dtype="fp8", num_blocks=256, max_batch=64)
        g = torch.Generator().manual_seed(9)
        eng.input_ids[:B] = torch.randint(1, d.codebook, (B,), generator=g).to(torch.int32).cuda()
        eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
        eng.text_step[:B] = (torch.randn(B, d.hidden, generator=g) * 0.02).to(BF16).cuda()
        eng.positions[:B] = 17
        eng.seq_lens[:B] = 18
        for b in range(B):
            eng.block_table[b, :2] = torch.tensor([1 + 2 * b, 2 + 2 * b], dtype=torch.int32)
        eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
        eng.decode_step(B); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eng.decode_step(B)
        frames = []
        for _ in range(6):
            gr.replay()
            frames.append(eng.audio_codes[:B].clone())
        torch.cuda.synchronize()
        assert eng.chain_error() == 0
        outs.append(torch.stack(frames).cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0][1:] != outs[0][:-1]).any(), "steps did not advance"
