"""The reference's shipped TTS deployment puts the talker stage AND the code2wav stage on device "0", two processes
(V/model_executor/stage_configs/qwen3_tts.yaml:7,39).  The persistent chains are 256-workgroup launches whose workgroups wait for
each other: next to another process's kernels they must still complete -- their waits are bounded by wall clock, and the stage
degrades to the launch path if one ever runs out (tests/test_gpu_fallback.py) -- and must produce the same bits.

A child process (started with the `spawn` method: a fresh interpreter, never a re-exec of a process that holds the GPU) decodes
50-frame windows through the full-size Code2Wav decoder in a loop on the same GPU while this process replays captured decode
steps of the 1.7B-shaped talker with both chains on."""
import time

import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _code2wav_loop(ready, stop, count, unstable=None):
    import torch as T
    from ht_vllm_omni_amd.code2wav import Code2WavDecoder
    from tests.codec_util import FULL_CODEC, make_codec_state
    T.cuda.set_device(0)
    dec = Code2WavDecoder(FULL_CODEC, make_codec_state(FULL_CODEC, 0, device="cuda"))
    codes = T.randint(0, 2048, (1, 16, 50), device="cuda")
    first = dec(codes).clone()
    T.cuda.synchronize()
    ready.set()
    n = bad = 0
    while not stop.is_set():
        for _ in range(4):
            out = dec(codes)
        bad += int(not T.equal(out, first))          # the vocoder's own bits beside the talker's kernels
        T.cuda.synchronize()
        n += 4
    count.value = n
    if unstable is not None:
        unstable.value = bad


def _replay(d, w, B, steps, kv="fp8"):
    from ht_vllm_omni_amd.engine import TalkerEngine
    eng = TalkerEngine(d, w, kv_dtype=kv, num_blocks=64 * 40 + 2, max_batch=64)
    g = torch.Generator().manual_seed(9)
    eng.input_ids[:B] = torch.randint(1, d.codebook, (B,), generator=g).to(torch.int32).cuda()
    eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
    eng.text_step[:B] = (torch.randn(B, d.hidden, generator=g) * 0.02).to(BF16).cuda()
    eng.positions[:B] = 17
    eng.seq_lens[:B] = 18
    nb = (18 + steps + 40) // 16 + 1
    for b in range(B):
        eng.block_table[b, :nb] = torch.arange(1 + nb * b, 1 + nb * (b + 1), dtype=torch.int32)
    eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
    eng.decode_step(B)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eng.decode_step(B)
    hist = torch.empty(steps, B, d.num_code_groups, dtype=torch.int64, device="cuda")
    stat = torch.zeros(steps, 4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        gr.replay()
        hist[s] = eng.audio_codes[:B]
        stat[s] = eng.status
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return hist.cpu(), stat.cpu(), dt / steps * 1e3, eng.chain_error(), eng.chains_ran()


# every KV storage type has its own K / V conversion code in the attention kernel, and 16 q / 4 kv heads run the four-heads-per-
# workgroup instantiation (round 6: the backbone chain reads its qkv width from the arguments, so this head ratio runs both chains too)
# ...; "moe": the Omni talker's sparse-MoE backbone (router, route, expert and combine kernels; 16 q / 2 kv heads) at 3 layers
@pytest.mark.parametrize("kv,kv_heads,chains", [("fp8", 8, 3), ("bf16", 8, 3), ("int8", 8, 3), ("fp16", 8, 3), ("fp8", 4, 3), ("int8", "moe", None)])
@pytest.mark.timeout(900)
def test_chained_steps_beside_a_code2wav_process_on_the_same_gpu(kv, kv_heads, chains):
    import json
    import os
    import torch.multiprocessing as mp
    if kv_heads == "moe":
        d = get_dims("omni-talker").with_(layers=3, max_model_len=1024)
    else:
        d = get_dims("tts-1.7b").with_(layers=4, max_model_len=1024, kv_heads=kv_heads)
    w = make_weights(d, seed=4, std=0.02)
    # 200 steps per case; the fp8 headline case and the MoE case (its two persistent launches per layer share the flag region with the
    # predictor's chains) run 500: the long runs are what exercises flag-epoch drift over tens of thousands of stages
    B, steps = 64, (500 if (kv, kv_heads) in (("fp8", 8), ("int8", "moe")) else 200)
    solo, st0, ms0, err0, ran0 = _replay(d, w, B, steps, kv)
    chains = ran0 if chains is None else chains
    assert err0 == 0 and ran0 == chains and int(st0[:, :2].abs().sum()) == 0
    ctx = mp.get_context("spawn")
    ready, stop, count = ctx.Event(), ctx.Event(), ctx.Value("i", 0)
    child = ctx.Process(target=_code2wav_loop, args=(ready, stop, count))
    child.start()
    try:
        assert ready.wait(300), "the code2wav process did not come up"
        both, st1, ms1, err1, ran1 = _replay(d, w, B, steps, kv)
    finally:
        stop.set()
        child.join(120)
    assert child.exitcode == 0
    assert count.value > 0, "the code2wav process decoded nothing while the talker ran: no co-location was tested"
    assert err1 == 0 and ran1 == chains, f"chain error word {err1:#x} beside the code2wav process (chains ran: {ran1})"
    assert int(st1[:, :2].abs().sum()) == 0, "a step reported a status word"
    assert torch.equal(solo, both), "codes differ between the solo run and the run beside the code2wav process"
    rep = {"steps": steps, "batch": B, "layers": d.layers, "solo_ms_per_step": ms0, "colocated_ms_per_step": ms1, "slowdown": ms1 / ms0,
           "code2wav_windows_decoded_meanwhile": int(count.value), "chain_error_word": err1}
    if (kv, kv_heads) == ("fp8", 8):
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(rep, open("gpurun_out/colocation.json", "w"), indent=1)
    print(rep)


def _prefill_once(d, w, T_tok, reqs):
    """The all-tokens prefill (omni_gemm_tile, the MFMA prefill attention, norm / RoPE / KV-write kernels) of `reqs` equal prompts."""
    from ht_vllm_omni_amd.engine import TalkerEngine
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=reqs * (T_tok // 16 + 2) + 2, max_batch=64)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(reqs * T_tok, d.hidden, generator=g) * 0.5).to(BF16).cuda()
    pos = torch.arange(T_tok, dtype=torch.int32).repeat(reqs).cuda()
    req = torch.arange(reqs, dtype=torch.int32).repeat_interleave(T_tok).cuda()
    nb = T_tok // 16 + 1
    for b in range(reqs):
        eng.block_table[b, :nb] = torch.arange(1 + nb * b, 1 + nb * (b + 1), dtype=torch.int32)
    slots = (eng.block_table[req.long(), (pos // 16).long()].long() * 16 + (pos % 16).long())
    outs = []
    for _ in range(6):
        outs.append(eng.prefill(x, pos, req, slots, use_blas=True).clone())
    torch.cuda.synchronize()
    kv = [c.clone() for c in eng.kv_caches]
    return outs, kv


@pytest.mark.timeout(900)
def test_prefill_and_vocoder_bits_beside_each_other():
    """The same question for the rest of the path: the all-tokens prefill repeated beside the looping Code2Wav process gives
    the solo run's bits (hidden states and every KV byte), and the vocoder's windows keep THEIR bits meanwhile."""
    import torch.multiprocessing as mp
    d = get_dims("tts-1.7b").with_(layers=4, max_model_len=1024)
    w = make_weights(d, seed=4, std=0.02)
    solo, kv0 = _prefill_once(d, w, 96, 32)
    assert all(torch.equal(solo[0], o) for o in solo[1:]), "the prefill is not run-to-run stable even alone"
    ctx = mp.get_context("spawn")
    ready, stop, count, unstable = ctx.Event(), ctx.Event(), ctx.Value("i", 0), ctx.Value("i", -1)
    child = ctx.Process(target=_code2wav_loop, args=(ready, stop, count, unstable))
    child.start()
    try:
        assert ready.wait(300), "the code2wav process did not come up"
        both, kv1 = _prefill_once(d, w, 96, 32)
        _replay(d, w, 64, 200)                       # keep the talker busy for a while longer: the vocoder checks itself meanwhile
    finally:
        stop.set()
        child.join(120)
    assert child.exitcode == 0 and count.value > 0
    assert all(torch.equal(solo[0], o) for o in both), "prefill hidden states differ beside the code2wav process"
    assert all(torch.equal(a, b) for a, b in zip(kv0, kv1)), "prefill KV bytes differ beside the code2wav process"
    assert unstable.value == 0, f"{unstable.value} code2wav windows differed from the first one while the talker ran"


@pytest.mark.timeout(900)
def test_steps_and_prefill_beside_a_bare_mfma_loop(tmp_path):
    """The harshest neighbour found in round 4: another process that does nothing but issue MFMAs on every SIMD
    (scripts/probes/aggressor.hip).  Beside it one operand selector of the packed fp32 instructions is wrong in 5 % of its executions
    (profiles/r04_pkfma_probe.txt) -- 10 000 x the rate beside the vocoder -- so a kernel of this library that is sensitive to the
    neighbour's matrix instructions in ANY way shows up within a few steps.  Decode steps (both chains), the all-tokens prefill with
    every KV byte, and the vocoder's windows must keep the solo run's bits."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "probes", "aggressor.hip")
    exe = str(tmp_path / "aggressor")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, src], check=True, capture_output=True, timeout=600)
    d = get_dims("tts-1.7b").with_(layers=4, max_model_len=1024)
    w = make_weights(d, seed=4, std=0.02)
    solo, st0, _, err0, ran0 = _replay(d, w, 64, 300)
    pre0, kv0 = _prefill_once(d, w, 96, 32)
    assert err0 == 0 and ran0 == 3
    from ht_vllm_omni_amd.code2wav import Code2WavDecoder
    from tests.codec_util import FULL_CODEC, make_codec_state
    dec = Code2WavDecoder(FULL_CODEC, make_codec_state(FULL_CODEC, 0, device="cuda"))
    wcodes = torch.randint(0, 2048, (1, 16, 50), device="cuda")
    wav0 = dec(wcodes).clone()
    child = subprocess.Popen([exe, "mfma", "60"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        time.sleep(3.0)                                   # (a separate executable: never a re-exec of this process)
        assert child.poll() is None, "the MFMA neighbour exited early: " + (child.stdout.read() if child.stdout else "")
        both, st1, _, err1, ran1 = _replay(d, w, 64, 300)
        pre1, kv1 = _prefill_once(d, w, 96, 32)
        wav_bad = sum(int(not torch.equal(dec(wcodes), wav0)) for _ in range(20))
        torch.cuda.synchronize()
        still_there = child.poll() is None
    finally:
        child.kill()
        child.wait(30)
    assert still_there, "the MFMA neighbour was gone before the talker finished: nothing was tested"
    assert err1 == 0 and ran1 == 3 and int(st1[:, :2].abs().sum()) == 0
    assert torch.equal(solo, both), "codes differ beside the MFMA loop"
    assert all(torch.equal(pre0[0], o) for o in pre1), "prefill hidden states differ beside the MFMA loop"
    assert all(torch.equal(a, b) for a, b in zip(kv0, kv1)), "prefill KV bytes differ beside the MFMA loop"
    assert wav_bad == 0, f"{wav_bad} of 20 code2wav windows differ beside the MFMA loop"
