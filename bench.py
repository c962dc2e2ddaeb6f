#!/usr/bin/env python3
"""bench.py -- talker AR decode throughput on MI355X (BASELINE.json metric).

Workload W3 (BASELINE config #3): Qwen3-TTS-1.7B-shaped talker, random bf16 weights
(seed 1234), fp8-e4m3fn KV (unit scales), B = 64 concurrent requests, prompt lengths U{32..160}
(seed 7), KV block 16, sampling T=0.9 / top-k 50 / rep-penalty 1.05 / seed 42 with EOS masked, the
whole decode step (code predictor + 28-layer backbone + lm_head + sampler) captured as ONE hipGraph.
A "step" = one decode step of the batch = 64 speech tokens (64 codec frames of 16 codes).

  python bench.py --gpus N --steps K --warmup W
N > 1 without WORLD_SIZE in the environment: this process (which has made no GPU call) starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits with
the child's code; under torchrun (WORLD_SIZE set) each rank runs the tensor-parallel step over RCCL (strong scaling).
The timed window is centred on the BASELINE context (mean ctx 352) whatever --steps is: untimed decode steps advance
the batch there first, over KV the steps themselves wrote.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_engine(args, rank, world):
    from ht_vllm_omni_amd.config import get_dims
    from ht_vllm_omni_amd.engine import TalkerEngine
    from ht_vllm_omni_amd.weights import make_weights
    d = get_dims(args.model)
    t0 = time.time()
    w = make_weights(d, seed=1234, std=0.02, device="cuda" if args.device_weights else "cpu")
    log(f"[rank {rank}] weights generated in {time.time() - t0:.1f}s")
    ar = None
    if (world > 1 or args.tp_force) and args.parallel == "tp" and args.allreduce == "oneshot":
        ar = setup_peer_allreduce(d, args, rank, world)
    args.allreduce_used = "none" if (world == 1 and not args.tp_force) or args.parallel == "dp" else ("oneshot-xgmi (peer-mapped, fused with residual add)" if ar else "rccl")
    eng = TalkerEngine(d, w, kv_dtype=args.kv, num_blocks=args.num_blocks, block_size=16, max_batch=args.batch,
                       device=f"cuda:{torch.cuda.current_device()}", tp_rank=0 if args.parallel == "dp" else rank,
                       tp_size=1 if args.parallel == "dp" else world, allow_eos=False,
                       n_sub=args.sub_batches, tp_force=args.tp_force, peer_allreduce=ar, prefill_gemm=args.prefill_gemm)
    args.tp_backbone_chain = False
    if ar is not None and world > 1:      # the all-reduce inside the backbone's persistent launches: four scratch steps both ways on every rank first
        from ht_vllm_omni_amd.tp_comm import check_backbone_chain
        args.tp_backbone_chain = check_backbone_chain(eng, log=log)
    return d, w, eng


def setup_peer_allreduce(d, args, rank, world):
    """The checked set-up lives with the worker surface (ht_vllm_omni_amd/tp_comm.py); the bench uses the same function."""
    from ht_vllm_omni_amd.tp_comm import setup_peer_allreduce as setup
    return setup(d.hidden, args.batch, rank, world, log=log)


def setup_requests(d, eng, args):
    """Prompt lengths, block tables for the whole run (all blocks allocated up front: no host work in
    the timed loop), prefill through the native path; returns (lens, prefill_ms)."""
    from ht_vllm_omni_amd.sched import BlockPool
    B, bs = args.batch, 16
    rng = np.random.default_rng(7)
    lens = rng.integers(32, 161, size=B).tolist()
    total_steps = args.warmup + args.steps + args.ttfa_steps + 4 + args.ctx_extra + max(args.target_ctx, 0)   # incl. the untimed advance
    pool = BlockPool(args.num_blocks, bs)
    g = torch.Generator().manual_seed(7)
    bt = torch.zeros(eng.max_batch, eng.bt_stride, dtype=torch.int32)
    for r, n in enumerate(lens):
        pool.allocate(f"r{r}", n + total_steps)
        ids = pool.block_ids(f"r{r}")
        bt[r, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    eng.block_table.copy_(bt)
    x = (torch.randn(sum(lens), d.hidden, generator=g) * 0.05).to(torch.bfloat16).cuda()
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32)
    req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32)
    slots = torch.tensor([int(bt[int(req[t]), int(pos[t]) // bs]) * bs + int(pos[t]) % bs for t in range(len(pos))])
    pos, req, slots = pos.cuda(), req.cuda(), slots.cuda()
    last = torch.tensor(np.cumsum(lens) - 1, dtype=torch.int32).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hid = eng.prefill(x, pos, req, slots)
    from ht_vllm_omni_amd import ops
    hl = ops.embed(last, hid)        # row gather on the library's kernel (torch's first fancy-index launch loads a code object: 44 ms
                                     # of process start-up that is no request latency, scripts/diag_b1_prefill.py)
    logits = eng.compute_logits(hl)
    from ht_vllm_omni_amd import ops
    s = eng.sampling
    eng.seen.zero_()
    eng.seen[:B, d.codec_pad_id] = 1
    eng.steps.zero_()
    ids = ops.sample(logits, greedy=bool(s["greedy"]), temperature=s["temperature"], top_k=s["top_k"], rep_penalty=s["rep_penalty"],
                     seen=eng.seen[:B], seed=s["seed"], steps=eng.steps[:B], inc_steps=True)
    torch.cuda.synchronize()
    prefill_ms = (time.perf_counter() - t0) * 1e3
    eng.input_ids[:B] = ids
    eng.last_hidden[:B] = hl
    # --ctx-extra: long-context points -- the decode starts `ctx_extra` positions later, over cache blocks that hold
    # whatever the allocator left there (timing only; attention cost does not depend on the values)
    eng.positions[:B] = torch.tensor(lens, dtype=torch.int32).cuda() + args.ctx_extra
    eng.seq_lens[:B] = torch.tensor(lens, dtype=torch.int32).cuda() + 1 + args.ctx_extra
    eng.text_step[:B] = (torch.randn(B, d.hidden, generator=g) * 0.05).to(torch.bfloat16).cuda()   # tts_pad_embed rows
    return lens, prefill_ms


def cpu_baseline(d, w, args, lens):
    """The oracle (CPU restatement of the reference's algorithm, kind 'port') timed on the host cores on a bounded
    sample of the same workload: `cpu_batch` (default: all 64) of the 64 requests, fp8 KV pre-filled with random bytes at the
    W3 mean context, `cpu_steps` full decode steps (no prefill; fp32 weight views built before the clock starts).
    A reported baseline, not the optimisation target."""
    from oracle import talker_oracle as O
    B, bs = args.cpu_batch, 16
    n_steps = args.cpu_steps
    # the thread count the oracle runs FASTEST at on this class of host (many small torch ops: one thread per core of a 128-core box is
    # ~10 x slower than 16 threads, profiles/r05_oracle_threads.txt): the fair CPU figure, and `cores` says how many were used
    threads0 = torch.get_num_threads()
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    lens = lens[:B]
    ctx = [n + getattr(args, "ctx_before_timed", args.warmup) + args.steps // 2 for n in lens]     # the timed window's mean context
    nblk = sum((c + n_steps + bs) // bs for c in ctx) + 1
    orc = O.TalkerOracle(d, w, kv_dtype="fp8", num_blocks=nblk, block_size=bs)
    g = torch.Generator().manual_seed(11)
    for kv in orc.kv:
        kv.data.view(torch.uint8).copy_(torch.randint(0, 120, kv.data.shape, generator=g, dtype=torch.uint8))
    bts, nxt = [], 1
    for c in ctx:
        need = (c + n_steps + bs) // bs
        bts.append(list(range(nxt, nxt + need)))
        nxt += need
    pad = (torch.randn(d.hidden, generator=g) * 0.05).to(torch.bfloat16)
    states = [O.OracleState(seq_len=c, last_id=int(torch.randint(1, d.codebook, (1,), generator=g)),
                            last_hidden=torch.randn(d.hidden, generator=g).to(torch.bfloat16), tts_pad=pad, prompt_len=lens[i],
                            out_ids=[1]) for i, c in enumerate(ctx)]
    samp = dict(temperature=0.9, top_k=50, rep_penalty=1.05, seed=42)
    cpk = dict(do_sample=True, temperature=0.9, top_k=50, seed=42)
    for k, t in w.items():                      # exact fp32 views of the bf16 weights (sgemm path), outside the clock
        if t.ndim >= 2:
            O._f32(t)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        orc.decode_step(states, bts, greedy=False, sampling=samp, cp_kw=cpk)
    dt = time.perf_counter() - t0
    O.clear_weight_cache()
    used = torch.get_num_threads()
    torch.set_num_threads(threads0)
    return {"value": B * n_steps / dt, "unit": "speech-tokens/s", "cores": used, "cores_host": os.cpu_count(), "kind": "port",
            "sample": f"{n_steps} full decode step(s) of the CPU oracle (re-prefill code predictor as in the reference) on "
                      + (f"all {B} requests of the batch" if B == args.batch else f"the first {B} of the {args.batch} requests")
                      + f" (SURVEY 8d's 64 x 32 steps would be ~{32 * dt / n_steps / 60:.1f} min of CPU: bounded to {n_steps} step(s), ~{dt:.0f} s), "
                      f"fp8 KV, mean ctx {int(np.mean(ctx))}, no prefill; {dt / n_steps * 1e3:.0f} ms/step"}


def engine_loop(d, w, args, lens, async_on, n_steps=200):
    """The number a drop-in stage delivers: wall time per decode step THROUGH the reference's own loop -- scheduler.schedule ->
    worker.execute_model -> worker.sample_tokens (-> AsyncStepOutput.get_output) -> scheduler.update_from_output -- on the same
    workload, same context window as the bare hipGraph replay of `value`.  async_on = the stage config's `async_scheduling: true`
    (stage_configs/qwen3_tts.yaml:16): step t + 1 is dispatched before step t's outputs are read."""
    from ht_vllm_omni_amd.payloads import SamplingParams, encode_tensor
    from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, TalkerStageEngine
    from ht_vllm_omni_amd.worker import MI355XARWorker, make_config
    B, bs = args.batch, 16
    sp = SamplingParams(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=42, max_tokens=100000, stop_token_ids=())
    cfg = make_config(d, kv_cache_dtype=args.kv, block_size=bs, max_num_seqs=B, num_gpu_blocks_override=args.num_blocks, weights=w,
                      default_sampling_params=sp, async_scheduling=async_on)
    wk = MI355XARWorker(cfg, local_rank=torch.cuda.current_device(), rank=0)
    wk.init_device(); wk.load_model(); wk.initialize_from_config(None)
    wk.engine.set_sampling(cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
    wk.compile_or_warm_up_model()
    sched = MI355XARScheduler(num_blocks=args.num_blocks, block_size=bs, max_num_seqs=B, max_num_batched_tokens=8192,
                              max_model_len=d.max_model_len, need_send_cache=False, async_scheduling=async_on)
    core = TalkerStageEngine(wk, sched)
    g = torch.Generator().manual_seed(7)
    for r, n in enumerate(lens):
        info = {"talker_prompt_embeds": encode_tensor((torch.randn(n, d.hidden, generator=g) * 0.05).to(torch.bfloat16)),
                "tts_pad_embed": encode_tensor((torch.randn(d.hidden, generator=g) * 0.05).to(torch.bfloat16))}
        core.add_request(Request(request_id=f"s{r}", num_prompt_tokens=n, prompt_token_ids=[d.codec_pad_id] * n, sampling_params=sp,
                                 additional_information=info, ignore_eos=True))
    advance = max(8, int(round(args.target_ctx - float(np.mean(lens)) - n_steps / 2.0))) if args.target_ctx > 0 else 16
    for _ in range(advance):
        core.step()
    torch.cuda.synchronize()
    run = wk.model_runner
    t_sched = t_upd = t_disp = t_get = 0.0
    n_tok = 0
    t0 = time.perf_counter()
    for _ in range(n_steps):              # TalkerStageEngine.step, unrolled for the per-phase clocks
        a = time.perf_counter()
        so = sched.schedule()
        b = time.perf_counter()
        core.inflight.append((so, core._dispatch(so)))
        c = time.perf_counter()
        t_sched += b - a
        t_disp += c - b
        if async_on and len(core.inflight) < core.max_inflight:
            continue
        so0, h = core.inflight.popleft()
        out = h.get_output() if hasattr(h, "get_output") else h
        e_ = time.perf_counter()
        outs = sched.update_from_output(so0, out)
        f = time.perf_counter()
        t_get += e_ - c
        t_upd += f - e_
        n_tok += sum(len(o.new_token_ids) for o in outs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx = float(wk.engine.seq_lens[:B].float().mean().item()) - n_steps / 2.0
    while core.inflight:
        so0, h = core.inflight.popleft()
        sched.update_from_output(so0, h.get_output() if hasattr(h, "get_output") else h)
    words = wk.engine.status.cpu().tolist()
    res = {"ms_per_step": dt / n_steps * 1e3, "tokens_per_s": B * n_steps / dt, "steps": n_steps, "mean_ctx": ctx,
           "async_scheduling": bool(async_on),
           "runner_ms": (t_disp + t_get) / n_steps * 1e3, "runner_dispatch_ms": t_disp / n_steps * 1e3,
           "runner_get_output_ms": t_get / n_steps * 1e3,
           "scheduler_ms": (t_sched + t_upd) / n_steps * 1e3, "tokens_seen": n_tok, "chains_ran": int(words[2]),
           "chain_fallbacks": int(run.chain_fallbacks), "replays": int(run.cudagraph_stats["replays"])}
    wk.shutdown()
    del core, sched, wk
    torch.cuda.empty_cache()
    return res


def copy_probe_gbs():
    """Achievable HBM bandwidth: device-to-device copy of 1 GiB (read + write bytes / time)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    # the ONE JSON line owns stdout: native libraries (the RCCL version banner, hipBLASLt notices) write to fd 1 too, so
    # fd 1 is pointed at stderr for the whole run and the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="tts-1.7b")
    ap.add_argument("--kv", default="fp8")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--num-blocks", type=int, default=8192)
    ap.add_argument("--ttfa-steps", type=int, default=16, help="initial_chunk_size at full load (chunk_size_utils.py:12-33)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch threads of the CPU-oracle leg (its fastest setting on a 128-core host)")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-diagnostics", action="store_true", help="profiler runs: skip the untimed per-family / backbone-only replays "
                    "behind the timed region (their launches would be counted into the profile)")
    ap.add_argument("--no-engine-loop", action="store_true", help="skip the untimed-for-`value` leg that runs the step through the "
                    "scheduler / worker loop (reported under \"engine_loop\")")
    ap.add_argument("--device-weights", action="store_true", help="profiler runs only: draw weights on the GPU (no H2D copy)")
    ap.add_argument("--sub-batches", type=int, default=1, help="independent row ranges run as parallel graph branches")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--parallel", choices=("tp", "dp"), default="tp",
                    help="N > 1: tp = one batch of 64 sharded tensor-parallel over the ranks (BASELINE config, strong scaling); "
                         "dp = independent replicas, 64 requests per GPU, no data-path collective (weak scaling)")
    ap.add_argument("--tp-force", action="store_true", help="diagnostics: run the tensor-parallel code path (process group, "
                    "separate norms, all-reduces inside the graph) on a 1-rank group")
    ap.add_argument("--no-replica-leg", action="store_true", help="N > 1 with --parallel tp: skip the second, untimed-for-`value` "
                    "leg that runs the same step as independent replicas (reported under \"replicas\")")
    ap.add_argument("--allreduce", choices=("oneshot", "rccl"), default="oneshot",
                    help="N > 1 tensor parallel: peer-mapped one-shot all-reduce inside the native step (self-checked against RCCL "
                         "at start-up, falls back to it) or RCCL all-reduces between the phase calls")
    ap.add_argument("--prefill-gemm", choices=("tile", "blas"), default="tile", help="GEMMs of the TTFA prefill: omni_gemm_tile on "
                    "the decode step's fragment-major weights (stored once) | hipBLASLt on row-major copies")
    ap.add_argument("--ctx-extra", type=int, default=0, help="long-context points: start decoding this many positions later")
    ap.add_argument("--target-ctx", type=int, default=352, help="mean context of the timed window (W3: 96 + 256); untimed decode "
                    "steps advance the batch until the window is centred there (0: time from wherever warm-up ends)")
    ap.add_argument("--master-port", type=int, default=0, help="N > 1 self-launch: rendezvous port (0: pick a free one)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # One process per GPU: start the ranks as CHILDREN of this process, which has not touched the GPU (no exec after
        # HIP initialisation on this pool), relay rank 0's JSON line, fail when any rank fails.
        import socket
        import subprocess
        port = args.master_port
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # (OMNI_BENCH_LAUNCHER: tests put a stub module in the launcher's place to check the relay of a successful child's line)
        cmd = [sys.executable, "-m", os.environ.get("OMNI_BENCH_LAUNCHER", "torch.distributed.run"), "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        log("self-launch:", " ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            log(f"bench.py: the {args.gpus}-rank launch failed (exit {r.returncode}, {len(lines)} JSON line(s))")
            sys.exit(r.returncode or 1)
        os.write(json_fd, (lines[-1] + "\n").encode())
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number as {args.gpus} GPUs")
        sys.exit(2)
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.tp_force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                 # --tp-force without a launcher: a 1-rank group on a free port
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))

    d, w, eng = build_engine(args, rank, world)
    for env, sym in (("OMNI_EXTRA_TRIVIAL", "omni_debug_extra_trivial"),     # diagnostics: price of a trivial launch inside the real step
                     ("OMNI_INT8_MAX_G", "omni_debug_int8_max_g")):        # diagnostics: q heads per workgroup of the int8-KV decode attention
        if os.environ.get(env):
            import ctypes
            if not hasattr(eng.lib, sym):
                raise SystemExit(f"{env} is a hook of the diagnostics build: run with OMNI_TALKER_DEBUG=1 (libomni_talker_debug.so), "
                                 f"the product library does not export {sym}")
            getattr(eng.lib, sym).argtypes = [ctypes.c_int]
            getattr(eng.lib, sym)(int(os.environ[env]))
    B = args.batch
    if args.greedy:
        eng.set_sampling(greedy=1, cp_greedy=1)
    else:
        eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9,
                         cp_top_k=50)
    setup_requests(d, eng, args)                 # untimed: hipBLASLt handle / heuristics, code objects
    lens, prefill_ms = setup_requests(d, eng, args)
    log(f"[rank {rank}] prefill of {sum(lens)} prompt tokens: {prefill_ms:.1f} ms ({'omni_gemm_tile (hand-written MFMA)' if args.prefill_gemm == 'tile' else 'hipBLASLt'} GEMMs + native norm/rope/attention)")

    # ---- capture the whole decode step as one hipGraph
    graph, use_graph = None, not args.no_graph
    eng.decode_step(B)                       # eager warm-up step (also the first TTFA step)
    torch.cuda.synchronize()
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eng.decode_step(B)
            torch.cuda.synchronize()
        except Exception as e:               # noqa: BLE001
            log(f"[rank {rank}] hipGraph capture failed ({e!r}); falling back to eager launches")
            graph = None
    run = (graph.replay if graph is not None else (lambda: eng.decode_step(B)))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- TTFA: prefill + IC decode steps (IC = 16 at full load), vocoder / HTTP excluded
    steps_done = 2 if graph is not None else 1
    n_ic = max(args.ttfa_steps - steps_done, 2)      # at IC = 2 (B = 1) both steps are already behind us: time two more
    sync()
    t0 = time.perf_counter()
    for _ in range(n_ic):
        run()
    sync()
    ic_ms = (time.perf_counter() - t0) * 1e3 / n_ic * args.ttfa_steps
    ttfa_ms = prefill_ms + ic_ms

    # ---- centre the timed window on the BASELINE context: untimed decode steps (they write the KV the timed steps read)
    advance = 0
    if args.target_ctx > 0:
        cur = float(eng.seq_lens[:B].float().mean().item())         # context incl. the token of the next step
        advance = max(0, int(round(args.target_ctx - (cur + args.warmup + (args.steps - 1) / 2.0))))
        for _ in range(advance):
            run()
    for _ in range(args.warmup):
        run()
    sync()
    ctx0 = eng.seq_lens[:B].cpu().numpy().astype(np.int64)          # context incl. the token of the next step
    args.ctx_before_timed = int(round(float(np.mean(ctx0)) - float(np.mean(lens))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ev_ms = e0.elapsed_time(e1) / args.steps
    chains_ran = int(eng.status[2])          # status word 2 of the LAST REPLAYED step (what it launched: bit 0 the code-predictor chain, bit 1
                                             # the backbone chain) -- not the last CAPTURED one
    # a timed-out flag wait (peer all-reduce, persistent chains) makes the remaining steps wrong AND faster: void the run
    # on every rank when any rank saw one
    dev_err = int(eng.chain_error() != 0) + 2 * int(eng.ar is not None and eng.ar.error() != 0)
    rank_words = [dev_err]
    if dist is not None:
        allw = torch.zeros(world, dtype=torch.int32, device="cuda")
        allw[rank] = dev_err
        dist.all_reduce(allw, op=dist.ReduceOp.MAX)
        rank_words = [int(x) for x in allw.tolist()]      # every rank's error words in the line (0 = clean)
        dev_err = max(rank_words)
    if dev_err:
        log(f"[rank {rank}] in-kernel hand-off timed out during the timed region (code {dev_err}: 1 = chain flags, 2 = peer all-reduce): result void")
        sys.exit(3)

    # ---- tensor-parallel ranks: what ONE all-reduce of the step costs on this node, and which one ran (the first
    # 8-GPU run must say where its time went).  After the timed region, every rank in lock-step: 2 x layers launches of the step's
    # own all-reduce -- the one-shot kernel over the peer-mapped buffers (fused residual add + slabs), or RCCL on the [B, H] message
    ar_diag = None
    if dist is not None and args.parallel == "tp":
        try:
            n_ar = 2 * d.layers
            sync()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # every rank agrees that its buffers exist BEFORE the first lock-step collective: a rank-local failure (an allocation, a bad
            # error word) must not leave the other ranks waiting in a collective until the RCCL time-out with the measured line lost
            ok, r_io, slabs, msg = 1, None, None, None
            try:
                if eng.ar is not None:
                    r_io = torch.zeros(eng.ar.rows16, d.hidden, dtype=torch.bfloat16, device="cuda")
                    slabs = torch.zeros(d.hidden // 16, 64, dtype=torch.float32, device="cuda")
                    ok = int(eng.ar.error() == 0)
                else:
                    msg = torch.zeros(B, d.hidden, dtype=torch.bfloat16, device="cuda")
            except Exception as e:   # noqa: BLE001
                log(f"[rank {rank}] all-reduce diagnostic: set-up failed on this rank: {e!r}")
                ok = 0
            okt = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) == 0:
                raise RuntimeError("a rank could not set the diagnostic up: skipped on every rank")
            if eng.ar is not None:
                for it in range(4):
                    eng.ar.all_reduce(it & 1, r_io=r_io, accumulate=True, partials=slabs, M=B)
                a0.record()
                for it in range(n_ar * 4):
                    eng.ar.all_reduce(it & 1, r_io=r_io, accumulate=True, partials=slabs, M=B)
                a1.record()
                kind = "oneshot-xgmi (peer-mapped, fused with residual add)"
            else:
                for it in range(4):
                    dist.all_reduce(msg)
                a0.record()
                for it in range(n_ar * 4):
                    dist.all_reduce(msg)
                a1.record()
                kind = "rccl (fall-back: all-reduces between the phase calls)"
            sync()
            us = a0.elapsed_time(a1) * 1e3 / (n_ar * 4)
            tus = torch.tensor([us], dtype=torch.float64, device="cuda")
            allus = [torch.zeros_like(tus) for _ in range(world)]
            dist.all_gather(allus, tus)
            ar_diag = {"kind": kind, "launches_per_step": n_ar, "us_per_launch_by_rank": [round(float(x.item()), 2) for x in allus],
                       "us_per_step": round(max(float(x.item()) for x in allus) * n_ar, 1), "message_bytes": B * d.hidden * 2,
                       "error_word": int(eng.ar.error()) if eng.ar is not None else 0,
                       "note": "back-to-back launches of the step's own all-reduce after the timed region, all ranks in lock-step"}
            log(f"[rank {rank}] all-reduce: {kind}: {us:.2f} us per launch x {n_ar} per step = {us * n_ar / 1e3:.3f} ms; error word "
                f"{ar_diag['error_word']}; rank error words {rank_words}")
        except Exception as e:   # noqa: BLE001
            log(f"[rank {rank}] all-reduce diagnostic failed: {e!r}")

    # ---- diagnostics (after the timed region): the backbone half of the step alone, as its own graph
    bb_ms = None
    if graph is not None and world == 1 and args.sub_batches == 1 and not args.tp_force and not args.no_diagnostics:
        try:
            g2 = torch.cuda.CUDAGraph()
            eng.backbone_step(B)
            torch.cuda.synchronize()
            with torch.cuda.graph(g2):
                eng.backbone_step(B)
            g2.replay()
            torch.cuda.synchronize()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            for _ in range(32):
                g2.replay()
            b1.record()
            torch.cuda.synchronize()
            bb_ms = b0.elapsed_time(b1) / 32
        except Exception as e:   # noqa: BLE001
            log(f"backbone-only diagnostic failed: {e!r}")

    # ---- per-family times of the step, measured live in this run (after the timed region): each part of the step captured
    # as its own hipGraph and replayed alone (omni_talker_step_part).  A part run alone reads whatever the buffers hold: its time is
    # the step's, its outputs are not -- this engine's request state is void from here on (the replica leg builds its own).
    families = None
    if graph is not None and world == 1 and args.sub_batches == 1 and not args.tp_force and not args.no_diagnostics:
        try:
            split = bool(chains_ran & 2)
            plan = [("code_predictor_phase_ms", 1)] + ([("backbone_attention_ms", 2), ("backbone_chain_ms", 4)] if split else [("backbone_stack_ms", 6)]) \
                + [("lm_head_sampler_ms", 8)]
            families = {}
            for name, parts in plan:
                eng.step_part(B, parts)
                torch.cuda.synchronize()
                gp = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp):
                    eng.step_part(B, parts)
                gp.replay()
                torch.cuda.synchronize()
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(16):
                    gp.replay()
                f1.record()
                torch.cuda.synchronize()
                families[name] = f0.elapsed_time(f1) / 16
            families["sum_ms"] = sum(v for k, v in families.items())
            families["note"] = ("each family replayed alone as its own hipGraph after the timed region, at the final context: "
                                "code_predictor_phase = pair pass (positions 0/1) + group-1 head and sampler + the chain launch "
                                "(passes 2..15) + input assembly; backbone_attention = the 28 paged-attention launches; backbone_chain "
                                "= first qkv + the 28 four-stage launches; the parts of a step overlap by a launch boundary each, so "
                                "sum_ms reads a few percent above event_ms_per_step")
        except Exception as e:   # noqa: BLE001
            log(f"family breakdown failed: {e!r}")
            families = None

    # ---- roofline: algorithmic bytes of one step (each counted once) / measured step time
    mean_ctx = ctx0 + (args.steps - 1) / 2.0
    by = eng.step_bytes(mean_ctx)
    achieved = by["total"] / (ev_ms * 1e-3) / 1e9
    end_ctx = ctx0 + args.steps
    by_end = eng.step_bytes(end_ctx) if bb_ms is not None else None
    out = {
        "metric": "speech-tokens/sec", "value": B * args.steps / dt * (world if args.parallel == "dp" else 1),
        "unit": "speech-tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.parallel == "dp" else "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"W3: Qwen3-TTS-1.7B-shaped talker decode, random bf16 weights, {args.kv} KV, B={B}, "
                               f"prompts U{{32..160}} seed 7, KV block 16, T=0.9/top-k 50/rep 1.05 sampling"
                               if args.model == "tts-1.7b" else f"{args.model} {args.kv} B={B}",
                   "model": args.model, "kv_cache": args.kv, "batch": B, "mean_ctx": float(np.mean(mean_ctx)),
                   "parallelism": f"{args.parallel}{world}", "allreduce": getattr(args, "allreduce_used", "none"),
                   "prefill_gemm": "omni_gemm_tile (hand-written MFMA)" if args.prefill_gemm == "tile" else "hipBLASLt",
                   "hipgraph": graph is not None, "sub_batches": args.sub_batches,
                   # what the captured step actually launched (omni_talker_chains_ran), not what was asked for
                   "chains_ran": chains_ran,
                   "decode_launches": "; ".join(
                       (["code-predictor passes 2..15 (layer stacks, heads, samplers) = 1 persistent launch"] if chains_ran & 1
                        else ["code predictor: one launch per op"]) +
                       (["backbone = 1 attention + 1 four-stage persistent launch (o_proj, gate_up, down_proj, next qkv) per layer"]
                        if chains_ran & 2 else ["backbone: one launch per op"]) +
                       # round 6 (single rank, both chains): the step's small launches ride in the chains
                       (["final norm + lm_head = last stage of the last backbone launch; input assembly + layer 0's qkv = tail of the predictor launch"]
                        if chains_ran == 3 and world == 1 and not args.tp_force and args.sub_batches == 1 else [])),
                   "multi_gpu": ("one engine per GPU (replicas, no data-path collective) is the throughput mode of this stage; tensor "
                                 "parallelism divides the backbone's bytes but not the replicated code predictor: latency mode "
                                 "(DESIGN 5)"),
                   **({"rccl_ranks": world, "rank_error_words": rank_words,
                       "allreduce_in_backbone_launches": bool(chains_ran & 2) and getattr(args, "allreduce_used", "").startswith("oneshot")}
                      if world > 1 or args.tp_force else {}),
                   "sampling": "greedy" if args.greedy else "T=0.9,top_k=50,rep=1.05,seed=42",
                   "target_ctx": args.target_ctx, "untimed_advance_steps": advance,
                   **({"ctx_extra": args.ctx_extra} if args.ctx_extra else {})},
        "p50_ttfa_ms": ttfa_ms, "ttfa": {"prefill_ms": prefill_ms, "ic_steps": args.ttfa_steps, "ic_ms": ic_ms},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "launch": "one hipGraph replay = one decode step (per TP rank)", "event_ms_per_step": ev_ms,
                     "bytes_per_step": by},
    }
    if args.tp_force:
        out["config"]["parallelism"] += " (tensor-parallel code path forced on one rank)"
    if ar_diag is not None:
        out["allreduce"] = ar_diag
    if families is not None:
        out["roofline"]["families"] = families
    if bb_ms is not None:
        bb_bytes = by_end["weights_backbone"] + by_end["lm_head"] + by_end["kv_read"] + by_end["kv_write"]
        out["roofline"]["breakdown"] = {
            "note": f"diagnostic, outside the timed region: the backbone half ({d.layers} layers + lm_head + sampler) replayed alone "
                    "at the final context; code predictor + input assembly = step - backbone",
            "backbone_ms": bb_ms, "code_predictor_ms": ev_ms - bb_ms, "backbone_ctx": float(np.mean(end_ctx)),
            "backbone_gbs": bb_bytes / (bb_ms * 1e-3) / 1e9, "backbone_frac": bb_bytes / (bb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # ---- second leg (N > 1, tensor-parallel runs only; never part of `value`): the same step as independent replicas --
    # one full engine per GPU, 64 requests each, no data-path collective -- so the line carries both axes of SURVEY 8e.
    # Every rank makes the same collective calls whether or not its local part succeeded.
    if dist is not None and args.parallel == "tp" and not args.no_replica_leg:
        ok, dt_r, run2 = 1, 0.0, None
        try:
            graph = None
            del eng
            torch.cuda.empty_cache()
            from ht_vllm_omni_amd.engine import TalkerEngine
            eng2 = TalkerEngine(d, w, kv_dtype=args.kv, num_blocks=args.num_blocks, block_size=16, max_batch=args.batch,
                                device=f"cuda:{torch.cuda.current_device()}", tp_rank=0, tp_size=1, allow_eos=False)
            eng2.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
            setup_requests(d, eng2, args)
            eng2.decode_step(B)
            torch.cuda.synchronize()
            g3 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g3):
                eng2.decode_step(B)
            for _ in range(args.warmup + args.ttfa_steps + advance):      # same context as the tensor-parallel leg's timed region
                g3.replay()
            torch.cuda.synchronize()
            run2 = g3.replay
        except Exception as e:   # noqa: BLE001
            log(f"[rank {rank}] replica leg set-up failed: {e!r}")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run2()
            sync()
            tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            dt_r = float(tr.item())
            out["replicas"] = {"value": world * B * args.steps / dt_r, "unit": "speech-tokens/s", "ms_per_step": dt_r / args.steps * 1e3,
                               "scaling": "weak", "parallelism": f"dp{world}",
                               "note": "same step, one full engine and 64 requests per GPU, no data-path collective; "
                                       "second leg of this run, not part of `value`"}
        eng = None
    if rank == 0:
        # HBM bytes per step from the PMC passes (rocprofv3 cannot ride along with a timed run: separate
        # --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, summary committed under profiles/)
        # `traffic` stays null unless a committed PMC summary of this same command exists AT THIS CONTEXT (within 5 %);
        # a figure measured at another context is reported beside it, labelled, never as roofline.traffic
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_traffic.json")), reverse=True):
            if not (world == 1 and args.model == "tts-1.7b"):
                break
            try:
                tj = json.load(open(tpath))
                tctx = float(tj.get("mean_ctx", 105.0))
                rel = os.path.relpath(tpath, ROOT)
                if abs(tctx - float(np.mean(mean_ctx))) <= 0.05 * float(np.mean(mean_ctx)):
                    out["roofline"]["traffic"] = tj["traffic_bytes_per_step"]
                    out["roofline"]["traffic_ctx"] = tctx
                    out["roofline"]["traffic_note"] = f"bytes/step, PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes of this command ({rel})"
                    out["roofline"]["traffic_source"] = {"file": rel, "mean_ctx": tctx, "measured_by": "the builder, separate rocprofv3 --pmc passes of this command; not re-measured in this run"}
                else:
                    out["roofline"]["traffic_other_ctx"] = {"bytes_per_step": tj["traffic_bytes_per_step"], "mean_ctx": tctx, "source": rel}
                break
            except Exception as e:   # noqa: BLE001
                log(f"traffic file unreadable: {e!r}")
        try:
            out["roofline"]["copy_probe_gbs"] = copy_probe_gbs()
        except Exception as e:   # noqa: BLE001
            log(f"copy probe failed: {e!r}")
        if world == 1 and args.sub_batches == 1 and not args.tp_force and not args.no_diagnostics and not args.no_engine_loop:
            # outside `value`: the same step through the reference's scheduler / worker loop, async scheduling on (the shipped
            # stage config) and off
            try:
                eng = None
                graph = None
                torch.cuda.empty_cache()
                el = engine_loop(d, w, args, lens, True)
                el["bare_replay_ms"] = ev_ms
                el["over_bare_replay_ms"] = el["ms_per_step"] - ev_ms
                el["synchronous"] = engine_loop(d, w, args, lens, False)
                el["note"] = ("scheduler.schedule -> worker.execute_model -> sample_tokens -> AsyncStepOutput.get_output -> "
                              "update_from_output around every step (host copy of ids + status, code frames and hidden states, "
                              "per-request bookkeeping and stop checks included); runner_ms / scheduler_ms are HOST time, which "
                              "async scheduling hides under the GPU's next step")
                out["engine_loop"] = el
            except Exception as e:   # noqa: BLE001
                log(f"engine-loop leg failed: {e!r}")
                out["engine_loop"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                eng = None
                torch.cuda.empty_cache()
                out["cpu_baseline"] = cpu_baseline(d, w, args, lens)
            except Exception as e:   # noqa: BLE001
                log(f"cpu baseline failed: {e!r}")
                out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
