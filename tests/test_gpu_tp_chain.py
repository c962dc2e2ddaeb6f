"""Round 6: the backbone's persistent launches on TENSOR-PARALLEL ranks -- the one-shot all-reduce rides inside the o_proj / down_proj stages
(chain_gemm AR, bb_chain.hip) -- and on the HALF grid (128 workgroups play the 256 of the stage grid).  The launch-per-op tensor-parallel
step (omni_allreduce_resid launches, allreduce.hip) is the reference arithmetic: same partials, same rank-order sum, same slab order.

Reference: the tensor-parallel group of V/worker/gpu_ar_worker.py:69-75 (RowParallelLinear all-reduces of the vLLM Qwen3 decoder layer,
qwen3_tts_talker.py:341,414-422); stage_configs/qwen3_omni_moe.yaml:27 (tensor_parallel_size)."""
import os
import socket

import pytest
import torch

from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _state(eng, B, seed):
    """The same decode-step input on every engine built with the same seed: context 3..40 per row, KV bytes random (fp8: below 0x64)."""
    g = torch.Generator().manual_seed(seed)
    d = eng.d
    nblk = 3
    bt = torch.arange(1, 1 + B * nblk, dtype=torch.int32).view(B, nblk)
    eng.block_table[:B, :nblk] = bt.to(eng.block_table.device)
    pos = torch.randint(3, 40, (B,), generator=g, dtype=torch.int32)
    eng.positions[:B] = pos.cuda()
    eng.seq_lens[:B] = (pos + 1).cuda()
    eng.input_ids[:B] = torch.randint(0, d.codebook, (B,), generator=g, dtype=torch.int32).cuda()
    eng.last_hidden[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
    eng.text_step[:B] = torch.randn(B, d.hidden, generator=g).to(BF16).cuda()
    eng.steps[:B] = 1
    for c in eng.kv_caches:
        c.copy_(torch.randint(0, 100, c.shape, generator=g, dtype=torch.uint8))
    torch.cuda.synchronize()


def _run(eng, B, n_steps, mode, seed, sync=None):
    """n decode steps from the seeded state under one chain mode (0 launch per op, 1 chains, 2 half grid); what the steps left behind."""
    eng.chain_error(reset=True)
    eng.set_chains(mode)
    _state(eng, B, seed)
    if sync is not None:
        sync()
    outs = []
    for _ in range(n_steps):
        eng.decode_step(B)
        torch.cuda.synchronize()
        outs.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.audio_codes[:B].clone(), eng.input_ids[:B].clone()))
    return outs, eng.chains_ran(), eng.chain_error()


def _same(a, b, what):
    for s, (x, y) in enumerate(zip(a, b)):
        for name, u, v in zip(("logits", "last_hidden", "codes", "ids"), x, y):
            assert torch.equal(u, v), f"{what}: step {s} {name} differ ({(u != v).sum().item()} elements)"


@pytest.mark.parametrize("B", [64, 40])
def test_half_grid_backbone_chain_equals_the_full_grid_and_the_launch_path(B):
    """Single rank: 128 workgroups playing the 256 of the stage grid leave the same bits as the 256-workgroup launches and as launch per op."""
    from ht_vllm_omni_amd.engine import TalkerEngine
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=21, std=0.02)
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=B * 3 + 2, block_size=16, max_batch=64)
    eng.set_sampling(greedy=1, cp_greedy=1)
    off, ran0, _ = _run(eng, B, 3, 0, seed=5)
    full, ran1, e1 = _run(eng, B, 3, 1, seed=5)
    half, ran2, e2 = _run(eng, B, 3, 2, seed=5)
    assert ran0 == 0 and (ran1 & 2) and (ran2 & 2) and not (ran2 & 1), (ran0, ran1, ran2)      # half grid: backbone chain only
    assert e1 == 0 and e2 == 0
    _same(full, off, "full grid vs launch path")
    _same(half, off, "half grid vs launch path")


def test_two_concurrent_row_ranges_keep_the_backbone_chain_on_half_grids():
    """128 rows as two concurrent 64-row ranges (parallel graph branches, bench.py --sub-batches 2): each range's backbone runs its persistent
    launches on a half grid -- the two are co-resident -- and leaves the bits of launch per op; no flag wait times out."""
    from ht_vllm_omni_amd.engine import TalkerEngine
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=21, std=0.02)
    B = 128
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=B * 3 + 2, block_size=16, max_batch=B, n_sub=2)
    eng.set_sampling(greedy=1, cp_greedy=1)
    off, ran0, _ = _run(eng, B, 3, 0, seed=8)
    g = None
    eng.chain_error(reset=True)
    eng.set_chains(2)                        # (opt-in: at 2 x 64 rows the half grids measure slower than launch per op, engine.py)
    _state(eng, B, 8)
    eng.decode_step(B)                       # warm-up, then the same three steps as ONE captured graph each (two branches)
    torch.cuda.synchronize()
    _state(eng, B, 8)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.decode_step(B)
    _state(eng, B, 8)
    half = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        half.append((eng.logits[:B].clone(), eng.last_hidden[:B].clone(), eng.audio_codes[:B].clone(), eng.input_ids[:B].clone()))
    assert ran0 == 0 and eng.chains_ran() == 2 and eng.chain_error() == 0
    _same(half, off, "two half-grid ranges vs launch path")


@pytest.mark.parametrize("B,shard", [(64, 1), (40, 1), (64, 2), (48, 4), (64, 8)])
def test_one_rank_group_runs_the_backbone_chain_with_the_all_reduce_stages(B, shard):
    """The tensor-parallel code path on a group of ONE rank (bench.py --tp-force): the all-reduce instantiation of the backbone launches
    (partial -> bf16 -> sum over one rank -> residual) leaves the bits of the launch-per-op step with its omni_allreduce_resid launches --
    full grid and half grid -- and the all-reduce's epoch word counts two calls per layer either way.  shard > 1: a model whose WHOLE
    backbone has the dimensions of one rank of the 1.7B shape split `shard` ways (heads and intermediate divided): the stage sets that
    ranks of 2, 4 and 8 run."""
    from ht_vllm_omni_amd.engine import TalkerEngine
    from ht_vllm_omni_amd.tp_comm import PeerAllReduce
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256)
    d = d.with_(q_heads=d.q_heads // shard, kv_heads=d.kv_heads // shard, inter=d.inter // shard)
    w = make_weights(d, seed=21, std=0.02)
    ar = PeerAllReduce(0, 1, 64, d.hidden)
    PeerAllReduce.link_local([ar])
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=B * 3 + 2, block_size=16, max_batch=64, tp_rank=0, tp_size=1, tp_force=True, peer_allreduce=ar)
    assert eng.tp_path and eng.fused_norm
    eng.set_sampling(greedy=1, cp_greedy=1)
    off, ran0, _ = _run(eng, B, 3, 0, seed=6)
    full, ran1, e1 = _run(eng, B, 3, 1, seed=6)
    half, ran2, e2 = _run(eng, B, 3, 2, seed=6)
    assert ran0 == 0 and (ran1 & 2) and (ran2 & 2), (ran0, ran1, ran2)
    assert e1 == 0 and e2 == 0 and ar.error() == 0
    _same(full, off, "all-reduce stages (full grid) vs all-reduce launches")
    _same(half, off, "all-reduce stages (half grid) vs all-reduce launches")
    ar.close()


def _tp2_proc(rank, world, port, q, B):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)                                            # both ranks share the test box's one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ht_vllm_omni_amd.engine import TalkerEngine
    from ht_vllm_omni_amd.tp_comm import PeerAllReduce
    d = get_dims("tts-1.7b").with_(layers=3, max_model_len=256)
    w = make_weights(d, seed=23, std=0.02)
    ar = PeerAllReduce(rank, world, 64, d.hidden).connect()
    eng = TalkerEngine(d, w, kv_dtype="fp8", num_blocks=B * 3 + 2, block_size=16, max_batch=64, tp_rank=rank, tp_size=world, peer_allreduce=ar)
    eng.set_sampling(greedy=1, cp_greedy=1)
    off, ran0, _ = _run(eng, B, 3, 0, seed=7, sync=dist.barrier)
    dist.barrier()
    half, ran2, e2 = _run(eng, B, 3, 2, seed=7, sync=dist.barrier)
    dist.barrier()
    ok_bits = all(torch.equal(u, v) for x, y in zip(off, half) for u, v in zip(x, y))
    # the start-up comparison a worker runs before it trusts the exchange (tp_comm.check_backbone_chain): passes on the half grid; on the
    # FULL grid two ranks of one GPU cannot be co-resident -- a peer wait runs out, every rank agrees to keep the all-reduce launches, the
    # error words are cleared and the next steps are right again
    from ht_vllm_omni_amd.tp_comm import check_backbone_chain
    chk_half = check_backbone_chain(eng, log=lambda m: None, chain_mode=2)
    chk_full = check_backbone_chain(eng, log=lambda m: None, chain_mode=1) if B == 40 else False
    after, ran3, e3 = (_run(eng, B, 3, 0, seed=7, sync=dist.barrier) if B == 40 else (off, 0, 0))
    parts = (ok_bits, chk_half, not chk_full, ran3 == 0, e3 == 0, ar.error() == 0, all(torch.equal(u, v) for x, y in zip(off, after) for u, v in zip(x, y)))
    ok_bits = parts if not all(parts) else True      # (the tuple names the failing piece in the assertion message)
    import hashlib
    digests = [hashlib.sha1(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest() for t in half[-1]]      # (plain data through the queue)
    q.put((rank, ran0, ran2, e2, ar.error(), ok_bits, digests))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("B", [64, 40, 33])
def test_two_rank_processes_all_reduce_inside_the_backbone_launches(B):
    """BASELINE config #4's split (tensor parallel over 2 ranks: 8 q / 4 kv heads and intermediate 3072 per rank), one PROCESS per rank, both
    on this box's one GPU: hipIpc-mapped partial buffers and tile flags, the backbone's persistent launches on the half grid so that the
    two ranks' launches are co-resident (on a node every rank owns a GPU and runs the full grid: same stages, same exchange).  Every rank:
    the steps with the all-reduce inside the launches leave the bits of the launch-per-op steps with omni_allreduce_resid launches; no
    peer wait timed out; the ranks agree."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_tp2_proc, args=(r, 2, port, q, B)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ran0, ran2, e2, arerr, ok_bits, _ in res:
        assert ran0 == 0 and (ran2 & 2), f"rank {rank}: chains_ran {ran0} / {ran2}"
        assert e2 == 0 and arerr == 0, f"rank {rank}: a wait timed out (chain word {e2}, all-reduce word {arerr})"
        assert ok_bits is True, (f"rank {rank}: (steps bit-identical, start-up comparison passes on the half grid, fails on two full grids, fall-back ran "
                                 f"launch per op, chain word clean, all-reduce word clean, steps after the fall-back bit-identical) = {ok_bits}")
    assert res[0][6] == res[1][6], "the ranks disagree"
