"""Tensor-parallel communication of the decode step: peer-mapped one-shot all-reduce (csrc/allreduce.hip) set up over
torch.distributed.

The reference gets its tensor-parallel group from vLLM (``init_worker_distributed_environment`` at
V/worker/gpu_ar_worker.py:69-75) and all-reduces inside ``RowParallelLinear`` over NCCL/RCCL.  Here every rank of the group
allocates two partial buffers ([rows16, H] bf16, fragment-major: one for the o_proj, one for the down_proj all-reduce) and
one control block (arrival flags, epoch, error word) in fine-grained device memory, publishes their hipIpc handles with
``all_gather_object`` and maps the peers' -- after which the all-reduces are plain kernel launches of the native decode step
(captured into its hipGraph).  RCCL stays the transport for prefill-sized messages and for the one-time self-check.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

# control block: uint32 flags[8] | uint32 epoch, ticket | int32 error (64-byte aligned pieces) | uint32 tile_flags[8][256] (ABI v5: the
# arrival flags of the all-reduce inside the backbone's persistent launches, csrc/bb_chain.hip -- [source rank][tile])
TILE_OFF, TILE_BYTES = 256, 8 * 256 * 4
CTL_BYTES = TILE_OFF + TILE_BYTES
FLAGS_OFF, EPOCH_OFF, ERROR_OFF = 0, 64, 128


class PeerAllReduce:
    def __init__(self, rank: int, world: int, rows: int, hidden: int):
        if not 1 <= world <= 8:
            raise ValueError("peer all-reduce: world size 1..8 (one xGMI hop to every peer)")
        self.lib = L.load()
        self.rank, self.world, self.hidden = rank, world, hidden
        self.rows16 = (rows + 15) // 16 * 16
        self.nbytes = self.rows16 * hidden * 2
        self._own: list[int] = []
        self._opened: list[int] = []
        self.handles = []
        for nb in (self.nbytes, self.nbytes, CTL_BYTES):
            p, h = C.c_void_p(), (C.c_char * 64)()
            L.check(self.lib.omni_ar_alloc(nb, C.byref(p), h), "omni_ar_alloc")
            self._own.append(p.value)
            self.handles.append(bytes(h))
        self.data = [[None] * world, [None] * world]        # [buffer][rank] device pointers valid in this process
        self.ctl = [None] * world
        self.data[0][rank], self.data[1][rank], self.ctl[rank] = self._own
        self.peers = None

    # ---- wiring
    def connect(self, group=None) -> "PeerAllReduce":
        """Exchange the hipIpc handles over the (already initialised) process group and map every peer."""
        import torch.distributed as dist
        gathered = [None] * self.world
        dist.all_gather_object(gathered, self.handles, group=group)
        self.map_peers(gathered)
        dist.barrier(group=group)
        return self

    def map_peers(self, gathered: list) -> "PeerAllReduce":
        """Map every peer's buffers from the gathered hipIpc handles (local work only: no collective inside)."""
        for r, hs in enumerate(gathered):
            if r == self.rank:
                continue
            ptrs = []
            for h in hs:
                p = C.c_void_p()
                L.check(self.lib.omni_ar_open(C.create_string_buffer(h, 64), C.byref(p)), "omni_ar_open")
                self._opened.append(p.value)
                ptrs.append(p.value)
            self.data[0][r], self.data[1][r], self.ctl[r] = ptrs
        return self._finish()

    @staticmethod
    def link_local(members: list["PeerAllReduce"]) -> None:
        """All ranks inside ONE process (tests on a single GPU): peers are plain device pointers.  The ranks' steps run one after the other
        there, so the exchange inside a persistent launch (which waits for its peers within the launch) is left unconfigured: launch path."""
        for m in members:
            for o in members:
                m.data[0][o.rank], m.data[1][o.rank], m.ctl[o.rank] = o._own
            m._finish(in_chain=len(members) == 1)

    def _finish(self, in_chain: bool = True) -> "PeerAllReduce":
        self.peers = []
        for b in (0, 1):
            s = L.ArPeers()
            s.world, s.rank = self.world, self.rank
            for r in range(self.world):
                s.data[r] = self.data[b][r]
                s.flags[r] = self.ctl[r] + FLAGS_OFF
                s.tile_flags[r] = self.ctl[r] + TILE_OFF if in_chain else None
            s.epoch = self.ctl[self.rank] + EPOCH_OFF
            s.error = self.ctl[self.rank] + ERROR_OFF
            self.peers.append(s)
        return self

    # ---- views / calls
    def buffer(self, which: int) -> torch.Tensor:
        """This rank's partial buffer as a [rows16, H] bf16 tensor (fragment-major bytes)."""
        return _device_tensor(self.data[which][self.rank], self.rows16 * self.hidden, torch.bfloat16).view(self.rows16, self.hidden)

    def error(self) -> int:
        return int(_device_tensor(self.ctl[self.rank] + ERROR_OFF, 1, torch.int32).item())

    def clear_error(self) -> None:
        """Clear this rank's sticky error word (after the group agreed to stop using what timed out; every rank clears its own)."""
        _device_tensor(self.ctl[self.rank] + ERROR_OFF, 1, torch.int32).zero_()
        torch.cuda.synchronize()

    def all_reduce(self, which: int, *, r_io=None, accumulate=True, partials=None, out=None, M: int) -> None:
        """Launch on the current stream: sum of data[which] over the ranks (+ residual add / slabs / row-major copy)."""
        L.check(self.lib.omni_allreduce_resid(C.byref(self.peers[which]), L.ptr(r_io), int(accumulate), L.ptr(partials),
                                              64 if partials is None else partials.shape[-1], L.ptr(out), M, self.hidden,
                                              L.current_stream()), "omni_allreduce_resid")

    def close(self) -> None:
        for p in self._opened:
            self.lib.omni_ar_close(p)
        for p in self._own:
            self.lib.omni_ar_free(p)
        self._opened, self._own = [], []


def setup_peer_allreduce(hidden: int, rows: int, rank: int, world: int, group=None, log=print):
    """Peer-mapped one-shot all-reduce for the tensor-parallel step, with a self-check against RCCL on random partials
    before it is trusted: any failure (allocation, IPC mapping, a peer that does not arrive, a wrong sum on ANY rank) makes
    EVERY rank return None -- the engine then runs RCCL all-reduces between the phase calls.  Every rank makes the same
    sequence of collective calls whatever happens locally (local failures are recorded and agreed on by an all-reduce(MIN)).
    Called by MI355XARWorker.initialize_from_config (the reference's group: V/worker/gpu_ar_worker.py:69-75) and by bench.py."""
    import torch.distributed as dist
    from .engine import frag_shuffle

    dev = "cuda" if torch.cuda.is_available() else "cpu"      # cpu: the gloo tests of the fall-back agreement

    def agree(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    def fallback(ar):
        if rank == 0:
            log("peer all-reduce unavailable: falling back to RCCL all-reduces between the phase calls")
        try:
            if ar is not None:
                ar.close()
        except Exception:   # noqa: BLE001
            pass
        return None

    ar, ok = None, True
    try:
        ar = PeerAllReduce(rank, world, rows, hidden)
    except Exception as e:   # noqa: BLE001
        log(f"[rank {rank}] peer all-reduce buffers unavailable: {e!r}")
        ok = False
    if not agree(ok):
        return fallback(ar)
    gathered = [None] * world
    dist.all_gather_object(gathered, ar.handles, group=group)
    try:
        ar.map_peers(gathered)
    except Exception as e:   # noqa: BLE001
        log(f"[rank {rank}] hipIpc mapping of the peers failed: {e!r}")
        ok = False
    if not agree(ok):
        return fallback(ar)
    M = rows
    g = torch.Generator().manual_seed(100 + rank)
    for it in range(4):
        part = torch.zeros(ar.rows16, hidden, dtype=torch.bfloat16)
        part[:M] = torch.randn(M, hidden, generator=g).to(torch.bfloat16)
        out = torch.zeros(M, hidden, dtype=torch.bfloat16, device="cuda")
        ref = part[:M].float().cuda()
        try:
            ar.buffer(it & 1).copy_(frag_shuffle(part).cuda())
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            log(f"[rank {rank}] self-check staging failed: {e!r}")
            ok = False
        dist.barrier(group=group)
        try:
            if ok:
                ar.all_reduce(it & 1, out=out, M=M)
        except Exception as e:   # noqa: BLE001
            log(f"[rank {rank}] one-shot all-reduce launch failed: {e!r}")
            ok = False
        dist.all_reduce(ref, group=group)
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item()
        if ok and (ar.error() != 0 or not err <= 2.0 ** -7 * max(ref.abs().max().item(), 1.0)):
            log(f"[rank {rank}] one-shot all-reduce self-check failed (error word {ar.error()}, max diff {err})")
            ok = False
    if not agree(ok):
        return fallback(ar)
    return ar


def check_backbone_chain(engine, group=None, log=print, chain_mode: int = 1, n_steps: int = 4) -> bool:
    """Before a tensor-parallel group trusts the all-reduce INSIDE its backbone's persistent launches (csrc/bb_chain.hip, chain_gemm AR):
    `n_steps` scratch decode steps both ways on every rank -- all-reduce launches between launch-per-op GEMMs, then the persistent launches -- from
    the same state; the logits and hidden rows must agree bit for bit on every rank and no peer wait may have timed out.  Any rank's failure
    (agreed on by an all-reduce(MIN)) leaves every rank on the all-reduce launches (``set_chains(3)``); like setup_peer_allreduce's self-check
    against RCCL this runs once at start-up, before any request, so the KV slot the scratch step writes belongs to nobody.  Returns whether
    the persistent launches are in use.  chain_mode = the ``set_chains`` mode under test and left on when it passes (1: the full grid of a
    rank that owns its GPU; 2: the half grid of ranks that share one).  Called by MI355XARWorker._build_engine and bench.py."""
    import torch.distributed as dist
    ar = getattr(engine, "ar", None)
    if ar is None or engine.tp_size <= 1 or not getattr(engine, "persistent_chains", False):
        return False
    B = min(engine.max_batch, 64)
    if B < 33:
        return False                                # below the backbone chain's row range: launch per op anyway
    names = ("input_ids", "positions", "seq_lens", "last_hidden", "steps", "seen", "text_step")
    keep = {n: getattr(engine, n).clone() for n in names if hasattr(engine, n)}
    # consecutive steps read the KV slots the earlier ones wrote: every scratch row needs a block of its own (rows sharing block 0 of a fresh
    # block table would race for its slots and the two runs would differ for that reason alone); too few blocks: one step (it reads no history)
    bt_keep = engine.block_table[:B, :1].clone()
    if engine.num_blocks > B and n_steps <= engine.block_size:
        engine.block_table[:B, 0] = torch.arange(1, B + 1, dtype=engine.block_table.dtype, device=engine.block_table.device)
    else:
        n_steps = 1
    g = torch.Generator().manual_seed(4321)        # the same scratch rows on every rank
    hid = torch.randn(B, engine.d.hidden, generator=g).to(torch.bfloat16)
    outs, ran = [], 0
    ok = True
    try:
        for mode in (0, chain_mode):                # 0: launch per op throughout (the predictor's chains do not touch this step's logits)
            for n, v in keep.items():
                getattr(engine, n).copy_(v)
            engine.last_hidden[:B] = hid.to(engine.last_hidden.device)
            engine.text_step[:B] = hid.flip(0).to(engine.text_step.device)
            engine.positions[:B] = 0
            engine.seq_lens[:B] = 1
            engine.set_chains(mode)
            torch.cuda.synchronize()
            dist.barrier(group=group)
            got = []
            for _ in range(n_steps):                # consecutive steps: both data buffers, epochs e + 1 .. e + 2 * layers * n_steps
                engine.decode_step(B)
                got += [engine.logits[:B].clone(), engine.last_hidden[:B].clone()]
            torch.cuda.synchronize()
            outs.append(got)
            ran = engine.chains_ran()
        ok = bool(ran & 2) and ar.error() == 0 and engine.chain_error() == 0 and all(torch.equal(a, b) for a, b in zip(*outs))
    except Exception as e:   # noqa: BLE001
        log(f"[rank {engine.tp_rank}] backbone chain start-up comparison failed to run: {e!r}")
        ok = False
    for n, v in keep.items():
        getattr(engine, n).copy_(v)
    engine.block_table[:B, :1] = bt_keep
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    agreed = int(flag.item()) == 1
    if not agreed:
        if engine.tp_rank == 0:
            log("all-reduce inside the backbone's persistent launches NOT in use (start-up comparison failed or the shape has no stage set): "
                "all-reduce launches between launch-per-op GEMMs")
        engine.chain_error(reset=True)
        engine.set_chains(3)
        torch.cuda.synchronize()
        dist.barrier(group=group)                  # nobody is still inside a launch that could raise the word again
        ar.clear_error()
    torch.cuda.synchronize()
    dist.barrier(group=group)
    return agreed


def _device_tensor(ptr: int, n: int, dtype) -> torch.Tensor:
    """A torch view of device memory this module allocated through the C-ABI (no ownership: PeerAllReduce frees it)."""
    itemsize = torch.empty((), dtype=dtype).element_size()
    typestr = {torch.bfloat16: "<u2", torch.int32: "<i4", torch.uint32: "<u4", torch.float32: "<f4", torch.uint8: "|u1"}[dtype]

    class _Holder:
        __cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
    t = torch.as_tensor(_Holder(), device="cuda")
    return t.view(dtype) if t.dtype != dtype and t.element_size() == itemsize else t
