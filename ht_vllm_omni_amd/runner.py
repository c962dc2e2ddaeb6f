"""MI355XARModelRunner: the reference's two-phase AR runner contract over the native engine.

Mirrors ``GPUARModelRunner`` / ``OmniGPUModelRunner`` for the talker stage:
  execute_model(scheduler_output) -> None      V/worker/gpu_ar_model_runner.py:93-400
  sample_tokens(grammar_output) -> OmniModelRunnerOutput                   ...:403-660
  _update_states (persistent batch + block table maintenance)   V/worker/gpu_model_runner.py:246-514
  _preprocess decode branch / talker_mtp staging                 ...:1084-1303  (batched on device here:
      the per-request Python loop with 4 D2D copies per request and the O(B^2) list.index scatter are gone;
      the text-step queue is one index_select, everything else lives in the native step)
  KV transfer on finish + ack via kv_extracted_req_ids           gpu_ar_model_runner.py:106-124,639

Persistent-batch invariant: rows [0, n_decode) of the engine's per-step buffers are the requests in decode
phase (prompt fully computed), in arrival order; rows are swapped when a later request finishes its prefill
first.  Freed / padded rows point at the null block (block 0), so graph buckets larger than the live batch
are inert.
"""
from __future__ import annotations

import logging
import os
import zlib
from collections import deque
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch

from .connectors import OmniKVTransferManager
from .payloads import (EMPTY_MODEL_RUNNER_OUTPUT, LogprobsLists, OmniModelRunnerOutput, OmniSchedulerOutput, SamplingParams,
                       decode_additional_information)

logger = logging.getLogger("ht_vllm_omni_amd.runner")
BF16 = torch.bfloat16


@dataclass
class RequestState:
    req_id: str
    prompt_embeds: torch.Tensor            # [prompt_len, H] bf16 (CPU; prefill-only, like talker_prompt_embeds)
    block_ids: list[int]
    sampling: SamplingParams
    num_computed: int = 0
    output_ids: list[int] = field(default_factory=list)
    tail: torch.Tensor | None = None       # tailing_text_hidden [n, H] (device resident, gpu_resident_buffer_keys)
    tail_pos: int = 0
    tts_pad: torch.Tensor | None = None    # [H] device
    info: dict[str, Any] = field(default_factory=dict)
    codes_hist: list[list[int]] = field(default_factory=list)   # audio codes of every decode step taken (recompute after preemption)
    recompute: int = 0                     # resumed after preemption: decode inputs appended to the prompt for the KV recompute
    n_prompt: int = 0                      # prompt rows proper (prompt_embeds may carry `recompute` rebuilt rows behind them)
    # M-RoPE ids of the prompt with differing rows and the resulting offset of every later position (vLLM CachedRequestState
    # .mrope_positions / .mrope_position_delta, set by _init_mrope_positions, V/worker/gpu_model_runner.py:121-180); None / 0 for
    # the talker's usual text-only prompt (three identical rows == plain RoPE)
    mrope_positions: torch.Tensor | None = None     # int64 [3, prompt_len]
    mrope_delta: int = 0
    # h[t] the request's next decode step consumes (host view of the previous step's copy): what a step is redone from after a
    # chain flag wait timed out (MI355XARModelRunner._redo_step)
    last_hidden_cpu: torch.Tensor | None = None
    # tokens sampled for the request so far, INCLUDING those of steps whose host copy has not been read yet (async scheduling:
    # output_ids trails by the steps in flight; the device rows -- ids, h, positions -- are always current)
    n_sampled: int = 0

    def rope_ids(self, s0: int, n: int):
        """[3, n] rotary ids of sequence indices [s0, s0 + n): the prompt's own ids, then index + delta."""
        idx = torch.arange(s0, s0 + n, dtype=torch.int64)
        out = (idx + self.mrope_delta).expand(3, -1).clone()
        if self.mrope_positions is not None:
            k = max(0, min(n, self.mrope_positions.shape[1] - s0))
            if k > 0:
                out[:, :k] = self.mrope_positions[:, s0:s0 + k]
        return out

    @property
    def prompt_len(self) -> int:
        return self.n_prompt or int(self.prompt_embeds.shape[0])

    @property
    def prefill_len(self) -> int:
        """Tokens that go through the prefill path: the prompt, plus -- for a request resumed after a recompute
        preemption -- the inputs of the decode steps it had already taken."""
        return self.prompt_len + self.recompute

    @property
    def in_decode(self) -> bool:
        return self.num_computed >= self.prefill_len


@dataclass
class _StepState:
    scheduler_output: OmniSchedulerOutput
    decode_rows: list[int]
    prefill_done: dict[int, torch.Tensor]      # row -> hidden [H] of the last prompt token
    prefill_sampled: torch.Tensor | None       # ids for prefill_done rows (device)
    prefill_spans: dict[int, tuple[int, int, torch.Tensor]]  # row -> (start, n, hidden[n,H])
    # the scheduled rows AS DISPATCHED, in row order: (row, request id, request state).  The bookkeeping of a step may run after
    # later steps were dispatched (async scheduling): by then rows have moved and finished requests have left `requests`, so it
    # reads this snapshot, never `self.rows`
    sched_rows: list = field(default_factory=list)
    n_rows: int = 0
    # requested log-probabilities (SamplingParams.logprobs; gpu_ar_model_runner.py:516-519,631-636): row -> device tensors
    # (token ids [k + 1], logprobs [k + 1], rank, NaNs in the row's logits), filled at dispatch from the step's own logits
    logprobs: dict = field(default_factory=dict)


@dataclass
class _HostRecord:
    """One step's outputs on the host, indexed by the step's OWN row numbers (those of _StepState.sched_rows)."""
    ids: list[int]
    status: list[int]
    hidden: torch.Tensor                        # [n_rows, H] bf16
    codes: torch.Tensor | None                  # [nd, Q] int64
    spans: dict[int, torch.Tensor]              # row -> hidden [n, H] of its prefill span
    dropped: set = field(default_factory=set)   # rows whose decode output of this step does not exist (redo: they had left)
    logprobs: dict = field(default_factory=dict)   # row -> (ids list, logprobs list, rank, nans)


class AsyncStepOutput:
    """What ``sample_tokens`` returns under async scheduling: the reference's ``AsyncGPUModelRunnerOutput``
    (gpu_ar_model_runner.py:641-660; vLLM ``AsyncModelRunnerOutput.get_output``).  The step's device outputs were snapshotted in
    stream order when it was dispatched; ``get_output()`` waits for that snapshot only -- not for the steps dispatched since --
    reads it, checks the step's status words, does the bookkeeping and returns the ``OmniModelRunnerOutput``."""

    def __init__(self, runner, stt: _StepState, pending, kv_extracted):
        self.runner, self.stt, self.pending, self.kv_extracted = runner, stt, pending, kv_extracted
        self.record: _HostRecord | None = None       # set early by a redo cascade (an earlier step of the queue was invalid)
        self._out = None

    def get_output(self) -> OmniModelRunnerOutput:
        if self._out is None:
            self._out = self.runner._finish_step(self)
        return self._out


class _Range:
    """Profiler range with the reference's phase names (gpu_ar_model_runner.py:138 preprocess, :293 forward, :314 postprocess,
    :454 sample, :514 bookkeep): a torch.profiler record_function AND a roctx range (torch.cuda.nvtx is roctx on ROCm, so the
    names show up in `rocprofv3 --marker-trace` too).  Off unless the worker's profile() -- or OMNI_PROFILE_RANGES=1 -- turned
    ranges on: two Python context managers per phase are not free at 4 ms per step."""
    enabled = os.environ.get("OMNI_PROFILE_RANGES") == "1"

    def __init__(self, name: str):
        self.name, self.rf, self.pushed = name, None, False

    def __enter__(self):
        if _Range.enabled:
            self.rf = torch.profiler.record_function(self.name)
            self.rf.__enter__()
            if torch.cuda.is_available():
                torch.cuda.nvtx.range_push(self.name)
                self.pushed = True
        return self

    def __exit__(self, *exc):
        if self.pushed:
            torch.cuda.nvtx.range_pop()
        if self.rf is not None:
            self.rf.__exit__(*exc)
        return False


def request_seed(req_id: str, sp: SamplingParams) -> int:
    """RNG key of a request: its own seed, else one derived from the request id (distinct noise per unseeded request)."""
    return int(sp.seed) & 0xFFFFFFFF if sp.seed is not None else (zlib.crc32(req_id.encode()) ^ 0x9E3779B9) & 0xFFFFFFFF


_ROW_BUFFERS = ("input_ids", "positions", "seq_lens", "block_table", "last_hidden", "seen", "steps",
                "row_greedy", "row_temperature", "row_top_k", "row_top_p", "row_rep_penalty", "row_seed")


class MI355XARModelRunner:
    def __init__(self, engine, *, kv_transfer: OmniKVTransferManager | None = None, use_graphs: bool = True,
                 engine_output_type: str = "latent", prompt_builder=None, async_scheduling: bool = False):
        self.engine = engine
        # async scheduling (stage_configs/qwen3_tts.yaml:16): sample_tokens returns an AsyncStepOutput at once, the step's outputs
        # are snapshotted in stream order and read in get_output() -- after the NEXT step has been enqueued.  Everything
        # execute_model needs of a step (ids, h, positions, counters) is device-resident and advanced on the device; the host
        # fields it reads (num_computed, n_sampled, text cursor) are advanced at DISPATCH, not at bookkeeping
        self.async_scheduling = bool(async_scheduling)
        self._inflight: deque = deque()                 # AsyncStepOutput handles not yet finished, oldest first
        self._stage_dev: list = []                      # device snapshot slots of the step's output record
        self._stage_i = 0
        self._copy_stream = None
        self._pin: dict[str, list] = {}                 # pinned host rings for the dispatch-side H2D copies (async, never blocking)
        self._pin_i: dict[str, int] = {}          # ring position per call-site key (_h2d)
        self.chain_fallbacks = 0
        # after a fall-back the chains come back once `_rearm_after` launch-path steps went by clean (doubling after every fall-back:
        # a GPU that is shared for good settles on the launch path, a passing neighbour costs one redo) -- ADVICE r4
        self._rearm_after, self._clean_steps, self._chains_parked = 512, 0, False
        self._num_live = -1                             # what engine.num_live holds (device word rewritten only when it changes)
        # Qwen3-Omni requests arrive with the thinker's outputs instead of ready talker_prompt_embeds: the builder
        # (prompt_builder_omni.OmniTalkerPromptBuilder = the reference's talker_preprocess_prefill) turns them into the
        # prompt embeddings + the queue of text steps on the device
        self.prompt_builder = prompt_builder
        self.d = engine.d
        self.max_num_seqs = engine.max_batch
        self.kv_caches = engine.kv_caches               # runner-owned list, one [2,nb,bs,h,d] tensor per layer
        self.kv_transfer_manager = kv_transfer or OmniKVTransferManager(None)
        self.requests: dict[str, RequestState] = {}
        self.model_intermediate_buffer: dict[str, dict[str, Any]] = {}      # gpu_model_runner.py:60
        self.gpu_resident_buffer_keys: set[str] = set()                      # model hook (gpu_model_runner.py:1337-1339)
        self.rows: list[str] = []                       # input_batch.req_ids (row order)
        self.preempted: dict[str, RequestState] = {}    # state of preempted requests (vLLM keeps self.requests across preemption)
        self.execute_model_state: _StepState | None = None
        self.kv_extracted_req_ids: list[str] | None = None
        self.use_graphs = use_graphs
        self.graphs: dict[int, Any] = {}
        self.engine_output_type = engine_output_type
        self.cudagraph_stats = {"replays": 0, "eager_steps": 0}
        self.default_sampling = SamplingParams()        # requests that carry none (worker: the stage's default_sampling_params)
        # per-step text rows as ONE device gather: every request's queue (tailing_text_hidden rows, then its tts_pad row)
        # is a segment of a table rebuilt only when the batch membership / row order changes; row r of a step reads
        # table[off[r] + min(pos[r], len[r])] (index len[r] = the pad row)
        self._tt = None
        self._tt_rows: list[str] = []
        self._tt_off = self._tt_len = self._tt_pos = None

    # ------------------------------------------------------------------ persistent batch
    def _reset_row(self, r: int) -> None:
        e = self.engine
        e.block_table[r].zero_()          # null block
        e.positions[r] = 0
        e.seq_lens[r] = 1
        e.input_ids[r] = 0
        e.steps[r] = 0
        e.seen[r].zero_()
        if getattr(e, "rope_delta", None) is not None:
            e.rope_delta[r] = 0
        # the previous occupant's sampling parameters must not outlive it: back to the stage default (greedy-safe values)
        e.set_row_sampling(r, greedy=True, temperature=1.0, top_k=0, top_p=1.0, rep_penalty=1.0, seed=0)

    def _permute_rows(self, perm: list[int]) -> None:
        n = len(perm)
        if perm == list(range(n)):
            return
        self._tt_flush()
        idx = self._h2d("perm", np.asarray(perm, np.int64), torch.int64)
        for name in _ROW_BUFFERS + ("rope_delta",):
            buf = getattr(self.engine, name, None)
            if buf is None:
                continue
            buf[:n] = buf[:n].index_select(0, idx)
        self.rows = [self.rows[i] for i in perm]

    def _tt_flush(self) -> None:
        """Write the queue positions back to the requests and drop the table (rows are about to change)."""
        if self._tt is not None:
            for r, rid in enumerate(self._tt_rows):
                st = self.requests.get(rid)
                if st is not None:
                    st.tail_pos = int(self._tt_pos[r])
        self._tt = None

    def text_queue_pos(self, req_id: str) -> int:
        """How many queued text rows the request has consumed (the live count is kept per row while the batch is stable)."""
        if self._tt is not None and req_id in self._tt_rows:
            return int(self._tt_pos[self._tt_rows.index(req_id)])
        return self.requests[req_id].tail_pos

    def _tt_build(self) -> None:
        H = self.d.hidden
        segs, off, ln, pos, o = [], [], [], [], 0
        for rid in self.rows:
            st = self.requests[rid]
            n = 0 if st.tail is None else int(st.tail.shape[0])
            if n:
                segs.append(st.tail.reshape(n, H))
            segs.append(st.tts_pad.reshape(1, H))
            off.append(o); ln.append(n); pos.append(st.tail_pos)
            o += n + 1
        self._tt = torch.cat(segs, 0) if segs else None
        self._tt_rows = list(self.rows)
        self._tt_off, self._tt_len, self._tt_pos = np.asarray(off, np.int64), np.asarray(ln, np.int64), np.asarray(pos, np.int64)

    def _update_states(self, so: OmniSchedulerOutput) -> None:
        e = self.engine
        if so.finished_req_ids or so.preempted_req_ids or so.scheduled_new_reqs:
            self._tt_flush()
        # drop finished / preempted requests, closing holes with the last row (condense); a preempted request keeps its
        # host state (outputs, per-step codes, text queue): its KV is recomputed when the scheduler brings it back
        for rid in so.finished_req_ids:
            self.preempted.pop(rid, None)
        for rid in list(so.finished_req_ids) + list(so.preempted_req_ids):
            if rid not in self.requests:
                continue
            if rid in so.preempted_req_ids and rid not in so.finished_req_ids:
                self.preempted[rid] = self.requests[rid]
            r = self.rows.index(rid)
            last = len(self.rows) - 1
            if r != last:
                for name in _ROW_BUFFERS:
                    buf = getattr(e, name)
                    buf[r] = buf[last]
                self.rows[r] = self.rows[last]
            self.rows.pop()
            self._reset_row(last)
            del self.requests[rid]
            if rid not in self.preempted:
                self.model_intermediate_buffer.pop(rid, None)                # gpu_model_runner.py:259
        # new requests (a preempted request re-enters here with fresh blocks and num_computed_tokens = 0)
        for nr in so.scheduled_new_reqs:
            if len(self.rows) >= self.max_num_seqs:
                raise RuntimeError(f"batch overflow: max_num_seqs={self.max_num_seqs}")
            if nr.req_id in self.preempted:
                self._resume(nr.req_id, list(nr.block_ids[0]), int(nr.num_computed_tokens))
                continue
            info = decode_additional_information(nr.additional_information)
            pe = nr.prompt_embeds if nr.prompt_embeds is not None else info.get("talker_prompt_embeds")
            tail = info.get("tailing_text_hidden")
            pad = info.get("tts_pad_embed")
            if pe is None and info.get("thinker_prefill_embeddings") is not None:
                if self.prompt_builder is None:
                    raise ValueError(f"request {nr.req_id}: thinker outputs given but the runner has no prompt_builder")
                prompt, upd = self.prompt_builder.from_info(info)     # qwen3_omni.py:678-809
                pe, tail, pad = prompt.embeds, upd.get("trailing_text_hidden"), prompt.tts_pad
                info.update(upd)      # (merged into model_intermediate_buffer below, once the request is admitted)
            elif pe is None and getattr(self.prompt_builder, "from_info", None) is not None and (info.get("text") or info.get("input_ids") is not None):
                # Qwen3-TTS request (text / task_type / speaker / voice_clone_prompt): _build_prompt_embeds on the device
                # (prompt_builder_tts.TTSTalkerPromptBuilder = qwen3_tts_talker.py:1211-1567)
                tp = self.prompt_builder.from_info(info)
                pe, tail, pad = tp.embeds, tp.trailing_text_hidden, tp.tts_pad
                if tp.ref_code is not None:
                    info.update({"ref_code": tp.ref_code, "ref_code_len": tp.ref_code_len})
            if pe is None or pe.ndim != 2 or pe.shape[1] != self.d.hidden:
                raise ValueError(f"request {nr.req_id}: missing talker_prompt_embeds [T,{self.d.hidden}]")
            dev = e.input_ids.device
            if pad is None:
                raise ValueError(f"request {nr.req_id}: missing tts_pad_embed (prefill must initialise it)")
            row_sampling = self._row_sampling(nr.req_id, nr.sampling_params or self.default_sampling)   # validated BEFORE any state changes
            st = RequestState(req_id=nr.req_id, prompt_embeds=pe.to(BF16).cpu().contiguous(),
                              block_ids=list(nr.block_ids[0]), sampling=nr.sampling_params or self.default_sampling,
                              num_computed=int(nr.num_computed_tokens),
                              tail=None if tail is None else self._up(tail, BF16).reshape(-1, self.d.hidden),
                              tts_pad=self._up(pad, BF16).reshape(-1), info=info)
            mp = getattr(nr, "mrope_positions", None)
            mp = info.get("mrope_positions") if mp is None else mp
            if mp is not None:
                mp = torch.as_tensor(mp, dtype=torch.int64).reshape(3, -1)
                md = getattr(nr, "mrope_position_delta", None)
                md = info.get("mrope_position_delta", 0) if md is None else md
                md = int(md.item() if isinstance(md, torch.Tensor) else md)
                if mp.shape[1] != st.prompt_len:
                    raise ValueError(f"request {nr.req_id}: mrope_positions cover {mp.shape[1]} tokens, the prompt has {st.prompt_len}")
                # every rotary id of the request's life must be a row of the cos / sin table (max_model_len rows): the prompt's
                # ids as given (video temporal ids run ahead of the token count), then index + delta up to the last token it may
                # emit (an id outside the table read device memory out of bounds)
                rows = int(getattr(e, "rope_rows", 0) or self.d.max_model_len)
                # last sequence index the request can reach: the scheduler stops it at max_model_len whatever max_tokens says
                # (max_tokens None / 0 = vLLM's "no cap": the bound is then max_model_len itself -- ADVICE r4)
                mt = int(getattr(st.sampling, "max_tokens", 0) or 0)
                end = (min(st.prompt_len + mt, int(self.d.max_model_len)) if mt > 0 else int(self.d.max_model_len)) - 1
                lo = min([st.prompt_len + md, end + md] + ([int(mp.min())] if mp.numel() else []))
                hi = max([st.prompt_len + md, end + md] + ([int(mp.max())] if mp.numel() else []))
                if lo < 0 or hi >= rows:
                    raise ValueError(f"request {nr.req_id}: M-RoPE ids {lo}..{hi} (prompt ids, then index + delta {md} up to sequence index {end}) "
                                     f"leave the rotary table [0, {rows})")
                if getattr(e, "rope_delta", None) is None and not (torch.equal(mp[0], mp[1]) and torch.equal(mp[0], mp[2]) and md == 0
                                                                   and torch.equal(mp[0], torch.arange(mp.shape[1]))):
                    raise ValueError(f"request {nr.req_id}: M-RoPE ids with differing rows, but the model has no mrope_section")
                if getattr(e, "rope_delta", None) is not None:
                    st.mrope_positions, st.mrope_delta = mp, md
            self.requests[nr.req_id] = st
            # per-request model state the stage hooks read and extend (gpu_model_runner.py:60,371,935): same dict as st.info
            self.model_intermediate_buffer[nr.req_id] = st.info
            r = len(self.rows)
            self.rows.append(nr.req_id)
            self._reset_row(r)
            e.set_row_sampling(r, **row_sampling)
            e.block_table[r, :len(st.block_ids)] = self._up(st.block_ids, torch.int32)
        # cached requests: new blocks (block_table.append_row, gpu_model_runner.py:489); vLLM's own scheduler brings a
        # preempted request back HERE, flagged resumed_from_preemption, with its complete new block list (:470-489)
        c = so.scheduled_cached_reqs
        upd_at, upd_val = [], []
        for i, rid in enumerate(c.req_ids):
            resumed = i < len(c.resumed_from_preemption) and c.resumed_from_preemption[i]
            if resumed and rid in self.preempted:
                nb = c.new_block_ids[i] if i < len(c.new_block_ids) else None
                self._resume(rid, list(nb[0]) if nb else [], int(c.num_computed_tokens[i]) if i < len(c.num_computed_tokens) else 0)
                continue
            st = self.requests[rid]
            nb = c.new_block_ids[i] if i < len(c.new_block_ids) else None
            if nb:
                new = list(nb[0])
                r = self.rows.index(rid)
                base = r * int(e.block_table.shape[1]) + len(st.block_ids)
                upd_at.extend(range(base, base + len(new)))
                upd_val.extend(new)
                st.block_ids.extend(new)
        if upd_at:
            # the step's new blocks of ALL rows in one pinned upload + one scatter (a pageable copy per row would hold the host
            # until the step in flight had drained: profiles/r05_pinned_read_probe.txt)
            t = self._h2d("bt_upd", np.asarray([upd_at, upd_val], np.int64), torch.int64)
            e.block_table.view(-1).index_put_((t[0],), t[1].to(torch.int32))

    def _update_intermediate_buffer(self, req_id: str, upd: dict) -> None:
        """Merge a per-request update into ``model_intermediate_buffer`` (V/worker/gpu_model_runner.py:1330-1353, known answers
        T/worker/test_omni_gpu_model_runner.py:172-227): tensors go to the host unless the key is declared GPU-resident, tensors
        inside lists likewise, successive updates accumulate, an empty update or an unknown request is a no-op, and the same
        dict stays reachable as the request state's ``additional_information_cpu`` (here ``info``)."""
        if not isinstance(upd, dict) or not upd:
            return
        st = self.requests.get(req_id)
        if st is None:
            return
        existing = self.model_intermediate_buffer.setdefault(req_id, {})
        for k, v in upd.items():
            if isinstance(v, torch.Tensor):
                existing[k] = v.detach().clone() if k in self.gpu_resident_buffer_keys else v.detach().to("cpu").contiguous()
            elif isinstance(v, list):
                existing[k] = [(i.detach().to("cpu").contiguous() if isinstance(i, torch.Tensor) else i) for i in v]
            else:
                existing[k] = v
        st.info = existing
        st.additional_information_cpu = existing

    @staticmethod
    def _row_sampling(req_id: str, sp) -> dict:
        """One request's SamplingParams as the engine's per-row arguments, checked before the runner touches its state (a bad
        request is refused alone; its neighbours' step goes on).  vLLM: a seeded request owns a torch.Generator seeded with
        it, an unseeded one draws from the global stream (gpu_model_runner.py:315-319) -- here every request gets its own
        counter-RNG key: its seed, or one derived from the request id, so unseeded neighbours never share noise.  top_p < 1
        with top_k disabled (a common vLLM setting) runs as top_k = 1024, the nucleus over the sampler's candidate capacity."""
        greedy = bool(sp.greedy)
        top_k, top_p = int(sp.top_k or 0), float(sp.top_p if sp.top_p is not None else 1.0)
        if not greedy and not float(sp.temperature) > 0.0:
            raise ValueError(f"request {req_id}: temperature must be > 0 when sampling")
        if not float(sp.repetition_penalty) > 0.0:
            raise ValueError(f"request {req_id}: repetition_penalty must be > 0")
        if 0.0 < top_p < 1.0 and not 0 < top_k <= 1024:
            top_k = 1024
        return dict(greedy=greedy, temperature=float(sp.temperature), top_k=top_k, top_p=top_p,
                    rep_penalty=float(sp.repetition_penalty), seed=request_seed(req_id, sp))

    # ------------------------------------------------------------------ recompute preemption
    def _rebuild_decode_inputs(self, st: RequestState, J: int) -> torch.Tensor:
        """The backbone inputs of the first J decode steps the request already took, from what it emitted:
        x_j = bf16(bf16(embed[c0] + sum_g cp_embed[g-1][c_g]) + text_j) with codes_j = its audio codes of step j (an invalid
        layer-0 id zeroes the frame's codes but e0 stays the id's embedding) and text_j = its queue entry j or tts_pad --
        the arithmetic of talker_mtp (qwen3_tts_talker.py:1630-1641), summed in the native kernel's order (e0 first,
        groups ascending, fp32), so the rows are bit-identical to the ones the decode steps fed the backbone."""
        e, d = self.engine, self.d
        dev = e.input_ids.device
        codes = torch.as_tensor(st.codes_hist[:J], dtype=torch.long, device=dev).reshape(J, d.num_code_groups)
        c0 = torch.as_tensor(st.output_ids[:J], dtype=torch.long, device=dev)
        ok = (c0 >= 0) & (c0 < d.vocab)
        acc = torch.where(ok[:, None], e.embed[c0.clamp(0, d.vocab - 1)].float(), torch.zeros((), device=dev))
        for g in range(1, d.num_code_groups):
            acc = acc + e.cp_embed[g - 1][codes[:, g]].float()
        nt = 0 if st.tail is None else int(st.tail.shape[0])
        text = torch.stack([st.tail[j] if j < nt else st.tts_pad for j in range(J)])
        return (acc.to(BF16).float() + text.float()).to(BF16)

    def _resume(self, rid: str, block_ids: list[int], num_computed: int) -> None:
        """A preempted request comes back: new blocks, KV gone.  Its prefill now covers the prompt AND the inputs of the
        L - 1 decode steps behind its L emitted tokens; when that completes nothing is sampled -- the row continues as a
        decode row with input_ids = its last token, h = the hidden state of the last recomputed position, and its
        position, step counter (RNG key), repetition-penalty bitmap and text-queue cursor where they were."""
        e = self.engine
        # the rows are about to change: the per-step text table is keyed by row (the cached-request route gets here without
        # the flush at the top of _update_states: a resumed row would read a neighbour's text step otherwise)
        self._tt_flush()
        st = self.preempted.pop(rid)
        if len(self.rows) >= self.max_num_seqs:
            raise RuntimeError(f"batch overflow: max_num_seqs={self.max_num_seqs}")
        P = st.prompt_len
        L = len(st.output_ids)
        if st.n_sampled != L:
            # (cannot happen behind this package's scheduler: no admission in a preempting step, and the output of the step
            # in flight is read before the next schedule)
            raise RuntimeError(f"request {rid}: resumed with {st.n_sampled - L} sampled token(s) still in flight")
        J = max(L - 1, 0)
        if len(st.codes_hist) < J:
            raise RuntimeError(f"request {rid}: {len(st.codes_hist)} decode steps recorded, {J} needed to recompute")
        base = st.prompt_embeds[:P]
        if J:
            base = torch.cat([base, self._rebuild_decode_inputs(st, J).cpu()], 0)
        st.n_prompt, st.recompute, st.prompt_embeds = P, J, base.contiguous()
        st.block_ids, st.num_computed, st.tail_pos = list(block_ids), int(num_computed), J
        self.requests[rid] = st
        r = len(self.rows)
        self.rows.append(rid)
        self._reset_row(r)
        e.set_row_sampling(r, **self._row_sampling(rid, st.sampling))
        e.block_table[r, :len(st.block_ids)] = self._up(st.block_ids, torch.int32)

    def _restore_decode_row(self, r: int, st: RequestState, hidden_last: torch.Tensor, position: int | None = None) -> None:
        e, d = self.engine, self.d
        L = len(st.output_ids)
        position = st.prefill_len if position is None else position
        e.input_ids[r] = int(st.output_ids[-1])
        e.last_hidden[r] = hidden_last
        e.positions[r] = position
        e.seq_lens[r] = position + 1
        if getattr(e, "rope_delta", None) is not None:
            e.rope_delta[r] = st.mrope_delta
        e.steps[r] = L
        e.seen[r].zero_()
        ids = [d.codec_pad_id] + [t for t in st.output_ids if 0 <= t < d.vocab]
        e.seen[r, self._up(ids, torch.long)] = 1

    # ------------------------------------------------------------------ phase 1
    @torch.inference_mode()
    def execute_model(self, scheduler_output: OmniSchedulerOutput, intermediate_tensors=None):
        if self.execute_model_state is not None:
            raise RuntimeError("State error: sample_tokens() must be called after execute_model() returns None.")
        e = self.engine
        with _Range("gpu_model_runner: preprocess"):
            # [Omni] KV transfer BEFORE updating states (which removes finished requests)
            self.kv_extracted_req_ids = self.kv_transfer_manager.handle_finished_requests_kv_transfer(
                finished_reqs=scheduler_output.finished_requests_needing_kv_transfer, kv_caches=self.kv_caches,
                block_size=e.block_size, cache_dtype=str(e.kv_dtype), kv_scales=getattr(e, "kv_scales", None),
                tp_rank=getattr(e, "tp_rank", 0), tp_size=getattr(e, "tp_size", 1)) or None
            self._update_states(scheduler_output)
        if not scheduler_output.total_num_scheduled_tokens:
            if self.kv_extracted_req_ids:
                # nothing to run but a KV extraction happened: ack it now (the reference returns the bare empty output,
                # gpu_ar_model_runner.py:152-166, and the ack is overwritten by the next step -- the held blocks leak)
                ack, self.kv_extracted_req_ids = self.kv_extracted_req_ids, None
                return OmniModelRunnerOutput(req_ids=[], req_id_to_index={}, sampled_token_ids=[], kv_extracted_req_ids=ack)
            return EMPTY_MODEL_RUNNER_OUTPUT

        sched = scheduler_output.num_scheduled_tokens
        # a request's scheduled tokens split into a prefill part (prompt, or prompt + recomputed decode inputs after a
        # preemption) and at most ONE decode token; decode-first row order
        npre: dict[str, int] = {}
        dec = []
        for i, rid in enumerate(self.rows):
            n = sched.get(rid, 0)
            if n <= 0:
                continue
            st = self.requests[rid]
            npre[rid] = min(n, max(st.prefill_len - st.num_computed, 0))
            if n - npre[rid] > 1:
                raise RuntimeError("decode requests are scheduled one token per step (no spec decode on this path)")
            if n - npre[rid] == 1:
                if npre[rid] and not st.n_sampled:
                    raise RuntimeError(f"request {rid}: scheduled past its prompt before its first token was sampled")
                dec.append(i)
        decs = set(dec)
        rest = [i for i in range(len(self.rows)) if i not in decs]
        self._permute_rows(dec + rest)
        nd = len(dec)

        # ---- prefill spans (chunked prefill: a span is any slice of the prompt)
        prefill_done: dict[int, torch.Tensor] = {}
        spans: dict[int, tuple[int, int, torch.Tensor]] = {}
        xs, pos, req, slots, meta, rope = [], [], [], [], [], []
        any_mrope = False
        bs = e.block_size
        for r in range(len(self.rows)):
            rid = self.rows[r]
            n = npre.get(rid, 0)
            if n <= 0:
                continue
            st = self.requests[rid]
            s0 = st.num_computed
            take = st.prompt_embeds[s0:s0 + n]
            if take.shape[0] < n:   # placeholder longer than the embeddings: pad with tts_pad (talker.py:573-578)
                take = torch.cat([take, st.tts_pad.cpu().reshape(1, -1).expand(n - take.shape[0], -1)], 0)
            xs.append(take)
            p = np.arange(s0, s0 + n)
            pos.append(p)
            req.append(np.full(n, r))
            slots.append(np.asarray(st.block_ids)[p // bs] * bs + p % bs)
            meta.append((r, s0, n))
            rope.append(st.rope_ids(s0, n))
            any_mrope = any_mrope or st.mrope_positions is not None or st.mrope_delta != 0
        sampled, lp_pre = None, None
        if xs:
            dev = e.input_ids.device
            x = self._up(torch.cat(xs, 0))
            hid = e.prefill(x, self._up(np.concatenate(pos), torch.int32), self._up(np.concatenate(req), torch.int32),
                            self._up(np.concatenate(slots), torch.int64),
                            **({"rope_positions": self._up(torch.cat(rope, 1), torch.int32)} if any_mrope else {}))
            o = 0
            for r, s0, n in meta:
                st = self.requests[self.rows[r]]
                spans[r] = (s0, n, hid[o:o + n])
                st.num_computed = s0 + n
                if st.in_decode:
                    if st.n_sampled:       # resumed after preemption: nothing to sample, the row picks up where it was
                        self._restore_decode_row(r, st, hid[o + n - 1])
                    else:
                        prefill_done[r] = hid[o + n - 1]
                o += n
            if prefill_done:
                rows_done = sorted(prefill_done)
                hl = torch.stack([prefill_done[r] for r in rows_done])
                logits = e.compute_logits(hl)
                sampled = self._sample_prefill(rows_done, logits)
                kk = self._wants_logprobs([self.requests[self.rows[r]] for r in rows_done])
                if kk >= 0:
                    lp_pre = (rows_done, self._logprob_rows(logits, sampled, kk))
                for j, r in enumerate(rows_done):
                    st = self.requests[self.rows[r]]
                    e.last_hidden[r] = hl[j]
                    e.positions[r] = st.prompt_len
                    e.seq_lens[r] = st.prompt_len + 1
                    if getattr(e, "rope_delta", None) is not None:
                        e.rope_delta[r] = st.mrope_delta
                e.input_ids[self._up(rows_done, torch.long)] = sampled

        # ---- decode rows: text-step queue pop (talker.py:618-629), then the native step
        if nd:
            if self._tt is None:
                self._tt_build()
            idx = self._tt_off[:nd] + np.minimum(self._tt_pos[:nd], self._tt_len[:nd])
            if len(idx) != nd:
                raise RuntimeError(f"text-step table covers {len(idx)} rows, the step has {nd} decode rows (stale table)")
            self._tt_pos[:nd] += 1
            torch.index_select(self._tt, 0, self._h2d("tt_idx", idx, torch.int64), out=e.text_step[:nd])
            with _Range("gpu_model_runner: forward"):      # the native step: talker_mtp, backbone, compute_logits AND the sampler
                self._run_decode(nd)
        lp_rows: dict = {}
        if lp_pre is not None:
            lp_rows["pre"] = lp_pre
        if nd:
            kk = self._wants_logprobs([self.requests[self.rows[r]] for r in range(nd)])
            if kk >= 0:        # behind the step, from the logits its lm_head left and the ids its sampler drew (device, stream-ordered)
                lp_rows["dec"] = (list(range(nd)), self._logprob_rows(e.logits[:nd], e.input_ids[:nd], kk))
        # host counters move at DISPATCH (the scheduler advances num_computed_tokens when it schedules, too): the next
        # execute_model may run before this step's ids have been read
        sched_rows = []
        for r, rid in enumerate(self.rows):
            if sched.get(rid, 0) <= 0:
                continue
            st = self.requests[rid]
            if r < nd:
                st.num_computed += 1
                st.n_sampled += 1
            elif r in prefill_done:
                st.n_sampled += 1
            sched_rows.append((r, rid, st))
        self.execute_model_state = _StepState(scheduler_output, list(range(nd)), prefill_done, sampled, spans, sched_rows, len(self.rows),
                                              lp_rows)
        return None

    def _logprob_rows(self, logits: torch.Tensor, ids: torch.Tensor, k: int):
        """vLLM Sampler.gather_logprobs on the step's own (masked, raw: before temperature / penalties -- vLLM's default
        `raw_logprobs` mode) logits: per row the sampled token's log-probability, the top-k alternatives, the sampled token's rank
        (1 + how many tokens are more probable) and the NaN count of the row (num_nans_in_logits).  Device tensors; no sync."""
        lg = logits.float()
        lp = torch.log_softmax(lg, -1)
        sel = lp.gather(1, ids.long().reshape(-1, 1))
        topv, topi = lp.topk(k, -1) if k > 0 else (lp[:, :0], ids.long().reshape(-1, 1)[:, :0])
        rank = (lp > sel).sum(-1) + 1
        return (torch.cat([ids.long().reshape(-1, 1), topi], 1), torch.cat([sel, topv], 1), rank, torch.isnan(lg).sum(-1))

    def _wants_logprobs(self, rows) -> int:
        """-1: nobody asked; else the widest request's top-k (every sampled row of the step gets that many: vLLM's max_num_logprobs)."""
        ks = [int(st.sampling.logprobs) for st in rows if getattr(st.sampling, "logprobs", None) is not None]
        return max(ks) if ks else -1

    def _up(self, t, dtype=None) -> torch.Tensor:
        """Any host tensor / list / array onto the device through pinned memory (admission-time uploads: prompt embeddings, index
        lists): never a pageable copy, which would hold the host until the step in flight has finished."""
        dev = self.engine.input_ids.device
        t = t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t))
        if dtype is not None:
            t = t.to(dtype)
        if dev.type != "cuda" or t.device.type == "cuda":
            return t.to(dev)
        return t.contiguous().pin_memory().to(dev, non_blocking=True)

    def _h2d(self, key: str, values, dtype) -> torch.Tensor:
        """A small host array onto the device WITHOUT stalling the host: a pageable source makes the copy wait until the stream
        has drained (measured: profiles/r05_pinned_read_probe.txt -- the host would sit out the step in flight instead of running
        ahead of it); a pinned source is queued behind it.  One ring of pinned buffers AND one ring position per call site (a
        single shared counter made a key's reuse distance depend on how often the OTHER keys were used): a key's buffer is rewritten
        eight calls OF THAT KEY later, long after its copy ran (at most two steps are in flight, a key is used at most twice per step)."""
        dev = self.engine.input_ids.device
        arr = np.ascontiguousarray(values)
        if dev.type != "cuda":
            return torch.as_tensor(arr, dtype=dtype, device=dev)
        n = int(arr.size)
        ring = self._pin.get(key)
        if ring is None or ring[0].numel() < n or ring[0].dtype != dtype:
            cap = max(64, 1 << (max(n, 1) - 1).bit_length())
            ring = self._pin[key] = [torch.empty(cap, dtype=dtype).pin_memory() for _ in range(8)]
        i = self._pin_i[key] = self._pin_i.get(key, -1) + 1
        buf = ring[i % len(ring)]
        buf[:n].copy_(torch.from_numpy(arr.reshape(-1)).to(dtype))
        return buf[:n].to(dev, non_blocking=True).reshape(arr.shape)

    def _sample_prefill(self, rows: list[int], logits: torch.Tensor) -> torch.Tensor:
        """First token of the requests whose prompt completed this step, each with ITS request's sampling parameters
        (the per-row device arrays the decode sampler reads)."""
        e = self.engine
        idx = self._up(rows, torch.long)
        seen = e.seen.index_select(0, idx)
        seen[:, self.d.codec_pad_id] = 1       # prompt ids are codec_pad placeholders (talker.py:603-605)
        steps = torch.zeros(len(rows), dtype=torch.int32, device=logits.device)
        ids = e.sample_rows(logits, idx, seen=seen, steps=steps)
        e.seen[idx] = seen
        e.steps[idx] = steps
        return ids

    def _bucket(self, n: int) -> int:
        """Padded batch size a step of n live decode rows runs at: 1, 2, 4, 8, then multiples of 8 (vLLM's default cudagraph capture
        sizes).  Powers of two up to 64 put 33 live rows into the 64-row graph; a step costs 3.13 / 3.23 / 3.43 / 3.48 ms at 12 / 25 / 48 /
        64 rows (profiles/r05_ab_table.txt), so the finer buckets are worth their eleven captures."""
        if n <= 8:
            b = 1
            while b < n:
                b *= 2
        else:
            b = (n + 7) // 8 * 8
        return min(b, self.max_num_seqs)

    def capture_graphs(self, sizes=None) -> None:
        """hipGraph per padded batch bucket (reference: _dummy_run capture, gpu_model_runner.py:536-876).
        Capture records the launches without running them, so live state is untouched."""
        if not self.use_graphs or not torch.cuda.is_available():
            return
        sizes = sizes or sorted({self._bucket(n) for n in range(1, self.max_num_seqs + 1)})
        self.engine.decode_step(min(sizes), advance=False)        # one eager pass loads code objects
        torch.cuda.synchronize()
        for b in sizes:      # (layer-0 sampling parameters are per-row device arrays: nothing of them is baked in)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.engine.decode_step(b)
            self.graphs[b] = g

    def _redo_step(self, stt: _StepState, rec: _HostRecord, code: int) -> None:
        """The decode part of step `stt` is invalid: a persistent chain's flag wait timed out in it (status word 0: the chain's
        grid was not co-resident -- another process or engine held CUs), or it ran behind such a step (async scheduling: it was
        dispatched on that step's garbage).  Nothing of it has left the runner.  Degrade instead of dying: drain the device,
        clear its error words and turn this engine's chains off (launch-per-op shares a GPU without a deadline), re-capture the
        graphs, put every decode row of the step back to the state the step started from -- last id, h[t], position, step
        counter, repetition bitmap, text step, all from the host records of the steps BEFORE it (bookkept by now) -- and run the
        step again, synchronously; `rec` gets the redone rows' outputs.  The KV slots the bad step wrote are written again.
        Rows of the step that have since left the batch (preempted at the next dispatch) are dropped from its output: the
        scheduler and the runner both simply never see that token, the request recomputes and carries on from the one before."""
        e = self.engine
        dev = e.input_ids.device
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if getattr(e, "persistent_chains", False) or code:
            logger.warning("persistent-chain flag wait timed out (stage code %#x): engine falls back to the launch-per-op path, "
                           "step redone", code)
            e.recover_from_chain_timeout()
            sizes = sorted(self.graphs)
            self.graphs.clear()
            if self.use_graphs and sizes:
                e.num_live.fill_(0)             # the capture's eager warm-up pass must not touch live rows
                self._num_live = 0
                self.capture_graphs(sizes)
            self.chain_fallbacks += 1
            self._chains_parked, self._clean_steps = hasattr(e, "set_chains"), 0
            self._rearm_after *= 2
        dec = [(r, rid, st) for r, rid, st in stt.sched_rows if r < len(stt.decode_rows)]
        live = [(r, rid, st) for r, rid, st in dec if self.requests.get(rid) is st and rid in self.rows]
        rec.dropped = {r for r, _, _ in dec} - {r for r, _, _ in live}
        nd = len(live)
        if not nd:
            return
        cur = [self.rows.index(rid) for _, rid, _ in live]
        rest = [i for i in range(len(self.rows)) if i not in set(cur)]
        self._permute_rows(cur + rest)              # the step's decode rows first, in the step's own order
        text = []
        for k, (r, rid, st) in enumerate(live):
            L = len(st.output_ids)
            if not L:
                raise RuntimeError(f"request {rid}: no record to redo its decode step from")
            span = stt.prefill_spans.get(r)         # resumed after preemption: recomputed AND decoded in this step
            hid = span[2][span[1] - 1] if span is not None else st.last_hidden_cpu
            if hid is None:
                raise RuntimeError(f"request {rid}: no record to redo its decode step from")
            self._restore_decode_row(k, st, hid.reshape(-1).to(dev), position=st.prompt_len + L - 1)
            j, nt = L - 1, 0 if st.tail is None else int(st.tail.shape[0])       # decode step j of the request reads text row j
            text.append(st.tail[j] if j < nt else st.tts_pad)
        e.text_step[:nd] = torch.stack(text)
        self._run_decode(nd)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        buf = e.ids_status.cpu() if getattr(e, "ids_status", None) is not None else None
        ids = buf[:nd].tolist() if buf is not None else e.input_ids[:nd].cpu().tolist()
        status = buf[-4:].tolist() if buf is not None else [0, 0, 0, 0]
        if status[0]:
            raise RuntimeError(f"decode step invalid after the fall-back to the launch path (status {status})")
        hid_cpu = e.last_hidden[:nd].to("cpu", copy=True)
        codes_cpu = e.audio_codes[:nd].cpu()
        rec.hidden, rec.codes = rec.hidden.clone(), rec.codes.clone()
        for k, (r, _, _) in enumerate(live):        # back into the record at the step's own row numbers
            rec.ids[r] = ids[k]
            rec.hidden[r] = hid_cpu[k]
            rec.codes[r] = codes_cpu[k]
        rec.status = [0, status[1], status[2], 0]
        if any(r in rec.logprobs for r, _, _ in live):
            kk = max(len(rec.logprobs[r][0]) for r, _, _ in live if r in rec.logprobs) - 1
            tid, lp, rank, nans = (t.cpu().tolist() for t in self._logprob_rows(e.logits[:nd], e.input_ids[:nd], kk))
            for k, (r, _, _) in enumerate(live):
                rec.logprobs[r] = (tid[k], lp[k], int(rank[k]), int(nans[k]))

    def _run_decode(self, nd: int) -> None:
        # rows [nd, bucket) of the padded graph may be live PREFILL rows of the persistent batch (decode-first order):
        # the device-side live count keeps the step off their KV blocks, ids, hidden state and counters
        if self._chains_parked:
            self._clean_steps += 1
            if self._clean_steps >= self._rearm_after and not self._inflight_dirty():
                self._rearm_chains()
        if self._num_live != nd:
            self.engine.num_live.fill_(nd)
            self._num_live = nd
        g = self.graphs.get(self._bucket(nd)) if self.use_graphs else None
        if g is not None:
            g.replay()
            self.cudagraph_stats["replays"] += 1
        else:
            self.engine.decode_step(nd)
            self.cudagraph_stats["eager_steps"] += 1

    def _inflight_dirty(self) -> bool:
        return any(h.record is not None for h in self._inflight)

    def _rearm_chains(self) -> None:
        """Give the persistent chains another try after a long clean stretch on the launch path (same bits either way)."""
        e = self.engine
        logger.info("persistent chains re-armed after %d clean launch-path steps (fall-backs so far: %d)", self._clean_steps, self.chain_fallbacks)
        if e.input_ids.device.type == "cuda":
            torch.cuda.synchronize()
        e.set_chains(True)
        sizes = sorted(self.graphs)
        self.graphs.clear()
        if self.use_graphs and sizes:
            e.num_live.fill_(0)
            self._num_live = 0
            self.capture_graphs(sizes)
        self._chains_parked, self._clean_steps = False, 0

    # ------------------------------------------------------------------ phase 2
    def _snapshot(self, stt: _StepState):
        """Put the step's device outputs where the next step cannot overwrite them, in stream order, and return what
        `_collect` needs to bring them to the host.  Sync mode / CPU engines: nothing to do (the buffers are read at once)."""
        e = self.engine
        dev = e.input_ids.device
        if not self.async_scheduling:
            return None
        n, nd = stt.n_rows, len(stt.decode_rows)
        if dev.type != "cuda":
            ids = e.ids_status.clone() if getattr(e, "ids_status", None) is not None else e.input_ids[:n].clone()
            return ("host", ids, e.last_hidden[:n].clone(), e.audio_codes[:nd].clone() if nd else None, dict(stt.prefill_spans))
        rec = getattr(e, "out_record", None)
        if rec is None:
            raise RuntimeError("async scheduling needs an engine with a packed output record (engine.out_record)")
        if not self._stage_dev:
            self._stage_dev = [torch.empty_like(rec) for _ in range(4)]
            self._copy_stream = torch.cuda.Stream(device=dev)
        self._stage_i += 1
        slot = self._stage_dev[self._stage_i % len(self._stage_dev)]
        slot.copy_(rec, non_blocking=True)           # ONE device-to-device copy behind the step: ids + status, codes, h
        ev = torch.cuda.Event()
        ev.record()
        return ("dev", slot, ev, dict(stt.prefill_spans))

    def _collect(self, stt: _StepState, pending) -> _HostRecord:
        """The step's outputs on the host.  Async: waits for the step's snapshot only -- the copy runs on a side stream behind
        the snapshot's event, into pageable memory (a pinned target reads at ~0.5 GB/s from the CPU: profiles/r05_pinned_read_probe.txt)."""
        e = self.engine
        n, nd = stt.n_rows, len(stt.decode_rows)
        if pending is None:
            # one D2H for the whole batch (the reference syncs on hidden_states.to("cpu"), :550); the ids bring the status words along
            ids_status = getattr(e, "ids_status", None)
            if ids_status is not None:
                buf = ids_status.cpu()
                ids, status = buf[:n].tolist(), (buf[-4:].tolist() if nd else [0, 0, 0, 0])
            else:
                ids, status = e.input_ids[:n].cpu().tolist(), [0, 0, 0, 0]
            hid = e.last_hidden[:n].to("cpu", copy=True)          # a copy of its own: requests keep views of it
            codes = e.audio_codes[:nd].cpu() if nd else None
            spans = {r: h.cpu() for r, (_, _, h) in stt.prefill_spans.items()}
            return _HostRecord(ids, status, hid, codes, spans, logprobs=self._logprobs_host(stt))
        if pending[0] == "host":
            _, buf, hid, codes, sp = pending
            has_status = getattr(e, "ids_status", None) is not None
            ids = buf[:n].tolist()
            status = buf[-4:].tolist() if (has_status and nd) else [0, 0, 0, 0]
            return _HostRecord(ids, status, hid, codes, {r: h.cpu() for r, (_, _, h) in sp.items()}, logprobs=self._logprobs_host(stt))
        _, slot, ev, sp = pending
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ev)
            host = slot.cpu()                        # blocks on the side stream alone: the step after this one keeps running
            spans = {r: h.cpu() for r, (_, _, h) in sp.items()}
            lps = self._logprobs_host(stt)
        ids_all, codes_all, hid_all = e.split_out_record(host)
        ids = ids_all[:n].tolist()
        status = ids_all[-4:].tolist() if nd else [0, 0, 0, 0]
        return _HostRecord(ids, status, hid_all[:n], codes_all[:nd] if nd else None, spans, logprobs=lps)

    @staticmethod
    def _logprobs_host(stt: _StepState) -> dict:
        """row -> (token ids, logprobs, rank, NaN count) on the host (tensors computed at dispatch, behind the step)."""
        out: dict = {}
        for rows, (tid, lp, rank, nans) in stt.logprobs.values():
            tid, lp, rank, nans = tid.cpu().tolist(), lp.cpu().tolist(), rank.cpu().tolist(), nans.cpu().tolist()
            for j, r in enumerate(rows):
                out[r] = (tid[j], lp[j], int(rank[j]), int(nans[j]))
        return out

    @torch.inference_mode()
    def sample_tokens(self, grammar_output=None):
        """Phase 2 of the step (gpu_ar_model_runner.py:403-660).  The sampler itself ran inside the native step; this is the
        hand-over of its outputs: at once (an OmniModelRunnerOutput), or -- async scheduling -- as an AsyncStepOutput whose
        get_output() the engine core calls after it has dispatched the next step (:641-660)."""
        kv_extracted = self.kv_extracted_req_ids
        self.kv_extracted_req_ids = None
        if self.execute_model_state is None:
            return None
        stt, self.execute_model_state = self.execute_model_state, None
        handle = AsyncStepOutput(self, stt, self._snapshot(stt), kv_extracted)
        if not self.async_scheduling:
            return handle.get_output()
        self._inflight.append(handle)
        return handle

    def _finish_step(self, handle: AsyncStepOutput) -> OmniModelRunnerOutput:
        if self._inflight and self._inflight[0] is not handle and handle in self._inflight:
            for h in list(self._inflight):           # outputs are finished in dispatch order (records build on each other)
                if h is handle:
                    break
                h.get_output()
        stt = handle.stt
        nd = len(stt.decode_rows)
        with _Range("gpu_model_runner: sample"):     # the wait for the step plus the copy of its outputs
            rec = handle.record if handle.record is not None else self._collect(stt, handle.pending)
        redone = False
        if handle.record is None and nd and rec.status[0]:
            # a chain flag wait timed out: fall back to the launch-per-op path and redo THIS step
            self._redo_step(stt, rec, int(rec.status[0]))
            redone = True
        if rec.status[1]:
            # a tensor-parallel peer did not arrive: the kernel that timed out wrote the word into every rank's control
            # block, so all ranks leave their step here (the reference: an exception in the worker ends the engine)
            raise RuntimeError(f"peer all-reduce: a rank did not arrive in time (error word {rec.status[1]}): the step's outputs are invalid")
        out = self._bookkeep(stt, rec, handle.kv_extracted)
        if handle in self._inflight:
            self._inflight.remove(handle)
        if redone:
            # the steps dispatched since ran on this step's garbage: redo each of them now, oldest first, on the records just
            # written (their own get_output() then only does the bookkeeping)
            for h in list(self._inflight):
                h.record = self._collect(h.stt, h.pending)
                self._redo_step(h.stt, h.record, 0)
                if h is not self._inflight[-1]:
                    h.get_output()
        return out

    def _bookkeep(self, stt: _StepState, rec: _HostRecord, kv_extracted) -> OmniModelRunnerOutput:
        nd = len(stt.decode_rows)
        Q = self.d.num_code_groups
        with _Range("gpu_model_runner: postprocess"):       # the step's frames as host lists (the request records keep them)
            codes_list = rec.codes.tolist() if nd else []
        req_ids, sampled, pooler = [], [], []
        want_lp = bool(rec.logprobs)
        lp_ids, lp_vals, lp_rank, nans = [], [], [], {}
        with _Range("gpu_model_runner: bookkeep"):
            for r, rid, st in stt.sched_rows:
                payload: dict[str, Any] = {}
                if r < nd:
                    if r in rec.dropped:
                        st.n_sampled -= 1                # the token never existed (see _redo_step): the request recomputes without it
                        continue
                    tok = int(rec.ids[r])
                    st.output_ids.append(tok)
                    sampled.append([tok])
                    payload["hidden"] = rec.hidden[r:r + 1]                    # views of this step's own host copies
                    payload["audio_codes"] = rec.codes[r:r + 1]                # frame [c0..c15][t] (talker.py:1642)
                    st.codes_hist.append(codes_list[r])
                else:
                    s0, n, _ = stt.prefill_spans[r]
                    payload["hidden"] = rec.spans[r]
                    payload["audio_codes"] = torch.zeros(n, Q, dtype=torch.long)   # prefill rows: zero codes (607-612)
                    if r in stt.prefill_done:
                        tok = int(rec.ids[r])
                        st.output_ids.append(tok)
                        sampled.append([tok])
                    else:
                        sampled.append([])
                req_ids.append(rid)
                pooler.append(payload)
                if want_lp:
                    t = rec.logprobs.get(r) if sampled[-1] else None
                    lp_ids.append(t[0] if t else [])
                    lp_vals.append(t[1] if t else [])
                    lp_rank.append(t[2] if t else 0)
                    if t:
                        nans[rid] = t[3]
                st.last_hidden_cpu = rec.hidden[r]       # h[t] of the request's next decode step (a view of this step's host copy)
        return OmniModelRunnerOutput(
            req_ids=req_ids, req_id_to_index={rid: i for i, rid in enumerate(req_ids)}, sampled_token_ids=sampled,
            pooler_output=pooler if self.engine_output_type != "text" else None, kv_extracted_req_ids=kv_extracted,
            # SamplingParams.logprobs (gpu_ar_model_runner.py:516-519,631-636): filled when any request of the step asked
            logprobs=LogprobsLists(lp_ids, lp_vals, lp_rank) if want_lp else None, num_nans_in_logits=nans if want_lp else None,
            # which path the step ran on rides with every output (status word 2 of THIS step's record: bit 0 the code predictor's
            # chain, bit 1 the backbone's), with the fall-backs so far (a fall-back was visible in a log line only)
            cudagraph_stats=dict(self.cudagraph_stats, chain_fallbacks=self.chain_fallbacks, chains_ran=int(rec.status[2]),
                                 persistent_chains=bool(getattr(self.engine, "persistent_chains", False))))
