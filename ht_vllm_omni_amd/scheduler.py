"""Request scheduler of the talker AR stage and a minimal engine-core loop around it.

The reference's ``OmniARScheduler`` (V/core/sched/omni_ar_scheduler.py:40-590) subclasses vLLM's V1 ``Scheduler``
(third party, absent here); this module restates, as one standalone class, the part of both that the talker path
exercises:

  vLLM V1 semantics (stated here, the scheduler is its own oracle for them)
    * one token budget per step (``max_num_batched_tokens``), RUNNING requests first (a decode asks for 1 token, an
      unfinished prompt for the rest of it, chunked to the budget), then WAITING requests FCFS while fewer than
      ``max_num_seqs`` are running; no new admission in a step that had to preempt;
    * KV blocks from a free queue in id order, block 0 the null block, a freed request's blocks appended in reverse
      (``sched.BlockPool``); a request gets blocks for its computed + scheduled tokens before it runs; when the pool
      is exhausted the most recently admitted running request is preempted (blocks freed, recomputed later);
    * stop check after each sampled token: ``max_tokens`` / ``max_model_len`` -> length, EOS (unless ``ignore_eos``)
      or a ``stop_token_ids`` member -> stopped;
  Omni additions (omni_ar_scheduler.py)
    * new requests leave as ``OmniNewRequestData`` carrying ``prompt_embeds`` / ``additional_information`` (150-174);
    * KV hand-off to the next stage: criteria ``prefill_finished`` / ``special_token`` trigger a transfer once without
      stopping the request (80-137); a finished request that must ship its KV keeps its blocks
      (``waiting_for_transfer_free``) until the runner acks via ``kv_extracted_req_ids`` (456-479, 484-545); the
      metadata handed to the runner is ``{seq_len, block_ids}`` with the block list truncated to
      ceil(seq_len / block_size) (547-590) and is delivered exactly once (``finished_requests_needing_kv_transfer``);
    * with ``async_chunk`` every step's pooler output goes to the chunk transfer adapter (here:
      ``stage_input_processors.CodecChunkStreamer``).
"""
from __future__ import annotations

import enum
from collections import deque
from dataclasses import dataclass, field
from typing import Any

from .payloads import (OmniCachedRequestData, OmniModelRunnerOutput, OmniNewRequestData, OmniSchedulerOutput,
                       SamplingParams)
from .sched import BlockPool, truncate_blocks


class RequestStatus(enum.IntEnum):
    WAITING = 0
    RUNNING = 1
    PREEMPTED = 2
    FINISHED_STOPPED = 3
    FINISHED_LENGTH_CAPPED = 4
    FINISHED_ABORTED = 5

    @property
    def finished(self) -> bool:
        return self >= RequestStatus.FINISHED_STOPPED


_FINISH_REASON = {RequestStatus.FINISHED_STOPPED: "stop", RequestStatus.FINISHED_LENGTH_CAPPED: "length",
                  RequestStatus.FINISHED_ABORTED: "abort"}


@dataclass(eq=False)      # identity semantics: requests live in lists / deques that are searched and removed from
class Request:
    """The fields of vLLM's ``Request`` the path reads, plus the omni payloads (V/request.py)."""
    request_id: str
    num_prompt_tokens: int
    sampling_params: SamplingParams = field(default_factory=SamplingParams)
    prompt_token_ids: list[int] | None = None
    prompt_embeds: Any = None
    additional_information: Any = None
    external_req_id: str | None = None
    eos_token_id: int | None = None
    ignore_eos: bool = False
    status: RequestStatus = RequestStatus.WAITING
    num_computed_tokens: int = 0
    output_token_ids: list[int] = field(default_factory=list)
    stop_reason: int | None = None
    # async scheduling (vLLM Request.num_output_placeholders): tokens this request has been scheduled to sample whose ids the
    # scheduler has not seen yet -- the step that samples them is still in flight while the next one is scheduled
    num_output_placeholders: int = 0

    def __post_init__(self):
        if self.external_req_id is None:
            self.external_req_id = self.request_id

    @property
    def num_tokens(self) -> int:
        return self.num_prompt_tokens + len(self.output_token_ids)

    def is_finished(self) -> bool:
        return self.status.finished

    def get_finished_reason(self) -> str | None:
        return _FINISH_REASON.get(self.status)


@dataclass
class EngineCoreOutput:
    request_id: str
    new_token_ids: list[int]
    pooling_output: dict | None = None
    finish_reason: str | None = None
    stop_reason: int | None = None
    kv_transfer_params: dict | None = None
    new_logprobs: Any = None           # LogprobsLists slice of the request (omni_ar_scheduler.py:319-321), when it asked

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


class MI355XARScheduler:
    def __init__(self, *, num_blocks: int, block_size: int = 16, max_num_seqs: int = 64, max_num_batched_tokens: int = 8192,
                 max_model_len: int = 4096, kv_transfer_criteria: dict | None = None, need_send_cache: bool = False,
                 chunk_streamer=None, stage_id: int = 0, async_scheduling: bool = False):
        # async_scheduling (stage_configs/qwen3_tts.yaml:16; vLLM's AsyncScheduler): step t + 1 is scheduled before step t's
        # sampled ids are known -- a running request counts its in-flight token as a placeholder and is given its next decode
        # token at once; a stop is seen one step late (the surplus frame is dropped in update_from_output), a length cap is not
        # (the cap is known ahead: no step is scheduled past it)
        self.async_scheduling = bool(async_scheduling)
        self.block_size, self.max_num_seqs = block_size, max_num_seqs
        self.max_num_batched_tokens, self.max_model_len = max_num_batched_tokens, max_model_len
        self.pool = BlockPool(num_blocks, block_size)
        self.requests: dict[str, Request] = {}
        self.waiting: deque[Request] = deque()
        self.running: list[Request] = []
        self.finished_req_ids: set[str] = set()
        # omni KV hand-off state (omni_ar_scheduler.py:52-66)
        self.kv_transfer_criteria = kv_transfer_criteria
        self.need_send_cache = need_send_cache or bool(kv_transfer_criteria)
        self.requests_needing_kv_transfer: dict[str, dict[str, Any]] = {}
        self.waiting_for_transfer_free: set[str] = set()
        self.active_kv_transfers: set[str] = set()
        self.transfer_triggered_requests: set[str] = set()
        self.chunk_streamer, self.stage_id = chunk_streamer, stage_id

    # ------------------------------------------------------------------ requests
    def add_request(self, request: Request) -> None:
        if request.request_id in self.requests:
            raise ValueError(f"duplicate request id {request.request_id}")
        if request.num_prompt_tokens <= 0 or request.num_prompt_tokens >= self.max_model_len:
            raise ValueError(f"request {request.request_id}: prompt of {request.num_prompt_tokens} tokens outside (0, max_model_len)")
        self.requests[request.request_id] = request
        request.status = RequestStatus.WAITING
        self.waiting.append(request)

    def abort_request(self, request_id: str) -> None:
        req = self.requests.get(request_id)
        if req is None or req.is_finished():
            return
        if req in self.running:
            self.running.remove(req)
        elif req in self.waiting:
            self.waiting.remove(req)
        req.status = RequestStatus.FINISHED_ABORTED
        # whatever is still in flight for it is dropped on arrival (request finished); as on the stop path, the positions of the dropped
        # tokens do not count as computed -- a KV hand-off of the aborted request ships the synchronous loop's length
        req.num_computed_tokens -= req.num_output_placeholders
        req.num_output_placeholders = 0
        self._free_request(req)

    def has_unfinished_requests(self) -> bool:
        return bool(self.waiting or self.running)

    # ------------------------------------------------------------------ schedule
    def _preempt_one(self, keep: Request) -> bool:
        """Pool exhausted: the most recently admitted running request (not `keep`) gives its blocks back and will be
        recomputed from scratch (vLLM V1 recompute preemption)."""
        for victim in reversed(self.running):
            if victim is keep:
                continue
            self.running.remove(victim)
            self.pool.free_request(victim.request_id)
            victim.status = RequestStatus.PREEMPTED
            victim.num_computed_tokens = 0
            self.waiting.appendleft(victim)
            self._preempted.add(victim.request_id)
            return True
        return False

    def _allocate(self, req: Request, total_tokens: int) -> list[int] | None:
        while True:
            try:
                return self.pool.allocate(req.request_id, total_tokens)
            except MemoryError:
                if not self._preempt_one(req):
                    return None

    def schedule(self) -> OmniSchedulerOutput:
        budget = self.max_num_batched_tokens
        self._preempted: set[str] = set()
        new_reqs: list[OmniNewRequestData] = []
        cached = OmniCachedRequestData()
        num_sched: dict[str, int] = {}
        # 1) running requests, in admission order
        i = 0
        while i < len(self.running) and budget > 0:
            req = self.running[i]
            ph = req.num_output_placeholders
            if ph and (len(req.output_token_ids) + ph >= req.sampling_params.max_tokens or req.num_tokens + ph >= self.max_model_len):
                i += 1          # the token in flight is its last (length cap, known ahead): nothing more to schedule for it
                continue
            n = min(req.num_tokens + ph - req.num_computed_tokens, budget)
            if n <= 0:
                i += 1
                continue
            new_blocks = self._allocate(req, req.num_computed_tokens + n)
            if new_blocks is None:          # alone and still no room: leave it for a later step
                break
            if req.request_id in self._preempted:   # it preempted itself out (cannot happen: keep=req), defensive
                continue                            # (id test: `req not in self.running` compared whole dataclasses, 0.6 ms / step at 64)
            cached.req_ids.append(req.request_id)
            cached.resumed_from_preemption.append(False)
            cached.new_token_ids.append([])
            cached.new_block_ids.append((new_blocks,) if new_blocks else None)
            cached.num_computed_tokens.append(req.num_computed_tokens)
            num_sched[req.request_id] = n
            budget -= n
            i += 1
        # a preemption may have removed requests that were already scheduled above in this step
        for rid in list(num_sched):
            if rid in self._preempted:
                k = cached.req_ids.index(rid)
                for lst in (cached.req_ids, cached.resumed_from_preemption, cached.new_token_ids, cached.new_block_ids,
                            cached.num_computed_tokens):
                    lst.pop(k)
                budget += num_sched.pop(rid)
        # 2) waiting requests, FCFS, unless this step preempted
        while self.waiting and not self._preempted and budget > 0 and len(self.running) < self.max_num_seqs:
            req = self.waiting[0]
            if req.num_output_placeholders:      # preempted with a token still in flight: it comes back once that has arrived
                break
            n = min(req.num_tokens - req.num_computed_tokens, budget)
            need = self.pool.blocks_needed(req.num_computed_tokens + n) - len(self.pool.block_ids(req.request_id))
            if need > self.pool.num_free:
                break
            self.waiting.popleft()
            self.pool.allocate(req.request_id, req.num_computed_tokens + n)
            resumed = req.status == RequestStatus.PREEMPTED
            req.status = RequestStatus.RUNNING
            self.running.append(req)
            new_reqs.append(OmniNewRequestData(
                req_id=req.request_id, external_req_id=req.external_req_id, prompt_token_ids=req.prompt_token_ids,
                sampling_params=req.sampling_params, block_ids=(self.pool.block_ids(req.request_id),),
                num_computed_tokens=req.num_computed_tokens, prompt_embeds=req.prompt_embeds,
                additional_information=req.additional_information))
            del resumed      # a resumed request re-enters as a new one: its KV is recomputed
            num_sched[req.request_id] = n
            budget -= n
        # vLLM advances num_computed_tokens at schedule time (_update_after_schedule)
        for rid, n in num_sched.items():
            req = self.requests[rid]
            req.num_computed_tokens += n
            if self.async_scheduling and req.num_computed_tokens == req.num_tokens + req.num_output_placeholders:
                req.num_output_placeholders += 1         # this step samples a token for it (AsyncScheduler._update_after_schedule)
        out = OmniSchedulerOutput(
            scheduled_new_reqs=new_reqs, scheduled_cached_reqs=cached, num_scheduled_tokens=num_sched,
            total_num_scheduled_tokens=sum(num_sched.values()), finished_req_ids=self.finished_req_ids,
            preempted_req_ids=set(self._preempted),
            finished_requests_needing_kv_transfer=self.get_finished_requests_needing_kv_transfer())
        self.finished_req_ids = set()
        return out

    # ------------------------------------------------------------------ KV hand-off
    def _should_transfer_kv_for_request(self, req_id: str) -> bool:
        return self.need_send_cache

    def _mark_request_for_kv_transfer(self, req_id: str, seq_len: int) -> None:
        if req_id in self.requests_needing_kv_transfer or not self._should_transfer_kv_for_request(req_id):
            return
        blocks = truncate_blocks(self.pool.block_ids(req_id), seq_len, self.block_size)
        self.requests_needing_kv_transfer[req_id] = {"seq_len": seq_len, "block_ids": blocks}

    def get_finished_requests_needing_kv_transfer(self) -> dict[str, dict]:
        """Delivered once: the runner extracts these in its next step; they stay ACTIVE until acked."""
        out, self.requests_needing_kv_transfer = self.requests_needing_kv_transfer, {}
        self.active_kv_transfers.update(out)
        return out

    def _process_kv_transfer_trigger(self, req: Request, new_token_ids: list[int]) -> bool:
        c = self.kv_transfer_criteria
        if not c or req.request_id in self.waiting_for_transfer_free or req.request_id in self.transfer_triggered_requests:
            return False
        if c.get("type") == "prefill_finished":
            # SETTLED tokens, not scheduled ones: with async scheduling num_computed_tokens already counts the prompt's final chunk while
            # that chunk is still in flight, and the shipped length (settled) would be P - 1 one step early
            if self._settled_tokens(req) >= req.num_prompt_tokens:
                self.transfer_triggered_requests.add(req.request_id)
                self._mark_request_for_kv_transfer(req.request_id, self._settled_tokens(req))
        elif c.get("type") == "special_token":
            tok = c.get("token_id")
            if tok is not None and tok in new_token_ids:
                self.transfer_triggered_requests.add(req.request_id)
                exclude = len(new_token_ids) - (new_token_ids.index(tok) + 1)
                self._mark_request_for_kv_transfer(req.request_id, self._settled_tokens(req) - exclude)
        return False      # these criteria never stop the request

    @staticmethod
    def _settled_tokens(req: Request) -> int:
        """Tokens of the request whose KV the steps seen so far have written: what was scheduled minus what is still in flight
        (equal to num_computed_tokens without async scheduling -- the lengths shipped are the same in both modes)."""
        return req.num_computed_tokens - req.num_output_placeholders

    def _free_request(self, req: Request) -> dict | None:
        rid = req.request_id
        self.finished_req_ids.add(rid)
        if self._should_transfer_kv_for_request(rid):
            if rid in self.transfer_triggered_requests:
                if rid in self.active_kv_transfers or rid in self.requests_needing_kv_transfer:
                    self.waiting_for_transfer_free.add(rid)      # triggered earlier, extraction still pending
                    return None
            else:
                self.waiting_for_transfer_free.add(rid)
                self._mark_request_for_kv_transfer(rid, self._settled_tokens(req))
                data = self.requests_needing_kv_transfer.get(rid)
                if data is not None:
                    return {"past_key_values": data["block_ids"],
                            "kv_metadata": {"seq_len": data["seq_len"], "block_ids": data["block_ids"]}}
                return None
        self.pool.free_request(rid)
        self.requests.pop(rid, None)
        self.transfer_triggered_requests.discard(rid)
        return None

    # ------------------------------------------------------------------ output
    def _check_stop(self, req: Request) -> bool:
        sp = req.sampling_params
        if req.num_tokens >= self.max_model_len or len(req.output_token_ids) >= sp.max_tokens:
            req.status = RequestStatus.FINISHED_LENGTH_CAPPED
            return True
        last = req.output_token_ids[-1]
        if not req.ignore_eos and req.eos_token_id is not None and last == req.eos_token_id:
            req.status = RequestStatus.FINISHED_STOPPED
            return True
        if last in (sp.stop_token_ids or ()):
            req.status = RequestStatus.FINISHED_STOPPED
            req.stop_reason = last
            return True
        return False

    def update_from_output(self, scheduler_output: OmniSchedulerOutput, runner_output: OmniModelRunnerOutput) -> list[EngineCoreOutput]:
        outs: list[EngineCoreOutput] = []
        stream_reqs, stream_rows, stream_fin = [], [], []
        for rid, n in scheduler_output.num_scheduled_tokens.items():
            req = self.requests.get(rid)
            if req is None or req.is_finished():
                continue
            idx = runner_output.req_id_to_index.get(rid)
            if idx is None:
                # scheduled, but the runner has nothing for it: a decode row it dropped from a REDONE step (the request had been preempted
                # before the redo: runner._redo_step).  The token it was to sample never existed: give the placeholder back, or the request
                # would wait for it forever
                if req.num_output_placeholders and req.status == RequestStatus.PREEMPTED:
                    req.num_output_placeholders -= 1
                continue
            new_ids = list(runner_output.sampled_token_ids[idx]) if runner_output.sampled_token_ids else []
            if req.num_output_placeholders and new_ids:
                req.num_output_placeholders = max(req.num_output_placeholders - len(new_ids), 0)
            pooled = runner_output.pooler_output[idx] if runner_output.pooler_output else None
            stopped = False
            kept: list[int] = []
            for tok in new_ids:                      # append one by one: a stop cuts the rest (vLLM _update_request_with_output)
                req.output_token_ids.append(tok)
                kept.append(tok)
                if self._check_stop(req):
                    stopped = True
                    break
            if not stopped:
                self._process_kv_transfer_trigger(req, kept)
            kv_params = None
            if stopped:
                if req in self.running:
                    self.running.remove(req)
                elif req in self.waiting:        # preempted while the step that sampled its stop token was in flight
                    self.waiting.remove(req)
                if req.num_output_placeholders:
                    # a surplus step is in flight for it (scheduled before the stop was seen): its blocks are freed now -- the
                    # surplus step's KV write precedes any later owner's in stream order -- and its output is dropped on arrival
                    req.num_computed_tokens -= req.num_output_placeholders
                    req.num_output_placeholders = 0
                kv_params = self._free_request(req)
            new_lp = None
            if kept and runner_output.logprobs is not None and getattr(req.sampling_params, "logprobs", None) is not None:
                new_lp = runner_output.logprobs.slice_request(idx, len(kept), int(req.sampling_params.logprobs))   # omni_ar_scheduler.py:319-321
            if kept or pooled is not None or stopped:
                outs.append(EngineCoreOutput(rid, kept, pooled, req.get_finished_reason(), req.stop_reason, kv_params, new_lp))
            if self.chunk_streamer is not None and pooled is not None and n == 1 and "audio_codes" in pooled:
                stream_reqs.append(req)
                stream_rows.append(pooled["audio_codes"].reshape(-1))
                stream_fin.append(stopped)
        if self.chunk_streamer is not None and stream_reqs:
            import torch
            self.chunk_streamer.send_step(stream_reqs, torch.stack(stream_rows), stream_fin, stage_id=self.stage_id)
        # blocks held for a KV hand-off are released when the runner acks the extraction (omni_ar_scheduler.py:456-479)
        for rid in runner_output.kv_extracted_req_ids or []:
            self.active_kv_transfers.discard(rid)
            if rid in self.waiting_for_transfer_free:
                self.pool.free_request(rid)
                self.requests.pop(rid, None)
                self.transfer_triggered_requests.discard(rid)
                self.waiting_for_transfer_free.discard(rid)
        return outs


class TalkerStageEngine:
    """schedule -> execute_model -> sample_tokens -> update_from_output: the engine-core loop of one AR stage
    (what vLLM's EngineCore.step does around the worker), for tests and stand-alone serving of the talker.

    With ``async_scheduling`` (stage_configs/qwen3_tts.yaml:16) it is vLLM's ``step_with_batch_queue`` at depth 2: step t + 1
    is scheduled and handed to the worker BEFORE step t's output is waited for -- the worker's ``sample_tokens`` returns an
    ``AsyncModelRunnerOutput``-shaped handle (gpu_ar_model_runner.py:641-660) whose ``get_output()`` blocks on the step's host
    copy -- so the host side of step t (copy, bookkeeping, stop checks, this scheduler) runs under the GPU's step t + 1."""

    def __init__(self, worker, scheduler: MI355XARScheduler, async_scheduling: bool | None = None):
        self.worker, self.scheduler = worker, scheduler
        self.async_scheduling = scheduler.async_scheduling if async_scheduling is None else bool(async_scheduling)
        if self.async_scheduling != scheduler.async_scheduling:
            raise ValueError("the engine loop and its scheduler must agree on async_scheduling")
        self.inflight: deque = deque()          # (scheduler_output, runner output or handle), oldest first
        self.max_inflight = 2

    def add_request(self, request: Request) -> None:
        self.scheduler.add_request(request)

    @staticmethod
    def _has_work(so: OmniSchedulerOutput) -> bool:
        return bool(so.total_num_scheduled_tokens or so.finished_req_ids or so.preempted_req_ids
                    or so.finished_requests_needing_kv_transfer)

    def _dispatch(self, so: OmniSchedulerOutput):
        first = self.worker.execute_model(so)
        return first if first is not None else self.worker.sample_tokens(None)

    def step(self) -> list[EngineCoreOutput]:
        so = self.scheduler.schedule()
        if not self.async_scheduling:
            if not self._has_work(so):
                return []
            return self.scheduler.update_from_output(so, self._dispatch(so))
        if self._has_work(so):
            self.inflight.append((so, self._dispatch(so)))
            if len(self.inflight) < self.max_inflight:
                return []                                   # queue not full: schedule the next step before waiting for this one
        if not self.inflight:
            return []
        so0, handle = self.inflight.popleft()
        out = handle.get_output() if hasattr(handle, "get_output") else handle
        return self.scheduler.update_from_output(so0, out)

    def has_work(self) -> bool:
        s = self.scheduler
        return bool(self.inflight or s.has_unfinished_requests() or s.finished_req_ids or s.requests_needing_kv_transfer
                    or s.waiting_for_transfer_free)

    def run(self, max_steps: int = 1 << 30) -> dict[str, list[int]]:
        tokens: dict[str, list[int]] = {}
        for _ in range(max_steps):
            if not self.has_work():
                break
            for o in self.step():
                tokens.setdefault(o.request_id, []).extend(o.new_token_ids)
        return tokens
