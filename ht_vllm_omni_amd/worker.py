"""MI355XARWorker: the worker surface vLLM's executor drives, one process per GPU / TP rank.

Mirrors ``GPUARWorker`` + ``OmniGPUWorkerBase`` (V/worker/gpu_ar_worker.py:22-106, V/worker/base.py:49-156) and the
minimal-backend template ``XPUARWorker`` (V/platforms/xpu/worker/xpu_ar_worker.py:1-15):
  __init__(vllm_config, local_rank, rank, distributed_init_method, ...)  init_device()  load_model()
  determine_available_memory() -> bytes   initialize_from_config(kv_cache_config)   compile_or_warm_up_model()
  execute_model(scheduler_output)   sample_tokens(grammar_output)   profile(is_start, profile_prefix)
``vllm_config`` is either a real ``VllmConfig``-shaped object -- ``model_config.model``, ``cache_config.cache_dtype /
block_size / gpu_memory_utilization / num_gpu_blocks_override``, ``parallel_config.tensor_parallel_size``,
``scheduler_config.max_num_seqs`` (the fields GPUARWorker / GPUARModelRunner read: gpu_ar_worker.py:29-90,
gpu_ar_model_runner.py:118-124) -- which ``config_from_vllm`` flattens, or the flat namespace ``make_config`` returns.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any

import torch

from .config import TalkerDims, get_dims
from .connectors import OmniConnectorFactory, OmniKVTransferManager
from .weights import make_weights, weight_bytes

logger = logging.getLogger("ht_vllm_omni_amd.worker")


def make_config(model: str | TalkerDims = "tts-1.7b", *, kv_cache_dtype: str = "fp8", block_size: int = 16,
                max_num_seqs: int = 64, tensor_parallel_size: int = 1, gpu_memory_utilization: float = 0.9,
                num_gpu_blocks_override: int | None = None, weights: dict | None = None, seed: int = 1234,
                connector: str | None = None, enforce_eager: bool = False, default_sampling_params=None,
                prompt_builder: dict | None = None, model_path: str | None = None, calculate_kv_scales: bool = False,
                async_scheduling: bool = False) -> SimpleNamespace:
    if model_path and isinstance(model, str) and model == "tts-1.7b" and os.path.exists(os.path.join(model_path, "config.json")):
        from .checkpoint import dims_from_hf_config
        model = dims_from_hf_config(os.path.join(model_path, "config.json"))          # dimensions come from the checkpoint
    return SimpleNamespace(model=model, kv_cache_dtype=kv_cache_dtype, block_size=block_size, max_num_seqs=max_num_seqs,
                           tensor_parallel_size=tensor_parallel_size, gpu_memory_utilization=gpu_memory_utilization,
                           num_gpu_blocks_override=num_gpu_blocks_override, weights=weights, seed=seed, connector=connector,
                           enforce_eager=enforce_eager, default_sampling_params=default_sampling_params,
                           prompt_builder=prompt_builder, model_path=model_path, calculate_kv_scales=calculate_kv_scales,
                           async_scheduling=bool(async_scheduling))


_PRESET_HINTS = (("omni", "omni-talker"), ("1.7b", "tts-1.7b"), ("0.6b", "tts-0.6b"))


def _dims_from_model_config(mc) -> str | TalkerDims:
    """What the talker stage is, from ``model_config``: a TalkerDims, a preset name, a checkpoint directory (its config.json
    decides, as in the reference: configuration_qwen3_tts.py), an ``hf_config`` carrying ``talker_config``, else the size
    tag in the model id ("Qwen/Qwen3-TTS-1.7B-Base")."""
    from .config import PRESETS
    model = getattr(mc, "model", None)
    if isinstance(model, TalkerDims):
        return model
    if isinstance(model, str) and model in PRESETS:
        return model
    if isinstance(model, str) and os.path.exists(os.path.join(model, "config.json")):
        from .checkpoint import dims_from_hf_config
        return dims_from_hf_config(os.path.join(model, "config.json"), max_model_len=int(getattr(mc, "max_model_len", 4096) or 4096))
    hf = getattr(mc, "hf_config", None)
    if hf is not None:
        as_dict = hf if isinstance(hf, dict) else (hf.to_dict() if hasattr(hf, "to_dict") else None)
        if as_dict and ("talker_config" in as_dict or "code_predictor_config" in as_dict):
            from .checkpoint import dims_from_hf_config
            return dims_from_hf_config(as_dict, max_model_len=int(getattr(mc, "max_model_len", 4096) or 4096))
    low = str(model).lower()
    for tag, preset in _PRESET_HINTS:
        if tag in low:
            return preset
    raise ValueError(f"MI355XARWorker: cannot tell the talker's dimensions from model_config.model={model!r} "
                     "(pass a checkpoint directory with config.json, an hf_config with talker_config, or a preset name)")


def config_from_vllm(vllm_config) -> SimpleNamespace:
    """Flatten a VllmConfig-shaped object into the namespace this worker reads.  Field map (reference readers):
      model_config.model / .hf_config / .seed / .enforce_eager / .max_model_len      gpu_ar_worker.py:63,78
      cache_config.cache_dtype / .block_size / .gpu_memory_utilization / .num_gpu_blocks_override   gpu_ar_model_runner.py:118-124
      parallel_config.tensor_parallel_size                                                        gpu_ar_worker.py:44-46
      scheduler_config.max_num_seqs / .async_scheduling                          chunk_size_utils.py:5-33, gpu_ar_model_runner.py:641
      additional_config: {connector, prompt_builder, default_sampling_params, weights}            (stage YAML engine_args)
    A namespace that is already flat (make_config) is returned unchanged."""
    if not hasattr(vllm_config, "model_config"):
        return vllm_config
    mc, cc = vllm_config.model_config, getattr(vllm_config, "cache_config", None)
    pc, sc = getattr(vllm_config, "parallel_config", None), getattr(vllm_config, "scheduler_config", None)
    extra = getattr(vllm_config, "additional_config", None) or {}
    if not isinstance(extra, dict):
        extra = dict(vars(extra))
    model = _dims_from_model_config(mc)
    path = mc.model if isinstance(getattr(mc, "model", None), str) and os.path.isdir(mc.model) else extra.get("model_path")
    cache_dtype = str(getattr(cc, "cache_dtype", "auto") or "auto").replace("torch.", "")
    if cache_dtype == "bfloat16":
        cache_dtype = "bf16"
    from . import _lib as L
    if cache_dtype not in L.KV_CODES:
        raise ValueError(f"MI355XARWorker: cache_config.cache_dtype={cache_dtype!r} is not one of {sorted(L.KV_CODES)}")
    return SimpleNamespace(
        model=model, kv_cache_dtype=cache_dtype, block_size=int(getattr(cc, "block_size", 16) or 16),
        max_num_seqs=int(getattr(sc, "max_num_seqs", 64) or 64),
        tensor_parallel_size=int(getattr(pc, "tensor_parallel_size", 1) or 1),
        gpu_memory_utilization=float(getattr(cc, "gpu_memory_utilization", 0.9) or 0.9),
        num_gpu_blocks_override=getattr(cc, "num_gpu_blocks_override", None), weights=extra.get("weights"),
        calculate_kv_scales=bool(getattr(cc, "calculate_kv_scales", False)),
        seed=int(getattr(mc, "seed", 1234) or 0), connector=extra.get("connector"),
        enforce_eager=bool(getattr(mc, "enforce_eager", False)), default_sampling_params=extra.get("default_sampling_params"),
        prompt_builder=extra.get("prompt_builder"), model_path=path,
        # scheduler_config.async_scheduling (stage_configs/qwen3_tts.yaml:16): sample_tokens returns an AsyncStepOutput
        async_scheduling=bool(getattr(sc, "async_scheduling", False)))


@dataclass(frozen=True)
class AttentionSpec:
    """The fields of vLLM's FullAttentionSpec this boundary uses: one layer's share of a KV block ("page") = K and V of
    block_size tokens (+ the int8 cache's per-(token, head) fp32 scales)."""
    block_size: int
    num_kv_heads: int
    head_size: int
    dtype: torch.dtype
    scale_bytes_per_token: int = 0

    @property
    def page_size_bytes(self) -> int:
        return self.block_size * (2 * self.num_kv_heads * self.head_size * torch.empty((), dtype=self.dtype).element_size()
                                  + self.scale_bytes_per_token)


class MI355XARWorker:
    def __init__(self, vllm_config, local_rank: int = 0, rank: int = 0, distributed_init_method: str | None = None,
                 is_driver_worker: bool = True, **_: Any):
        self.raw_vllm_config = vllm_config
        vllm_config = config_from_vllm(vllm_config)
        self.vllm_config = vllm_config
        self.local_rank, self.rank = local_rank, rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.tp_size = int(getattr(vllm_config, "tensor_parallel_size", 1))
        self.dims: TalkerDims = vllm_config.model if isinstance(vllm_config.model, TalkerDims) else get_dims(vllm_config.model)
        self.device: torch.device | None = None
        self.model_runner = None
        self.engine = None
        self._weights = None
        self._profiler = None

    # ---- device + distributed (gpu_ar_worker.py:29-106)
    def init_device(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("MI355XARWorker.init_device: no MI355X visible (torch.cuda unavailable)")
        self.device = torch.device(f"cuda:{self.local_rank}")
        torch.cuda.set_device(self.device)
        if self.tp_size > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # "nccl" IS RCCL on ROCm; OMNI_DIST_BACKEND=gloo only for tests that put two ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("OMNI_DIST_BACKEND", "nccl")
            kw = {"device_id": self.device} if backend == "nccl" else {}
            torch.distributed.init_process_group(backend, init_method=self.distributed_init_method or "env://",
                                                 rank=self.rank, world_size=self.tp_size, **kw)
        torch.cuda.empty_cache()
        self.init_free, self.init_total = torch.cuda.mem_get_info(self.device)

    def load_model(self) -> None:
        """Weights, in this order: an explicit dict (tests), a checkpoint directory / .safetensors file in HF naming
        (``vllm_config.model_path``: the reference's load_weights + hf_to_vllm_mapper, qwen3_tts_talker.py:297-311,
        1569-1590 -- see checkpoint.py), else seeded random weights of the configured shape (benchmarks)."""
        w = getattr(self.vllm_config, "weights", None)
        path = getattr(self.vllm_config, "model_path", None)
        if w is None and path:
            from .checkpoint import check_against_dims, load_talker_checkpoint
            w, self.checkpoint_extras = load_talker_checkpoint(path)
            check_against_dims(w, self.dims)
        self._weights = w if w is not None else make_weights(self.dims, seed=getattr(self.vllm_config, "seed", 1234))

    # ---- memory (base.py:78-156): bytes available for the KV cache after weights and step scratch -- MEASURED, per process
    def determine_available_memory(self) -> int:
        """requested = total * gpu_memory_utilization; available = requested - what THIS process holds once the model is resident and a
        profile run (one full prefill chunk + one decode step at max_num_seqs: every workspace, code object and allocator pool) has gone
        through -- the reference's algorithm (V/worker/base.py:78-156: `memory_profiling` around `profile_run`, then per-PID memory, so that
        stages initialising concurrently on one device do not book each other's bytes; gpu_memory_utils.py:69-124).  Per-PID bytes come
        from the KFD's counters (gpu_memory.process_gpu_memory); where they are not readable, from the hipMemGetInfo delta against the
        init_device snapshot (counts other processes' allocations in between too: conservative).  The engine is built here on a probe
        cache just large enough for the profile run and keeps its weights; initialize_from_config resizes the cache (engine.resize_kv_cache).
        `kv_cache_memory_bytes` in the config short-cuts the arithmetic, not the profile run (base.py:103-110)."""
        from .gpu_memory import kv_cache_budget, process_gpu_memory
        cfg = self.vllm_config
        if self.engine is None:
            tokens = int(getattr(cfg, "max_num_batched_tokens", 8192) or 8192)
            probe_blocks = (tokens + cfg.block_size - 1) // cfg.block_size + 2 * int(cfg.max_num_seqs) + 2
            self._build_engine(probe_blocks)
            self.engine.profile_run(tokens, int(cfg.max_num_seqs))
        probe_kv = self.engine.kv_cache_bytes()
        explicit = getattr(cfg, "kv_cache_memory_bytes", None)
        if explicit:
            self.available_kv_cache_memory_bytes = int(explicit)
            return int(explicit)
        free_now, _ = self._mem_get_info()
        proc = self._process_memory()
        self.memory_accounting = "process-scoped (KFD)" if proc is not None else "snapshot delta"
        if proc is None:
            proc = max(self.init_free - free_now, 0)
        self.process_memory_bytes = int(proc)
        self.available_kv_cache_memory_bytes = kv_cache_budget(self.init_total, float(getattr(cfg, "gpu_memory_utilization", 0.9)), int(proc), probe_kv)
        logger.info("KV cache budget: %.2f GiB (%s: this process holds %.2f GiB, of it %.2f GiB probe cache; total %.2f GiB)",
                    self.available_kv_cache_memory_bytes / 2 ** 30, self.memory_accounting, proc / 2 ** 30, probe_kv / 2 ** 30,
                    self.init_total / 2 ** 30)
        return int(self.available_kv_cache_memory_bytes)

    def _mem_get_info(self):
        return torch.cuda.mem_get_info(self.device)

    def _process_memory(self):
        from .gpu_memory import process_gpu_memory
        torch.cuda.synchronize(self.device)
        return process_gpu_memory()

    # ---- KV cache spec (vLLM Worker.get_kv_cache_spec -> {layer name: KVCacheSpec}; the executor sizes the cache from the
    # specs' page sizes and hands the result back as a KVCacheConfig, V/worker/base.py:78-156, gpu_ar_model_runner.py:118-124)
    def get_kv_cache_spec(self) -> dict[str, "AttentionSpec"]:
        d, cfg = self.dims, self.vllm_config
        store = {"bf16": torch.bfloat16, "auto": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16, "half": torch.float16,
                 "fp8": torch.uint8, "fp8_e4m3": torch.uint8, "int8": torch.int8}[cfg.kv_cache_dtype]
        hkv = max(d.kv_heads // self.tp_size, 1)
        spec = AttentionSpec(block_size=int(cfg.block_size), num_kv_heads=hkv, head_size=d.head_dim, dtype=store,
                             scale_bytes_per_token=2 * hkv * 4 if cfg.kv_cache_dtype == "int8" else 0)
        return {f"model.layers.{l}.self_attn.attn": spec for l in range(d.layers)}

    def kv_bytes_per_block(self) -> int:
        return sum(s.page_size_bytes for s in self.get_kv_cache_spec().values())

    def _build_engine(self, num_blocks: int) -> None:
        from .engine import TalkerEngine
        cfg = self.vllm_config
        # tensor-parallel ranks: the all-reduce of the step is a kernel of the step (peer-mapped buffers over hipIpc, fused with
        # the residual add: tp_comm.py) -- checked against RCCL on random data first, every rank falls back to RCCL
        # all-reduces between the phase calls if any rank disagrees (the reference's group: gpu_ar_worker.py:69-75)
        self.peer_allreduce = None
        if self.tp_size > 1:       # dense and sparse-MoE backbones alike (MoE: omni_moe_experts_resid leaves the rank's partial)
            from .tp_comm import setup_peer_allreduce
            self.peer_allreduce = setup_peer_allreduce(self.dims.hidden, min(cfg.max_num_seqs, 64), self.rank, self.tp_size,
                                                       log=logger.info)
        self.engine = TalkerEngine(self.dims, self._weights, kv_dtype=cfg.kv_cache_dtype, num_blocks=int(num_blocks),
                                   block_size=cfg.block_size, max_batch=cfg.max_num_seqs, device=str(self.device),
                                   tp_rank=self.rank, tp_size=self.tp_size, peer_allreduce=self.peer_allreduce,
                                   calculate_kv_scales=bool(getattr(cfg, "calculate_kv_scales", False)))
        # ... and the all-reduce INSIDE the backbone's persistent launches only after four scratch steps both ways agreed on every rank
        self.tp_backbone_chain = False
        if self.peer_allreduce is not None:
            from .tp_comm import check_backbone_chain
            self.tp_backbone_chain = check_backbone_chain(self.engine, log=logger.info)
        self._embed_table = self._weights["embed"]
        self._weights = None

    def initialize_from_config(self, kv_cache_config: Any = None) -> None:
        from .runner import MI355XARModelRunner
        cfg = self.vllm_config
        # a KVCacheConfig-shaped object: .num_blocks, or .kv_cache_tensors = [{size, shared_by: [layer names]}] (one per layer in
        # vLLM's default layout): the smallest tensor bounds the block count
        nb = getattr(kv_cache_config, "num_blocks", None) or cfg.num_gpu_blocks_override
        tensors = getattr(kv_cache_config, "kv_cache_tensors", None)
        if nb is None and tensors:
            specs = self.get_kv_cache_spec()
            per = []
            for tns in tensors:
                names = list(getattr(tns, "shared_by", None) or [])
                page = sum(specs[n].page_size_bytes for n in names if n in specs) or next(iter(specs.values())).page_size_bytes
                per.append(int(getattr(tns, "size")) // page)
            nb = min(per)
        if nb is None:
            nb = max(self.determine_available_memory() // self.kv_bytes_per_block(), 2)
        if self.engine is not None:          # built (and measured) by determine_available_memory on a probe cache: give it the real one
            self.engine.resize_kv_cache(int(nb))
        else:
            self._build_engine(int(nb))
        embed_table = self._embed_table
        conn = OmniConnectorFactory.create_connector(cfg.connector) if getattr(cfg, "connector", None) else None
        # Qwen3-Omni talker: config.prompt_builder = {"weights": {text, hidden, codec_embed}, "ids": OmniPromptIds | dict}
        pb_cfg, builder = getattr(cfg, "prompt_builder", None), None
        if pb_cfg:
            from .prompt_builder_omni import OmniPromptIds, OmniTalkerPromptBuilder
            ids = pb_cfg["ids"] if isinstance(pb_cfg["ids"], OmniPromptIds) else OmniPromptIds.from_dict(pb_cfg["ids"])
            pw = dict(pb_cfg["weights"])
            pw.setdefault("codec_embed", embed_table)        # talker.embed_input_ids = the talker's own codec table
            builder = OmniTalkerPromptBuilder(pw, ids, self.device)
        self.model_runner = MI355XARModelRunner(self.engine, kv_transfer=OmniKVTransferManager(conn),
                                                use_graphs=not getattr(cfg, "enforce_eager", False), prompt_builder=builder,
                                                async_scheduling=bool(getattr(cfg, "async_scheduling", False)))

    def compile_or_warm_up_model(self) -> None:
        # sampling parameters are per-request device rows read inside the captured step (a request without any gets the
        # stage's default_sampling_params, stage_configs/qwen3_tts.yaml:27-34, on admission): nothing to bake in here
        from .payloads import SamplingParams
        self.model_runner.default_sampling = getattr(self.vllm_config, "default_sampling_params", None) or SamplingParams()
        self.model_runner.capture_graphs()

    # ---- step (executor RPC targets)
    def execute_model(self, scheduler_output, intermediate_tensors=None):
        return self.model_runner.execute_model(scheduler_output, intermediate_tensors)

    def sample_tokens(self, grammar_output=None):
        return self.model_runner.sample_tokens(grammar_output)

    # ---- profiling (base.py:49-76; range names of gpu_ar_model_runner.py:138,293,314,454,514 via torch.profiler)
    def profile(self, is_start: bool = True, profile_prefix: str | None = None):
        from .runner import _Range
        _Range.enabled = bool(is_start) or os.environ.get("OMNI_PROFILE_RANGES") == "1"      # the reference's five phase names as ranges
        if is_start:
            self._profiler = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU,
                                                                torch.profiler.ProfilerActivity.CUDA])
            self._profiler.__enter__()
            return None
        if self._profiler is None:
            return None
        self._profiler.__exit__(None, None, None)
        path = f"{profile_prefix or 'stage0'}_rank{self.rank}.json"
        self._profiler.export_chrome_trace(path)
        self._profiler = None
        return path

    def check_health(self) -> None:
        return None

    def shutdown(self) -> None:
        self.model_runner = None
        self.engine = None
        if torch.distributed.is_initialized() and self.tp_size > 1:
            torch.distributed.destroy_process_group()
