"""Platform plugin: the drop-in boundary (SURVEY 8b "Selection").

``engine_args.worker_type: ar`` makes the reference call ``current_omni_platform.get_omni_ar_worker_cls()``
(V/engine/stage_init_utils.py:78-93); the platform is chosen among the built-in probes and exactly one out-of-tree
plugin of the entry-point group ``vllm_omni.platform_plugins`` whose callable returns a platform class qualname or
None (V/platforms/__init__.py:108-148, V/plugins/__init__.py:14-81).  ``register()`` below is that callable;
INTEGRATION.md shows the entry-point stanza.  The class mirrors the ``OmniPlatform`` methods the AR stage touches
(V/platforms/interface.py:21-136, ROCm values V/platforms/rocm/platform.py:14-115).
"""
from __future__ import annotations

import os

import torch

PLATFORM_QUALNAME = "ht_vllm_omni_amd.platform.MI355XOmniPlatform"
AR_WORKER_QUALNAME = "ht_vllm_omni_amd.worker.MI355XARWorker"


class MI355XOmniPlatform:
    device_name = "rocm"
    device_type = "cuda"                 # PyTorch-ROCm exposes HIP devices under torch.cuda
    dist_backend = "nccl"                # == RCCL on ROCm
    device_control_env_var = "HIP_VISIBLE_DEVICES"

    def is_rocm(self) -> bool:
        return True

    def is_cuda(self) -> bool:
        return False

    def is_npu(self) -> bool:            # V/platforms/interface.py:32-35
        return False

    def is_xpu(self) -> bool:
        return False

    @classmethod
    def get_omni_ar_worker_cls(cls) -> str:
        return AR_WORKER_QUALNAME

    @classmethod
    def get_omni_generation_worker_cls(cls) -> str:
        # code2wav one-shot stage: the reference's own generation worker / runner host it; the decoder module it runs is
        # code2wav.Code2WavDecoder (swapped in Qwen3TTSCode2Wav._ensure_speech_tokenizer_loaded)
        return "vllm_omni.worker.gpu_generation_worker.GPUGenerationWorker"

    @classmethod
    def get_default_stage_config_path(cls) -> str:
        # V/platforms/interface.py:53 (ROCm: "vllm_omni/platforms/rocm/stage_configs"): the directory of this package's stage
        # YAMLs -- qwen3_tts.yaml puts stage 0 on MI355XARWorker
        return os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage_configs")

    @classmethod
    def get_profiler_cls(cls) -> str:
        # V/platforms/interface.py:129-136: the reference's torch-profiler wrapper serves this stage unchanged (the worker's
        # profile() also names its phases with the reference's five range names, runner._Range)
        return "vllm_omni.profiler.omni_torch_profiler.OmniTorchProfilerWrapper"

    @classmethod
    def supports_cpu_offload(cls) -> bool:
        # V/platforms/interface.py:113: weights and KV of this stage are device-resident by design (288 GB of HBM per GPU)
        return False

    @classmethod
    def supports_torch_inductor(cls) -> bool:
        return False                     # no tracing compiler on this path: HIP kernels + hipGraph

    @classmethod
    def get_torch_device(cls, local_rank: int | None = None) -> torch.device:
        return torch.device("cuda" if local_rank is None else f"cuda:{local_rank}")

    @classmethod
    def get_device_count(cls) -> int:
        return torch.cuda.device_count()

    @classmethod
    def get_device_version(cls) -> str | None:
        return getattr(torch.version, "hip", None)

    @classmethod
    def synchronize(cls) -> None:
        torch.cuda.synchronize()

    @classmethod
    def get_free_memory(cls, device: torch.device | None = None) -> int:
        return torch.cuda.mem_get_info(device)[0]

    @classmethod
    def set_device_control_env_var(cls, devices) -> None:
        os.environ[cls.device_control_env_var] = str(devices)

    @classmethod
    def unset_device_control_env_var(cls) -> None:
        os.environ.pop(cls.device_control_env_var, None)


def is_mi355x() -> bool:
    if not torch.cuda.is_available() or getattr(torch.version, "hip", None) is None:
        return False
    try:
        return "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    except Exception:  # noqa: BLE001
        return False


def register() -> str | None:
    """Entry point ``vllm_omni.platform_plugins``: activate only on gfx950 (or when forced for tests)."""
    if os.environ.get("HT_OMNI_FORCE_MI355X") == "1" or is_mi355x():
        return PLATFORM_QUALNAME
    return None


def resolve_worker_cls(engine_args: dict) -> dict:
    """What V/engine/stage_init_utils.py:78-93 does with the platform: fill worker_cls from worker_type."""
    if engine_args.get("worker_cls"):
        return engine_args
    wt = engine_args.get("worker_type")
    if wt == "ar":
        engine_args["worker_cls"] = MI355XOmniPlatform.get_omni_ar_worker_cls()
    elif wt == "generation":
        engine_args["worker_cls"] = MI355XOmniPlatform.get_omni_generation_worker_cls()
    elif wt is not None:
        raise ValueError(f"Unknown worker_type: {wt}")
    return engine_args
