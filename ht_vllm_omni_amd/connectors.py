"""OmniConnector surface kept for the talker stage (put/get/cleanup/health/close) + KV extraction.

Mirrors, for this path only:
  OmniConnectorBase / factory      V/distributed/omni_connectors/connectors/base.py:12-104, factory.py:24-111
  SharedMemoryConnector            V/distributed/omni_connectors/connectors/shm_connector.py:17-151
  wire format                      V/distributed/omni_connectors/utils/serialization.py:83-95,263-275 -- tensors
                                   are viewed as uint8 and travel as {dtype, shape, data}; rebuilt with
                                   getattr(torch, dtype) so bf16 / fp8 / int8 KV blocks cross stages as raw bytes
  KV extraction                    V/distributed/omni_connectors/kv_transfer_manager.py:160-361 (key format,
                                   3 retries with exponential back-off), utils/kv_utils.py:13-85
Error convention (SHM impl): return False / None and log, never raise to the caller.
"""
from __future__ import annotations

import fcntl
import logging
import os
import pickle
import time
from abc import ABC, abstractmethod
from multiprocessing import shared_memory
from typing import Any, Callable

import torch

logger = logging.getLogger("ht_vllm_omni_amd.connectors")

_MARK = "__omni_tensor__"


def _pack(obj: Any) -> Any:
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu().contiguous()
        return {_MARK: True, "dtype": str(t.dtype).replace("torch.", ""), "shape": list(t.shape),
                "data": t.view(torch.uint8).numpy().tobytes() if t.numel() else b""}
    if isinstance(obj, dict):
        return {k: _pack(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_pack(v) for v in obj)
    return obj


def _unpack(obj: Any) -> Any:
    if isinstance(obj, dict):
        if obj.get(_MARK):
            dt = getattr(torch, obj["dtype"])
            if not obj["data"]:
                return torch.empty(obj["shape"], dtype=dt)
            raw = torch.frombuffer(bytearray(obj["data"]), dtype=torch.uint8)
            return raw.view(dt).reshape(obj["shape"])
        return {k: _unpack(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_unpack(v) for v in obj)
    return obj


class OmniConnectorBase(ABC):
    supports_raw_data: bool = False

    @abstractmethod
    def put(self, from_stage: str, to_stage: str, put_key: str, data: Any) -> tuple[bool, int, dict[str, Any] | None]: ...

    @abstractmethod
    def get(self, from_stage: str, to_stage: str, get_key: str, metadata=None) -> tuple[Any, int] | None: ...

    @abstractmethod
    def cleanup(self, request_id: str) -> None: ...

    @abstractmethod
    def health(self) -> dict[str, Any]: ...

    @abstractmethod
    def close(self) -> None: ...

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def serialize_obj(obj: Any) -> bytes:
        return pickle.dumps(_pack(obj), protocol=pickle.HIGHEST_PROTOCOL)

    @staticmethod
    def deserialize_obj(data: bytes) -> Any:
        return _unpack(pickle.loads(data))

    @staticmethod
    def _make_key(key: str, from_stage: str, to_stage: str, separator: str = "@") -> str:
        return f"{key}{separator}{from_stage}_{to_stage}"


class SharedMemoryConnector(OmniConnectorBase):
    """POSIX shared memory segment per key, flock around the segment (shm_connector.py:54-58,82-85)."""

    def __init__(self, config: dict[str, Any] | None = None):
        self.config = config or {}
        self._metrics = {"puts": 0, "gets": 0, "bytes_transferred": 0, "shm_writes": 0}
        self._closed = False

    @staticmethod
    def _lock_path(name: str) -> str:
        return f"/dev/shm/shm_{name}_lockfile.lock"

    def put(self, from_stage, to_stage, put_key, data):
        try:
            payload = self.serialize_obj(data)
            size = len(payload)
            with open(self._lock_path(put_key), "wb+") as lockf:
                fcntl.flock(lockf, fcntl.LOCK_EX)
                try:
                    seg = shared_memory.SharedMemory(name=put_key, create=True, size=max(size, 1))
                except FileExistsError:
                    old = shared_memory.SharedMemory(name=put_key)
                    old.close()
                    old.unlink()
                    seg = shared_memory.SharedMemory(name=put_key, create=True, size=max(size, 1))
                seg.buf[:size] = payload
                seg.close()
                fcntl.flock(lockf, fcntl.LOCK_UN)
            self._metrics["puts"] += 1
            self._metrics["shm_writes"] += 1
            self._metrics["bytes_transferred"] += size
            return True, size, {"shm": {"name": put_key, "size": size}, "size": size}
        except Exception as e:  # noqa: BLE001  (never raise to the caller)
            logger.error("SharedMemoryConnector put failed for req %s: %s", put_key, e)
            return False, 0, None

    def get(self, from_stage, to_stage, get_key, metadata=None):
        if isinstance(metadata, dict) and get_key in metadata:
            metadata = metadata[get_key]
        name, size = get_key, None
        if isinstance(metadata, dict) and "shm" in metadata:
            name, size = metadata["shm"]["name"], int(metadata["shm"]["size"])
        lock = self._lock_path(name)
        try:
            with open(lock, "rb+") as lockf:
                fcntl.flock(lockf, fcntl.LOCK_EX)
                seg = shared_memory.SharedMemory(name=name)
                n = size if size is not None else seg.size
                raw = bytes(seg.buf[:n])
                seg.close()
                seg.unlink()                       # consumer reads and unlinks
                fcntl.flock(lockf, fcntl.LOCK_UN)
            obj = self.deserialize_obj(raw)
            self._metrics["gets"] += 1
            if os.path.exists(lock):
                os.remove(lock)
            return obj, n
        except Exception as e:  # noqa: BLE001
            logger.debug("SharedMemoryConnector get(%s) found nothing: %s", get_key, e)
            return None

    def cleanup(self, request_id: str) -> None:
        for name in (request_id,):
            try:
                seg = shared_memory.SharedMemory(name=name)
                seg.close()
                seg.unlink()
            except Exception:  # noqa: BLE001
                pass
            if os.path.exists(self._lock_path(name)):
                os.remove(self._lock_path(name))

    def health(self) -> dict[str, Any]:
        return {"status": "closed" if self._closed else "healthy", **self._metrics}

    def close(self) -> None:
        self._closed = True


class InProcConnector(OmniConnectorBase):
    """Dict store (what the reference's tests use as MockConnector, T/distributed/omni_connectors/test_kv_flow.py:15-31)."""

    def __init__(self, config: dict[str, Any] | None = None):
        self.store: dict[str, bytes] = {}

    def put(self, from_stage, to_stage, put_key, data):
        payload = self.serialize_obj(data)
        self.store[put_key] = payload
        return True, len(payload), None

    def get(self, from_stage, to_stage, get_key, metadata=None):
        raw = self.store.pop(get_key, None)
        return None if raw is None else (self.deserialize_obj(raw), len(raw))

    def cleanup(self, request_id):
        self.store.pop(request_id, None)

    def health(self):
        return {"status": "healthy", "keys": len(self.store)}

    def close(self):
        self.store.clear()


class OmniConnectorFactory:
    _registry: dict[str, Callable[[dict[str, Any]], OmniConnectorBase]] = {}

    @classmethod
    def register_connector(cls, name: str, ctor: Callable[[dict[str, Any]], OmniConnectorBase]) -> None:
        if name in cls._registry:
            raise ValueError(f"connector {name!r} already registered")
        cls._registry[name] = ctor

    @classmethod
    def create_connector(cls, name: str, config: dict[str, Any] | None = None) -> OmniConnectorBase:
        if name not in cls._registry:
            raise ValueError(f"unknown connector {name!r}; registered: {sorted(cls._registry)}")
        return cls._registry[name](config or {})

    @classmethod
    def list_registered_connectors(cls) -> list[str]:
        return sorted(cls._registry)


OmniConnectorFactory.register_connector("SharedMemoryConnector", SharedMemoryConnector)
OmniConnectorFactory.register_connector("InProcConnector", InProcConnector)


# ------------------------------------------------------------------ KV extraction for downstream stages
def normalize_layer_kv(layer_kv):
    """(key_blocks, value_blocks) from [2,nb,bs,h,d], [nb,2,bs,h,d] or a (K, V) tuple; None when invalid
    (utils/kv_utils.py:13-85)."""
    if isinstance(layer_kv, torch.Tensor):
        if layer_kv.ndim >= 3 and layer_kv.shape[0] == 2:
            return layer_kv[0], layer_kv[1]
        if layer_kv.ndim >= 3 and layer_kv.shape[1] == 2:
            return layer_kv[:, 0], layer_kv[:, 1]
        return None
    if isinstance(layer_kv, tuple) and len(layer_kv) == 2 and all(isinstance(t, torch.Tensor) for t in layer_kv):
        k, v = layer_kv
        return (k, v) if k.ndim >= 2 and v.ndim >= 2 else None
    return None


class OmniKVTransferManager:
    """Ships a finished request's KV blocks through a connector (kv_transfer_manager.py:160-361)."""

    def __init__(self, connector: OmniConnectorBase | None, from_stage: str = "0", to_stage: str = "1",
                 max_retries: int = 3, backoff_s: float = 0.05):
        self.connector, self.from_stage, self.to_stage = connector, from_stage, to_stage
        self.max_retries, self.backoff_s = max_retries, backoff_s

    def extract_kv_cache(self, req_id: str, block_ids: list[int], seq_len: int, kv_caches: list, block_size: int,
                         cache_dtype: str, custom_metadata: dict | None = None, kv_scales: list | None = None,
                         tp_rank: int = 0, tp_size: int = 1) -> dict | None:
        """kv_scales: the per-(token, kv head) fp32 scale tensors [2, nb, bs, h] of an int8 cache -- they travel as
        ``layer_blocks["key_scales" / "value_scales"]`` (the reference's payload carries raw bytes + a dtype string only,
        kv_transfer_manager.py:290-301: without them the receiver cannot dequantise).  Under tensor parallelism every rank
        holds a slice of the KV heads: metadata says which (``tp_rank``, ``tp_size``)."""
        num_layers = len(kv_caches)
        key_cache: list = [None] * num_layers
        value_cache: list = [None] * num_layers
        key_scales: list = [None] * num_layers
        value_scales: list = [None] * num_layers
        for li, layer_kv in enumerate(kv_caches):
            pair = normalize_layer_kv(layer_kv)
            if pair is None:
                continue
            kb, vb = pair
            mx = min(kb.shape[0], vb.shape[0]) - 1
            valid = [b for b in block_ids if 0 <= b <= mx]
            if not valid:
                continue
            fk, fv = kb[valid].flatten(0, 1), vb[valid].flatten(0, 1)
            if seq_len < fk.shape[0]:
                fk, fv = fk[:seq_len], fv[:seq_len]
            key_cache[li] = fk.detach().cpu().contiguous()
            value_cache[li] = fv.detach().cpu().contiguous()
            if kv_scales is not None and kv_scales[li] is not None:
                sk, sv = kv_scales[li][0][valid].flatten(0, 1), kv_scales[li][1][valid].flatten(0, 1)
                key_scales[li] = sk[:seq_len].detach().cpu().contiguous()
                value_scales[li] = sv[:seq_len].detach().cpu().contiguous()
        if not any(k is not None for k in key_cache):
            return None
        blocks = {"key_cache": key_cache, "value_cache": value_cache}
        if kv_scales is not None:
            blocks.update({"key_scales": key_scales, "value_scales": value_scales})
        meta = {"block_size": block_size, "num_layers": num_layers, "dtype": str(cache_dtype), "seq_len": seq_len,
                **(custom_metadata or {})}
        if tp_size > 1:
            meta.update({"tp_rank": tp_rank, "tp_size": tp_size})
        return {"request_id": req_id, "layer_blocks": blocks, "block_ids": block_ids, "metadata": meta}

    def handle_finished_requests_kv_transfer(self, finished_reqs: dict[str, dict], kv_caches: list, block_size: int,
                                             cache_dtype: str, request_id_resolver=None, kv_scales: list | None = None,
                                             tp_rank: int = 0, tp_size: int = 1) -> list[str]:
        """Returns the request ids whose blocks the scheduler may now free (kv_extracted_req_ids)."""
        done: list[str] = []
        if not finished_reqs or self.connector is None:
            return list(finished_reqs or {})
        for req_id, data in finished_reqs.items():
            try:
                payload = self.extract_kv_cache(req_id, list(data.get("block_ids", [])), int(data.get("seq_len", 0)),
                                                kv_caches, block_size, cache_dtype, data.get("custom_metadata"),
                                                kv_scales=kv_scales, tp_rank=tp_rank, tp_size=tp_size)
                if payload is not None:
                    gid = request_id_resolver(req_id) if request_id_resolver else req_id
                    payload["request_id"] = gid                      # the resolved transfer id (kv_transfer_manager.py:316)
                    key = f"omni_{self.from_stage}_to_{self.to_stage}_kv_cache_{gid}"
                    if tp_size > 1:
                        key += f"_tp{tp_rank}"                       # every rank ships its own KV-head slice
                    ok = False
                    for attempt in range(self.max_retries):
                        ok, _, _ = self.connector.put(self.from_stage, self.to_stage, key, payload)
                        if ok:
                            break
                        time.sleep(self.backoff_s * (2 ** attempt))
                    if not ok:
                        logger.error("KV transfer of %s failed after %d attempts", req_id, self.max_retries)
            except Exception as e:  # noqa: BLE001  (per-request failures are logged and swallowed)
                logger.warning("KV extraction failed for %s: %s", req_id, e)
            done.append(req_id)
        return done
