"""Process-scoped GPU memory accounting for the worker's KV-cache sizing (reference: V/worker/base.py:78-156,
V/worker/gpu_memory_utils.py:69-124 -- `requested = total * gpu_memory_utilization`, `available = requested - memory of THIS process`, so
that stages initialising concurrently on one device do not book each other's bytes; NVML there, the KFD's per-process counters here).

ROCm exposes what a process holds on every GPU node under /sys/class/kfd/kfd/proc/<pid>/vram_<gpu_id> (bytes).  One engine process drives
one device, so the sum over the nodes is its footprint on that device (a peer-mapped all-reduce buffer of another rank is that rank's).
Where the files are not readable (containers without the KFD sysfs) the caller falls back to the before / after snapshot of
hipMemGetInfo, which also counts what OTHER processes allocated in between -- conservative, never optimistic."""
from __future__ import annotations

import glob
import os


def process_gpu_memory(pid: int | None = None, root: str = "/sys/class/kfd/kfd/proc") -> int | None:
    """Bytes of VRAM the process holds according to the KFD, or None when the per-process counters are not available."""
    d = os.path.join(root, str(pid if pid is not None else os.getpid()))
    total, seen = 0, False
    for f in glob.glob(os.path.join(d, "vram_*")):
        try:
            total += int(open(f).read().strip())
            seen = True
        except (OSError, ValueError):
            continue
    return total if seen else None


def kv_cache_budget(total: int, utilization: float, process_bytes: int, probe_kv_bytes: int) -> int:
    """requested - (what this process holds besides the KV cache): the probe engine's own small cache is given back."""
    requested = int(total * float(utilization))
    return max(requested - max(process_bytes - probe_kv_bytes, 0), 0)
