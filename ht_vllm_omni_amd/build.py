"""Build the HIP libraries in-tree with hipcc for gfx950 (cross-compiles without a GPU).

  libomni_talker.so        the product: HIP kernels + the C-ABI of include/omni_talker.h, policy knobs compiled in
  libomni_talker_debug.so  the same sources with -DOMNI_DEBUG_HOOKS (run-time policy knobs, omni_debug_* setters) plus
                           csrc/debug.hip (launch / memory-system probes): scripts/ and the tile-sweep tests only

Both are git-ignored but travel with the repo snapshot.  `sources_digest()` is stamped into a side file next to each
library so that _lib.load() can refuse a library that is older than csrc/ (ADVICE r1).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libomni_talker.so")
LIB_DEBUG = os.path.join(HERE, "libomni_talker_debug.so")
SOURCES = ["capi.hip", "gemm.hip", "gemm_prefill.hip", "norm.hip", "rope_kv.hip", "paged_attn.hip", "prefill_attn.hip", "moe.hip",
           "sampler.hip", "cp_chain.hip", "bb_chain.hip", "moe_chain.hip", "allreduce.hip", "codec.hip", "conv_unit.hip"]
# debug library only: the probes, and the losing A/B arms of the backbone chain (rounds 3 and 4; no product knob reaches them)
DEBUG_ONLY = ["debug.hip", "bb_engine.hip", "bb_all.hip"]
# per-file extra flags
EXTRA = {"debug.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


def _headers() -> list[str]:
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".cuh", ".h"))]
    inc = os.path.normpath(os.path.join(HERE, "..", "include"))      # every ABI header: a struct change in any of them is a new library
    hdrs += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    return hdrs


def sources_digest(debug: bool = False) -> str:
    """sha256 over every file the library is built from (+ the flags): what _lib.load() compares the stamp with."""
    h = hashlib.sha256()
    names = [s for s in SOURCES + (DEBUG_ONLY if debug else []) if os.path.exists(os.path.join(CSRC, s))]
    for f in [os.path.join(CSRC, s) for s in names] + _headers():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    return h.hexdigest()


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose=False):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
    if verbose and r.stderr:
        print(r.stderr, file=sys.stderr)


def _build_one(lib: str, sources: list[str], objdir: str, extra: list[str], force: bool, verbose: bool, digest: str) -> str:
    hdrs = _headers()
    os.makedirs(objdir, exist_ok=True)
    sources = [s for s in sources if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *FLAGS, *extra, *EXTRA.get(s, []), "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in sources]
    stamp = lib + ".digest"
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or jobs or _stale(lib, objs) or old != digest:
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], verbose)
        with open(stamp, "w") as f:
            f.write(digest + "\n")
    return lib


def build(force: bool = False, verbose: bool = False, debug: bool = True) -> str:
    lib = _build_one(LIB, SOURCES, os.path.join(HERE, "build"), [], force, verbose, sources_digest(False))
    if debug:
        _build_one(LIB_DEBUG, SOURCES + DEBUG_ONLY, os.path.join(HERE, "build", "debug"), ["-DOMNI_DEBUG_HOOKS"], force, verbose,
                   sources_digest(True))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
