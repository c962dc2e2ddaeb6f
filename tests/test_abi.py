"""The C-ABI library loads and exports every symbol include/omni_talker.h declares; ctypes mirrors of the
structs have the C layout (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "omni_talker.h")


def header_functions():
    names = set()
    for h in (HDR, os.path.join(ROOT, "include", "omni_codec.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(omni_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_header_symbols_exported_and_bound():
    from ht_vllm_omni_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.omni_abi_version() == 5
    assert lib.omni_last_error() is not None


def test_struct_layout_matches_c(tmp_path):
    from ht_vllm_omni_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(omni_talker_desc),sizeof(omni_step_io),sizeof(omni_layer_weights),offsetof(omni_talker_desc,scratch_bytes),'
                   'offsetof(omni_step_io,advance),offsetof(omni_talker_desc,k_cache),sizeof(omni_ar_peers),'
                   'offsetof(omni_ar_peers,tile_flags));return 0;}\n' % HDR)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[0] == C.sizeof(_lib.TalkerDesc)
    assert got[1] == C.sizeof(_lib.StepIO)
    assert got[2] == C.sizeof(_lib.LayerWeights)
    assert got[3] == _lib.TalkerDesc.scratch_bytes.offset
    assert got[4] == _lib.StepIO.advance.offset
    assert got[5] == _lib.TalkerDesc.k_cache.offset
    assert got[6] == C.sizeof(_lib.ArPeers)
    assert got[7] == _lib.ArPeers.tile_flags.offset


def test_argument_errors_do_not_abort():
    """Bad arguments come back as error codes + message (host-side checks run before any launch)."""
    from ht_vllm_omni_amd import _lib
    lib = _lib.load()
    rc = lib.omni_gemm_bf16(None, 0, None, None, None, 4, 16, 32, 0, None, None)
    assert rc == -1 and b"null" in lib.omni_last_error()
    rc = lib.omni_rmsnorm(None, None, None, None, None, 1, 8, 1e-6, None)
    assert rc == -1
    with pytest.raises(_lib.OmniError):
        _lib.check(rc, "omni_rmsnorm")
    assert lib.omni_talker_scratch_bytes(None) == -1


def test_engine_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ht_vllm_omni_amd import _lib
    from ht_vllm_omni_amd.config import get_dims
    from ht_vllm_omni_amd.engine import TalkerEngine
    with pytest.raises(_lib.OmniError):
        TalkerEngine(get_dims("tiny"), {})
    from ht_vllm_omni_amd import ops
    with pytest.raises(_lib.OmniError):
        ops.rmsnorm(torch.zeros(1, 8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16), 1e-6)


def test_debug_header_symbols_exported():
    """include/omni_talker_debug.h (diagnostic hooks, outside the boundary) = what libomni_talker_debug.so exports on top
    of the full ABI; the product library exports none of them."""
    from ht_vllm_omni_amd import _lib
    prod = _lib.load()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "omni_talker_debug.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(omni_debug_[a-z0-9_]+)\s*\(", txt)))
    assert len(names) >= 6
    with _lib.debug_library() as lib:
        for n in names:
            assert hasattr(lib, n), f"{n} declared in omni_talker_debug.h but not exported by the debug library"
        for n in header_functions():
            assert hasattr(lib, n), f"{n} missing from the debug library"
    assert _lib.load() is prod
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    assert "omni_debug_" not in out, "diagnostic hooks leaked into the product library"


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """_lib refuses a library whose digest stamp does not match csrc/ (the git-ignored .so may be older)."""
    from ht_vllm_omni_amd import _lib
    stamp = _lib.LIB_PATH + ".digest"
    good = open(stamp).read()
    try:
        open(stamp, "w").write("0" * 64 + "\n")
        with pytest.raises(_lib.OmniError, match="stale"):
            _lib._open(_lib.LIB_PATH, False)
    finally:
        open(stamp, "w").write(good)
    _lib._open(_lib.LIB_PATH, False)


def test_torch_library_ops_are_registered_and_have_no_cpu_kernel():
    """SURVEY 8b / north_star: the per-op entry points as PyTorch custom ops (`TORCH_LIBRARY(mi355x_omni)` in the survey's
    words; registered through torch.library over the same C symbols).  On a host without a GPU the ops exist, and a CPU
    tensor fails loudly -- there is no CPU kernel to fall back to."""
    import pytest
    import torch
    from ht_vllm_omni_amd import torch_ops as T
    for n in ("rmsnorm_residual_", "qknorm_rope_kvwrite_", "paged_attn_decode", "paged_attn_prefill", "silu_mul", "skinny_gemm",
              "lmhead_mask", "topk_sample", "allreduce_oneshot_", "decode_step_"):
        assert n in T.OPS and hasattr(torch.ops.mi355x_omni, n), n
    # the whole decode step is an op too: the engine's per-step call goes through it, its schema names what it mutates
    sch = str(torch.ops.mi355x_omni.decode_step_.default._schema)
    assert "Tensor(a!) out_record" in sch and "Tensor(f!)[] kv_caches" in sch
    import inspect
    from ht_vllm_omni_amd.engine import TalkerEngine
    assert "torch.ops.mi355x_omni.decode_step_" in inspect.getsource(TalkerEngine.decode_step)
    assert "Tensor(b!) out" in str(torch.ops.mi355x_omni.rmsnorm_residual_.default._schema)      # caller-allocated output: graph-safe
    with pytest.raises(NotImplementedError):
        torch.ops.mi355x_omni.silu_mul(torch.zeros(2, 8, dtype=torch.bfloat16))
    with pytest.raises(NotImplementedError):
        torch.ops.mi355x_omni.topk_sample(torch.zeros(2, 8), None, None, True, 1.0, 0, 1.0, 1.0, 0, 1, 0, False)


def test_tile_flag_block_matches_the_kernels_indexing():
    """tp_comm lays the per-tile arrival flags out behind the control block; chain_gemm.cuh indexes them [source rank][CH_AR_TILES]: the two
    sizes must agree (a host that allocated less would let a rank write past a peer's control block)."""
    import re
    from ht_vllm_omni_amd import tp_comm
    src = open(os.path.join(os.path.dirname(HDR), "..", "ht_vllm_omni_amd", "csrc", "chain_gemm.cuh")).read()
    tiles = int(re.search(r"#define\s+CH_AR_TILES\s+(\d+)", src).group(1))
    world = int(re.search(r"#define\s+CH_AR_MAX_WORLD\s+(\d+)", src).group(1))
    assert tp_comm.TILE_BYTES == world * tiles * 4
    assert tp_comm.CTL_BYTES == tp_comm.TILE_OFF + tp_comm.TILE_BYTES and tp_comm.TILE_OFF >= tp_comm.ERROR_OFF + 4
    assert tiles >= 256, "the o_proj / down_proj tile grid of the 64-row stage set: up to 128 column tiles x 2 row tiles"

