"""Build rules the GPU found the hard way, checked on the CPU by cross-compiling to gfx950 assembly.

Rule 1 (round 4): in the attention kernels no packed fp32 operation (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) has a destination
pair that overlaps a source pair whose HIGH dword the LOW half reads through op_sel (nor a destination that overlaps a source pair
by one register).  The decode attention's QK product once packed two
heads per FMA with the K element broadcast by op_sel; hipcc let the destination overlap the broadcast source, and that build --
bit-stable alone on the GPU, parity-green -- produced garbage in the LOW half in ~10 % of the launches beside another process's
kernels (tests/test_gpu_colocation.py; scripts/probes/attn_coloc_probe.py).  The QK product now pairs adjacent dimensions (no
selector at all); this test keeps the pattern from coming back through a later edit or a compiler's register allocation.
(The mirror pattern -- the HIGH half reading the low dword of an overlapping source, op_sel_hi = 0 -- is what hipcc makes of
scalar-times-pair code all over these kernels since round 2; it has never shown the effect, alone or co-located, and is allowed.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ht_vllm_omni_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(name, tmp_path):
    out = os.path.join(tmp_path, name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", os.path.join(CSRC, name), "-o", out], check=True, capture_output=True, timeout=600)
    return open(out).read().split("\n")


def _cross_half_overlaps(lines):
    """(kernel, instruction) of every packed fp32 op whose destination pair is a VGPR source pair of which the LOW lane reads the
    high dword (op_sel[i] = 1), or overlaps a source pair by one register."""
    bad = []
    kernel = None
    for l in lines:
        k = re.match(r"^(_Z\S+):", l)
        if k:
            kernel = k.group(1)
        m = re.match(r"\s*v_pk_(?:fma|mul|add)_f32 v\[(\d+):(\d+)\], (.*)$", l.rstrip())
        if not m:
            continue
        d0 = int(m.group(1))
        rest = m.group(3)
        mods = rest[rest.index("op_sel"):] if "op_sel" in rest else ""
        srcs = [t.strip() for t in (rest[:rest.index("op_sel")] if mods else rest).split(",") if t.strip()]
        sel, sel_hi = [0, 0, 0], [1, 1, 1]
        for name, dst in (("op_sel", sel), ("op_sel_hi", sel_hi)):
            mm = re.search(name + r":\[([0-9,]+)\]", mods)
            if mm:
                v = [int(x) for x in mm.group(1).split(",")]
                dst[:len(v)] = v
        for i, t in enumerate(srcs[:3]):
            ms = re.match(r"v\[(\d+):(\d+)\]", t)
            if not ms:
                continue
            s0 = int(ms.group(1))
            if (s0 == d0 and sel[i] == 1) or abs(s0 - d0) == 1:
                bad.append((kernel, l.strip()))
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_attention_kernels_have_no_packed_op_overlapping_a_cross_read_source(tmp_path):
    lines = _asm("paged_attn.hip", str(tmp_path))
    assert sum("v_pk_fma_f32" in l for l in lines) > 500, "the packed QK / PV products are gone: the rule has nothing to check"
    bad = _cross_half_overlaps(lines)
    assert not bad, f"{len(bad)} packed fp32 ops whose destination overlaps a source read across halves, e.g. {bad[:4]}"
