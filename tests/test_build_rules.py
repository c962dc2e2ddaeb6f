"""Build rules the GPU found the hard way, checked on the CPU by cross-compiling to gfx950 assembly.

Rule 1 (round 4): no `v_pk_fma_f32` of the attention kernels takes a BROADCAST VGPR pair in src1.  The decode attention's QK
product packs two heads per FMA with the K element broadcast by op_sel; with the splat in src1 (op_sel:[0,1,0] /
op_sel_hi:[1,0,1] on a pair fresh out of v_cvt_pk_f32_fp8) the kernel was bit-stable alone on the GPU and produced different low
bits beside another process's kernels (tests/test_gpu_colocation.py; scripts/coloc_probe.py), with the same arithmetic and the
splat in src0 it is stable.  The operand order in pa_body.cuh decides which form hipcc emits: this test keeps it decided."""
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


def _src1_broadcasts(lines):
    bad = []
    kernel = None
    for l in lines:
        k = re.match(r"^(_Z\S+):", l)
        if k:
            kernel = k.group(1)
        m = re.match(r"\s*v_pk_fma_f32 \S+, \S+, (\S+?), \S+?(?: (op_sel.*))?$", l.rstrip())
        if not m or not m.group(1).startswith("v["):
            continue
        mods = m.group(2) or ""
        sel, sel_hi = [0, 0, 0], [1, 1, 1]
        for name, dst in (("op_sel", sel), ("op_sel_hi", sel_hi)):
            mm = re.search(name + r":\[([0-9,]+)\]", mods)
            if mm:
                v = [int(x) for x in mm.group(1).split(",")]
                dst[:len(v)] = v
        if sel[1] == sel_hi[1]:                      # both halves read the same dword of src1
            bad.append((kernel, l.strip()))
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_attention_kernels_keep_packed_fma_broadcasts_out_of_src1(tmp_path):
    lines = _asm("paged_attn.hip", str(tmp_path))
    assert sum("v_pk_fma_f32" in l for l in lines) > 500, "the packed QK / PV products are gone: the rule has nothing to check"
    bad = _src1_broadcasts(lines)
    assert not bad, f"{len(bad)} v_pk_fma_f32 with a broadcast VGPR pair in src1, e.g. {bad[:3]}"
