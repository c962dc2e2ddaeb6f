"""Build rules the GPU found the hard way, checked on the CPU by cross-compiling every kernel source to gfx950 assembly.

Rule 1 (round 4): no packed fp32 operation (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) reads the HIGH dword of a VGPR src1 pair in
its LOW lane (op_sel[1] = 1).  Measured on MI355X (scripts/probes/pkfma_src1.hip, profiles/r04_pkfma_probe.txt): every broadcast
form of these instructions -- src0 / src1 / src2, low or high dword -- agrees with the scalar instruction on 1e12 executions when
the process is alone on the GPU; while ANOTHER WAVE ON THE SAME SIMD ISSUES MFMAs (another process's GEMM: the Code2Wav loop of
tests/test_gpu_colocation.py, or a bare MFMA loop: scripts/probes/aggressor.hip) exactly the two forms with op_sel[1] = 1 return wrong
results -- 5 % of the executions beside the bare loop, ~4 per million beside the vocoder -- and all the others still none; loads, LDS
traffic, LDS-DMA and VALU-only neighbours do nothing.  The decode attention's
first packed QK product (two heads per FMA, K broadcast from the high dword of src1) hit it: parity-green and bit-stable alone,
garbage in one head of a pair in ~10 % of its launches beside the vocoder (scripts/probes/attn_coloc_probe.py).  hipcc chooses the
operand order and the selector: this test keeps the form out of every kernel of the library."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ht_vllm_omni_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(name, tmp_path, defines=()):
    out = os.path.join(tmp_path, name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *defines, "-I" + os.path.join(ROOT, "include"), "-S",
                    "--cuda-device-only", os.path.join(CSRC, name), "-o", out], check=True, capture_output=True, timeout=900)
    return open(out).read().split("\n")


def src1_high_into_low_lane(lines):
    """(kernel, instruction) of every packed fp32 op with a VGPR src1 and op_sel[1] = 1."""
    bad, kernel, packed = [], None, 0
    for l in lines:
        k = re.match(r"^(_Z\S+):", l)
        if k:
            kernel = k.group(1)
        m = re.match(r"\s*v_pk_(?:fma|mul|add)_f32 \S+, \S+, (\S+?)(?:,.*?)?(?: (op_sel.*))?$", l.rstrip())
        if not m:
            continue
        packed += 1
        mm = re.search(r"op_sel:\[([0-9,]+)\]", m.group(2) or "")
        if mm and m.group(1).startswith("v["):
            sel = [int(x) for x in mm.group(1).split(",")]
            if len(sel) > 1 and sel[1] == 1:
                bad.append((kernel, l.strip()))
    return bad, packed


def test_the_scanner_sees_the_form():
    bad, n = src1_high_into_low_lane(["_Zk:", "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]",
                                      "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]",
                                      "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]",
                                      "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[1,0,1]",
                                      "\tv_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel:[0,1]"])
    assert n == 5 and [b[1].split()[0] for b in bad] == ["v_pk_fma_f32", "v_pk_mul_f32"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_the_scanner_still_reads_this_compilers_assembly(tmp_path):
    """A canary for the scan itself (a compiler update that prints the modifier differently would let the rule pass
    silently): the probe that PROVED the hazard (scripts/probes/pkfma_src1.hip) contains, by inline assembly, exactly two instructions of
    the forbidden form -- v_pk_fma_f32 and v_pk_mul_f32 with op_sel[1] = 1 on a VGPR src1 -- next to every harmless selector form.  Cross-
    compiled with THIS hipcc, the scanner must find those two and only those."""
    out = os.path.join(str(tmp_path), "pk.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "scripts", "probes", "pkfma_src1.hip"), "-o", out], check=True, capture_output=True, timeout=900)
    bad, n = src1_high_into_low_lane(open(out).read().split("\n"))
    assert n >= 10, f"the probe's packed fp32 instructions are not recognised any more ({n} seen)"
    assert sorted(b[1].split()[0] for b in bad) == ["v_pk_fma_f32", "v_pk_mul_f32"], bad


# both libraries: libomni_talker.so and the diagnostics build (-DOMNI_DEBUG_HOOKS turns the policy knobs into run-time variables:
# other code, other register allocation)
@pytest.mark.parametrize("defines", [(), ("-DOMNI_DEBUG_HOOKS",)], ids=["product", "debug-hooks"])
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_kernel_reads_the_high_dword_of_src1_in_the_low_lane_of_a_packed_fp32_op(tmp_path, defines):
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        asms = list(pool.map(lambda n: _asm(n, str(tmp_path), defines), names))
    total, bad = 0, []
    for name, lines in zip(names, asms):
        b, n = src1_high_into_low_lane(lines)
        total += n
        bad += [(name,) + x for x in b]
    assert total > 10000, "the packed fp32 products are gone from the library: the rule has nothing to check"
    assert not bad, f"{len(bad)} packed fp32 ops with op_sel[1] = 1 on a VGPR src1, e.g. {bad[:4]}"
