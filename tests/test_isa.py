"""Static checks on the gfx950 code hipcc generates for the render kernel (cross-compiled, no GPU needed).

The packet walker's speed rests on properties the compiler only grants under conditions that are easy to break
by accident (a store ahead of a scene load, an `asm volatile`, a shuffle result used as a loop exit ...): then
the kernel stays CORRECT but silently falls back to per-lane gathers and runs 2x slower.  These tests pin:
  * node / primitive records of the packet walker arrive through scalar loads (s_load_dwordx8 / x4),
  * no FMA contraction anywhere outside the IEEE divide / sqrt expansions (parity with the non-fused reference),
  * no scratch (register spilling to memory), register budget fits at least 4 waves per SIMD.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yahr_amd", "csrc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "render_kernel.s"
    # same code-generation flags as yahr_amd/csrc/Makefile (HIPFLAGS)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-ffp-contract=off" in mk and "-fno-slp-vectorize" in mk
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                    "-fno-slp-vectorize", "-S", "--cuda-device-only", os.path.join(CSRC, "render_kernel.hip"),
                    "-o", str(out)], check=True, capture_output=True)
    return open(out).read()


def kernel_body(asm, mangled):
    m = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(mangled), asm, flags=re.S | re.M)
    assert m, mangled
    return m.group(1)


def meta(asm, mangled):
    m = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(mangled), asm, flags=re.S)
    assert m, mangled
    return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(1))}


PACKET_FAST = "_Z8k_renderILi2ELb0EEv6DScene7RParamsPfPy"
LANE_FAST = "_Z8k_renderILi1ELb0EEv6DScene7RParamsPfPy"


def test_packet_walker_uses_scalar_loads(asm):
    body = kernel_body(asm, PACKET_FAST)
    x8 = len(re.findall(r"s_load_dwordx8", body))
    x4 = len(re.findall(r"s_load_dwordx4", body))
    vmem = len(re.findall(r"global_load_dword", body))
    # node fetches (current / next / skip target) and primitive records are scalar ...
    assert x8 >= 5 and x4 >= 4, (x8, x4)
    # ... and what is left on the vector path is the per-lane data only (normals of candidate hits, shading)
    assert vmem <= 16, vmem
    lane = kernel_body(asm, LANE_FAST)
    assert len(re.findall(r"global_load_dword", lane)) >= 10      # the lane walker gathers per lane, by design


def test_no_fma_contraction(asm):
    for k in (PACKET_FAST, LANE_FAST):
        body = kernel_body(asm, k)
        assert "v_pk_fma_f32" not in body and "v_mad_f32" not in body and "v_mac_f32" not in body
        fma = len(re.findall(r"v_fma_f32|v_fmac_f32", body))
        div = len(re.findall(r"v_div_scale_f32", body)) // 2 + len(re.findall(r"v_sqrt_f32", body))
        # every v_fma belongs to an IEEE divide / sqrt / powf expansion: ~6 per divide, more in the libm code
        assert div > 0 and fma <= 12 * div + 80, (fma, div)


def test_register_budget_and_no_scratch(asm):
    for k in (PACKET_FAST, LANE_FAST):
        m = meta(asm, k)
        assert m["private_segment_fixed_size"] == 0, m          # no spilling to scratch memory
        assert m["vgpr_spill_count"] == 0
        assert m["vgpr_count"] <= 128, m                         # >= 4 waves per SIMD
