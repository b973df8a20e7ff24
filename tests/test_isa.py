"""Static checks on the gfx950 code hipcc generates for the render kernel (cross-compiled, no GPU needed).

The packet walker's speed rests on properties the compiler only grants under conditions that are easy to break
by accident (a store ahead of a scene load, an `asm volatile`, a shuffle result used as a loop exit ...): then
the kernel stays CORRECT but silently falls back to per-lane gathers and runs 2x slower.  These tests pin:
  * node / primitive records of the packet walker arrive through scalar loads (s_load_dwordx8 / x4),
  * no FMA contraction anywhere outside the IEEE divide / sqrt expansions (parity with the non-fused reference),
  * packet walker: 7 waves per SIMD with no spill inside the loop nest; lane walker: 8 waves.
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


def kernel_body(asm, prefix):
    """body of the kernel whose mangled name starts with `prefix` (template arguments; the parameter list follows)"""
    m = re.search(r"^%s\w*:[^\n]*\n(.*?)s_endpgm" % re.escape(prefix), asm, flags=re.S | re.M)
    assert m, prefix
    return m.group(1)


def meta(asm, prefix):
    m = re.search(r"\.name:\s+%s\w*\n(.*?)\.wavefront_size" % re.escape(prefix), asm, flags=re.S)
    assert m, prefix
    return {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(1))}


# k_render<MODE, STATS, WPB, DEEP>: MODE 1 lane / 2 packet / 3 adaptive; WPB waves per workgroup (4 = ray regrouping);
# DEEP = recursion depth > 4, upper levels of the radiance stack in global memory
PACKET_FAST = "_Z8k_renderILi2ELb0ELi1ELb0EE"
LANE_FAST = "_Z8k_renderILi1ELb0ELi1ELb0EE"
ADAPTIVE_FAST = "_Z8k_renderILi3ELb0ELi1ELb0EE"
PACKET_STATS = "_Z8k_renderILi2ELb1ELi1ELb0EE"
PACKET_REGROUP = "_Z8k_renderILi2ELb0ELi4ELb0EE"
ADAPTIVE_REGROUP = "_Z8k_renderILi3ELb0ELi4ELb0EE"
PACKET_DEEP = "_Z8k_renderILi2ELb0ELi1ELb1EE"


def test_packet_walker_uses_scalar_loads(asm):
    body = kernel_body(asm, PACKET_FAST)
    x8 = len(re.findall(r"s_load_dwordx8", body))
    x4 = len(re.findall(r"s_load_dwordx4", body))
    vmem = len(re.findall(r"global_load_dword", body))
    # node fetches (current / next / skip target) and primitive records are scalar ...
    assert x8 >= 5 and x4 >= 4, (x8, x4)
    # ... and what is left on the vector path is the per-lane data only (normals of candidate hits, shading)
    # (the window records of the reference tiling are loaded where the pixel is worked out, i.e. twice since round 3)
    # (+ 6 since round 4: the fat-leaf culling loads one triangle record per LANE, three 16-byte words, in each of the two
    # NaN-free copies of the loop)
    assert vmem <= 30, vmem
    lane = kernel_body(asm, LANE_FAST)
    assert len(re.findall(r"global_load_dword", lane)) >= 10      # the lane walker gathers per lane, by design
    # the record arrays are `const __restrict__` kernel arguments, so counters (atomics, clock reads) and the
    # workgroup barriers of the regrouping variants do not cost the packet walker its scalar loads either
    # ... nor do the stores of the deep-scene variant (levels of the radiance stack spilled to global memory mid-loop)
    for k in (PACKET_STATS, PACKET_REGROUP, ADAPTIVE_REGROUP, PACKET_DEEP):
        body = kernel_body(asm, k)
        assert len(re.findall(r"s_load_dwordx8", body)) >= 5, k
    assert "s_barrier" in kernel_body(asm, PACKET_REGROUP)


def test_no_fma_contraction(asm):
    for k in (PACKET_FAST, LANE_FAST):
        body = kernel_body(asm, k)
        assert "v_pk_fma_f32" not in body and "v_mad_f32" not in body and "v_mac_f32" not in body
        fma = len(re.findall(r"v_fma_f32|v_fmac_f32", body))
        div = len(re.findall(r"v_div_scale_f32", body)) // 2 + len(re.findall(r"v_sqrt_f32", body))
        # every v_fma belongs to an IEEE divide / sqrt / powf expansion: ~6 per divide, more in the libm code
        assert div > 0 and fma <= 12 * div + 80, (fma, div)


def test_register_budget_and_no_scratch(asm):
    # packet walker (issue bound): 7 waves per SIMD = 72 VGPRs and NO scratch at all in the kernel of the benchmark: the
    # pixel's identity is recomputed for the final store instead of being carried (and spilled) through the radiance
    # computation -- until round 3 six spilled dwords per lane made 290 MB of scratch writes per 8-frame launch
    for k in (PACKET_FAST, PACKET_REGROUP):
        m = meta(asm, k)
        assert m["vgpr_count"] <= 72, m
        assert m["vgpr_spill_count"] <= 32, m
        if k != PACKET_FAST:                                     # the regrouping variant is a diagnostics build
            continue
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, m
        assert "scratch_" not in kernel_body(asm, k)
        lines = kernel_body(asm, k).split("\n")
        # the traversal loops: from the first depth-3 loop header (node loop; the primitive loops are depth 4) to
        # the depth-2 inner loop that follows them (the fold of the radiance levels)
        d3 = [i for i, l in enumerate(lines) if "Loop Header: Depth=3" in l]
        assert d3
        after = [i for i, l in enumerate(lines) if "Inner Loop Header: Depth=2" in l and i > d3[-1]]
        assert after
        scratch = [i for i, l in enumerate(lines) if "scratch_" in l and d3[0] <= i <= after[0]]
        assert not scratch, (k, scratch[:4])
    # lane / adaptive walkers (gather-latency bound): built for all 8 wave slots of a SIMD, i.e. 64 VGPRs; the
    # values that do not fit (radiance state that is dead during a traversal) are spilled around the loops
    for k in (LANE_FAST, ADAPTIVE_FAST):
        m = meta(asm, k)
        assert m["vgpr_count"] <= 64, m
        assert m["vgpr_spill_count"] <= 40, m                      # (lane 12, adaptive 36 as of round 3)


# the wavefront form: k_stream_primary<STATS>, k_stream_level<MODE, STATS>, the counting sort and the resolve kernels
STREAM_PRIMARY = "_Z16k_stream_primaryILb0EE"
STREAM_LEVEL = "_Z14k_stream_levelILi3ELb0EE"


def whole_body(asm, prefix):
    """like kernel_body, but to the end of the function: these kernels leave early (s_endpgm) when their wavefront lies
    beyond the queue's length"""
    m = re.search(r"^%s\w*:[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(prefix), asm, flags=re.S | re.M)
    assert m, prefix
    return m.group(1)


def test_wavefront_form_kernels(asm):
    # camera rays: the packet walker with its scalar record loads, although the kernel ends in queue appends
    # (atomics and stores -- they all come after the last traversal, so the loads keep their scalar path)
    body = whole_body(asm, STREAM_PRIMARY)
    assert len(re.findall(r"s_load_dwordx8", body)) >= 5
    m = meta(asm, STREAM_PRIMARY)
    assert m["vgpr_count"] <= 72 and m["vgpr_spill_count"] <= 8, m
    # level launches: adaptive walker (packet start with scalar loads, then per-lane gathers), all 8 wave slots
    body = whole_body(asm, STREAM_LEVEL)
    assert len(re.findall(r"s_load_dwordx8", body)) >= 5 and len(re.findall(r"global_load_dword", body)) >= 20
    m = meta(asm, STREAM_LEVEL)
    assert m["vgpr_count"] <= 64 and m["vgpr_spill_count"] <= 24, m
    for k in (STREAM_PRIMARY, STREAM_LEVEL):
        b = whole_body(asm, k)
        assert "v_mad_f32" not in b and "v_mac_f32" not in b and "v_pk_fma_f32" not in b
    # the small kernels between them: nothing spilled
    for k in ("_Z11k_sort_scan", "_Z14k_sort_scatter", "_Z22k_stream_resolve_lanes", "_Z16k_stream_resolve7RParams"):
        m = meta(asm, k)
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (k, m)


def test_fat_leaf_culling_code(asm):
    """leaf_culled (render_kernel.hip): the ray-space box of the packet comes from DPP ladders whose cross-lane move rides
    on the v_min / v_max itself (6 instructions per reduction, 12 reductions, in each of the two NaN-free loops --
    closest hit, first hit -- whose bodies exist twice with the node buffers swapped);
    the packet kernel still fits its 72 registers without scratch (test_register_budget_and_no_scratch), and the level
    launches of the wavefront form, which are built without it, spill nothing any more"""
    body = kernel_body(asm, PACKET_FAST)
    assert len(re.findall(r"v_min_f32_dpp", body)) == 4 * 36 and len(re.findall(r"v_max_f32_dpp", body)) == 4 * 36
    m = meta(asm, STREAM_LEVEL)
    assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, m
    assert "v_min_f32_dpp" not in whole_body(asm, STREAM_LEVEL)
