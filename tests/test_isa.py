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
    # node fetches (current / next / skip target) and primitive records are scalar.  Since round 4 the record arrays are
    # read through constant-address-space pointers (render_kernel.hip: ld4), so a wave-uniform address IS a scalar load
    # whatever precedes it -- stores, atomics, barriers, the inline assembly of the octant run
    assert x8 >= 5 and x4 >= 4, (x8, x4)
    # ... and what is left on the vector path is per-lane data: normals of candidate hits and shading (one copy per
    # inlined copy of the leaf code: six loops since round 4), the per-lane triangle records of the fat-leaf culling
    # (three 16-byte words in each of those loops), the window records of the reference tiling
    vmem = len(re.findall(r"global_load_dword", body))
    assert vmem <= 100, vmem
    lane = kernel_body(asm, LANE_FAST)
    assert len(re.findall(r"global_load_dword", lane)) >= 10      # the lane walker gathers per lane, by design
    for k in (PACKET_STATS, PACKET_REGROUP, ADAPTIVE_REGROUP, PACKET_DEEP):
        body = kernel_body(asm, k)
        assert len(re.findall(r"s_load_dwordx8", body)) >= 5, k
    assert "s_barrier" in kernel_body(asm, PACKET_REGROUP)


def test_octant_run_is_17_valu_per_node_step(asm):
    """oct_run (render_kernel.hip): the hand-written walk from node to node for packets that share a direction octant.
    Per step: the request for node i + 1 goes out BEFORE the slab arithmetic; 17 VALU (six subtractions, six
    multiplications, max with 0, min with tMax, v_max3, v_min3, the compare into vcc) against the 29 of the generic
    loop: no per-axis v_min / v_max pairs and -- the tree's boxes are nested -- no per-lane memory of the subtrees a lane
    has left (no v_cndmask, no integer VALU); eight octants x two loops (closest hit, first hit), each body twice (record
    buffers swapped)."""
    body = kernel_body(asm, PACKET_FAST)
    runs = re.findall(r"^\.Loct(\d+)_A:\n(.*?)^\.Loct\1_end:", body, flags=re.S | re.M)
    assert len(runs) == 16, len(runs)
    for _, text in runs:
        lines = [l.strip() for l in text.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        first_branch = next(i for i, l in enumerate(lines) if l.startswith("s_cbranch"))
        head = lines[:first_branch]
        assert head[0].startswith("s_waitcnt lgkmcnt(0)") and any(l.startswith("s_load_dwordx8") for l in head[:5]), head[:5]
        assert sum(l.startswith("v_") for l in head) == 17, head
        assert lines[first_branch + 1].startswith("s_cbranch_vccz"), lines[first_branch:first_branch + 2]
        assert not re.search(r"v_cndmask|v_max_i32|v_cmp_ge_i32", text)
        assert len(re.findall(r"v_max3_f32", text)) == 2 and len(re.findall(r"v_min3_f32", text)) == 2       # bodies A and B
        assert len(re.findall(r"v_(min|max)_f32_e32", text)) == 4                                                # max(., 0), min(., tMax)
    # the counting build keeps the generic loop (its counters are per step); the camera-ray launch of the wavefront form
    # has the run, and so have its level launches for their shadow rays (first hit only: eight copies)
    assert ".Loct" not in kernel_body(asm, PACKET_STATS)
    assert ".Loct" in whole_body(asm, STREAM_PRIMARY)
    assert len(re.findall(r"^\.Loct\d+_A:", whole_body(asm, STREAM_LEVEL), flags=re.M)) == 8


def test_no_fma_contraction(asm):
    for k in (PACKET_FAST, LANE_FAST):
        body = kernel_body(asm, k)
        assert "v_pk_fma_f32" not in body and "v_mad_f32" not in body and "v_mac_f32" not in body
        fma = len(re.findall(r"v_fma_f32|v_fmac_f32", body))
        div = len(re.findall(r"v_div_scale_f32", body)) // 2 + len(re.findall(r"v_sqrt_f32", body))
        # every v_fma belongs to an IEEE divide / sqrt / powf expansion: ~6 per divide, more in the libm code
        assert div > 0 and fma <= 12 * div + 80, (fma, div)


def test_register_budget_and_no_scratch(asm):
    # packet walker (issue bound): 7 waves per SIMD = 72 VGPRs.  The state of the radiance recursion waits in LDS between
    # its uses (render_kernel.hip: radiance), so NOTHING is spilled inside the step loop -- neither in the traversal loops
    # nor around them, and the kernel of the benchmark touches no scratch memory at all (70 VGPRs since round 4).
    for k in (PACKET_FAST, PACKET_REGROUP):
        m = meta(asm, k)
        assert m["vgpr_count"] <= 72, m
        assert m["vgpr_spill_count"] <= 32, m
        if k != PACKET_FAST:                                     # the regrouping variant is a diagnostics build
            continue
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, m
        assert "scratch_" not in kernel_body(asm, k)
        lines = kernel_body(asm, k).split("\n")
        # the step loop (depth 2: one traversal per trip) with the traversal loops (depth 3, 4) inside it: from its header
        # to the depth-2 inner loop that follows it (the fold of the radiance levels)
        d2 = [i for i, l in enumerate(lines) if "Loop Header: Depth=2" in l and "Inner" not in l]
        d3 = [i for i, l in enumerate(lines) if "Loop Header: Depth=3" in l]
        assert d2 and d3 and d2[0] < d3[0]
        after = [i for i, l in enumerate(lines) if "Inner Loop Header: Depth=2" in l and i > d3[-1]]
        assert after
        scratch = [i for i, l in enumerate(lines) if "scratch_" in l and d2[0] <= i <= after[0]]
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
    assert m["vgpr_count"] <= 72 and m["vgpr_spill_count"] <= 12, m       # (per sample, outside the traversal loops)
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
    on the v_min / v_max itself (6 instructions per reduction, 12 reductions, in each copy of the leaf code of the NaN-free
    loops: closest hit and first hit, generic loop -- whose body exists twice -- and octant loop);
    the packet kernel still fits its 72 registers without scratch (test_register_budget_and_no_scratch); the level
    launches of the wavefront form are built without it and touch no scratch memory inside their loops"""
    body = kernel_body(asm, PACKET_FAST)
    assert len(re.findall(r"v_min_f32_dpp", body)) == 6 * 36 and len(re.findall(r"v_max_f32_dpp", body)) == 6 * 36
    # (at 64 registers the level kernels park ten of them in scratch memory around the shading of a hit -- the binary64
    # evaluation of `**` (device_math.hpp: glibc_powf) is what tipped it: +0.5 % on the 1 M-triangle scene, DESIGN §5 -- at
    # the kernel's entry, after its last loop and in the loop over the lights; the traversal loops hold one reload)
    m = meta(asm, STREAM_LEVEL)
    assert m["vgpr_spill_count"] <= 16 and m["private_segment_fixed_size"] <= 48, m
    label, note, deep = "entry", "", 0
    for line in whole_body(asm, STREAM_LEVEL).split("\n"):
        lm = re.match(r"^(\.L\w+):\s*(;.*)?$", line)
        if lm:
            label, note = lm.group(1), lm.group(2) or ""
        elif "scratch_" in line and re.search(r"Depth=([2-9])", note):
            assert "scratch_load" in line, (label, note, line)
            deep += 1
    assert deep <= 2, deep
    assert "v_min_f32_dpp" not in whole_body(asm, STREAM_LEVEL)
