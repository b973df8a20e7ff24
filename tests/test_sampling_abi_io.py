"""Sampling.hs tile windows, the C-ABI surface, error behaviour without a GPU, image writers.  CPU only."""
import ctypes as C
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_py
import yahr_amd
from yahr_amd import _lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- Sampling.hs -----------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,n", [(40, 40, 5), (40, 40, 16), (1, 1, 1), (20, 1, 3), (1, 20, 3),      # Spec.hs:332
                                   (1920, 1080, 8192), (256, 256, 256), (37, 91, 12), (7, 5, 64)])
def test_square_batches_cover_without_overlap_and_match_oracle(w, h, n):
    L = oracle_py.lib()
    cnt = np.zeros((h, w), dtype=np.int32)
    for b in range(n):
        win = yahr_amd.batch_window(w, h, b, n)
        ref = (C.c_int32 * 4)()
        L.oracle_batch_window(w, h, b, n, ref)
        assert win == tuple(ref)
        cnt[win[1]:win[3], win[0]:win[2]] += 1
    assert (cnt == 1).all()


def test_square_batches_enumeration_is_u_major():
    b = yahr_amd.square_batches(4, 4, 1)
    assert b[0][:5] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0)]          # [(u,v) | u <- .., v <- ..]  Sampling.hs:6


def test_n_batches_matches_oracle():
    L = oracle_py.lib()
    for t, w, h in [(1, 1920, 1080), (8, 256, 256), (1, 3840, 2160), (64, 1920, 1080), (2, 10, 10), (1, 1024, 768)]:
        assert yahr_amd.n_batches(t, w, h) == L.oracle_n_batches(t, w, h)
    assert yahr_amd.n_batches(1, 1920, 1080) == 8192


def test_shard_rows_partition_the_image():
    for height, world, strip in [(1080, 8, 8), (1080, 3, 16), (10, 4, 8), (7, 2, 1), (2160, 8, 8), (5, 8, 8)]:
        seen = np.zeros(height, dtype=np.int32)
        for r in range(world):
            rows = yahr_amd.shard_global_rows(height, world, r, strip)
            assert len(rows) == yahr_amd.shard_rows(height, world, r, strip)
            assert (np.diff(rows) > 0).all()
            seen[rows] += 1
        assert (seen == 1).all()


# ---- C ABI surface -----------------------------------------------------------------------------------
def header_functions():
    text = open(os.path.join(ROOT, "include", "yahr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yahr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == names                      # the Python binding covers the whole header
    # no C++ or torch types leak through the boundary
    assert not [s for s in exported if s.startswith("yahr_") and s not in names]
    assert b"gfx950" in _lib.lib.yahr_version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(_lib.Node) == 32 and C.sizeof(_lib.Prim) == 48 and C.sizeof(_lib.PrimNormals) == 48
    assert C.sizeof(_lib.Material) == 32 and C.sizeof(_lib.Light) == 32
    assert C.sizeof(_lib.RenderOpts) == 64
    o = yahr_amd.make_opts()
    assert (o.spp, o.n_shards, o.strip_rows, o.traverse, o.tiles) == (1, 1, 8, 0, 0)


@pytest.mark.skipif(yahr_amd.device_count() > 0, reason="checks the no-GPU error path")
def test_device_entry_points_fail_loudly_without_a_gpu():
    s = scenes.scene_fixed(8, 8).to_scene().build()
    with pytest.raises(yahr_amd.YahrError) as e:
        yahr_amd.Context(s, 0)
    assert e.value.code == -6 and "no CPU fallback" in str(e.value)
    with pytest.raises(yahr_amd.YahrError) as e:
        yahr_amd.render_image(s)
    assert e.value.code == -6
    with pytest.raises(yahr_amd.YahrError):
        yahr_amd.render_image(s, n_devices=2)
    with pytest.raises(yahr_amd.YahrError) as e:
        yahr_amd.MultiContext(s, 2)
    assert e.value.code == -6


def test_malformed_flat_scene_is_rejected_before_any_launch():
    s = scenes.scene_fixed(8, 8).to_scene().build()
    f = s.flat.contents
    bad = _lib.FlatScene()
    C.memmove(C.byref(bad), C.byref(f), C.sizeof(_lib.FlatScene))
    nodes = (_lib.Node * f.n_nodes)()
    C.memmove(nodes, f.nodes, C.sizeof(nodes))
    nodes[0].a = 0                                   # a skip link that does not move forward
    bad.nodes = C.cast(nodes, C.POINTER(_lib.Node))
    h = C.c_void_p()
    rc = _lib.lib.yahr_ctx_create(C.byref(bad), 0, C.byref(h))
    assert rc == -1 and b"skip link" in _lib.lib.yahr_last_error()
    bad.nodes = f.nodes
    bad.abi_version = 99
    assert _lib.lib.yahr_ctx_create(C.byref(bad), 0, C.byref(h)) == -1


def test_programmatic_api_argument_errors():
    s = yahr_amd.Scene()
    with pytest.raises(yahr_amd.YahrError):
        s.set_culling(4, 7)
    with pytest.raises(yahr_amd.YahrError):
        s.end_subsampled()
    s.begin_subsampled(0.5)
    with pytest.raises(yahr_amd.YahrError):
        s.build()                                    # unterminated Subsampled


# ---- image writers ---------------------------------------------------------------------------------
def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body) & 0xFFFFFFFF
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[0], hdr[1], hdr[2], hdr[3]
    assert depth == 8 and ctype == 2
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_ppm_pfm(tmp_path):
    rgb = np.array([[[0.0, 0.5, 1.0], [1.5, -0.25, 0.999]], [[np.nan, 0.00391, 0.2], [1e-9, np.inf, 0.75]]], dtype=np.float32)
    # truncate . (255 *) . min 1 . max 0   (JuicyPixels RGBF -> RGB8; parity unpinned)
    want = np.array([[[0, 127, 255], [255, 0, 254]], [[0, 0, 51], [0, 255, 191]]], dtype=np.uint8)
    p = str(tmp_path / "o.png")
    yahr_amd.write_png(p, rgb)
    got = decode_png(p)
    assert np.array_equal(got, want)
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), want)
    except ImportError:
        pass
    p = str(tmp_path / "o.ppm")
    yahr_amd.write_ppm(p, rgb)
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n2 2\n255\n") and np.array_equal(np.frombuffer(data[11:], dtype=np.uint8).reshape(2, 2, 3), want)
    p = str(tmp_path / "o.pfm")
    yahr_amd.write_pfm(p, rgb)
    data = open(p, "rb").read()
    assert data.startswith(b"PF\n2 2\n-1.0\n")
    back = np.frombuffer(data[len(b"PF\n2 2\n-1.0\n"):], dtype=np.float32).reshape(2, 2, 3)[::-1]
    assert np.array_equal(back.view(np.uint32), rgb.view(np.uint32))
    with pytest.raises(yahr_amd.YahrError):
        yahr_amd.write_png("/nonexistent_dir/x.png", rgb)


def test_png_deflated_in_bands_decodes_to_the_same_pixels(tmp_path, monkeypatch):
    """images of 128 rows and more are deflated in bands by several threads into ONE zlib stream (sync flush between
    the bands, combined Adler-32): whatever the number of bands, the file must decode to the same pixels -- with this
    file's decoder (zlib.decompress checks the Adler-32 of the whole) and with PIL when it is there"""
    rng = np.random.default_rng(5)
    h, w = 333, 257
    rgb = (rng.random((h, w, 3)) ** 3).astype(np.float32)
    rgb[100:200] = np.linspace(0, 1, w, dtype=np.float32)[None, :, None]            # a compressible part too
    want = np.floor(np.float32(255.0) * np.clip(rgb, np.float32(0), np.float32(1))).astype(np.uint8)   # binary32 product, as to_ldr
    sizes = {}
    for threads in (1, 2, 5, 16):
        monkeypatch.setenv("YAHR_PNG_THREADS", str(threads))
        p = str(tmp_path / ("o%d.png" % threads))
        yahr_amd.write_png(p, rgb)
        assert np.array_equal(decode_png(p), want), threads
        sizes[threads] = os.path.getsize(p)
        try:
            from PIL import Image
            assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), want)
        except ImportError:
            pass
    assert sizes[5] < 1.02 * sizes[1]                                               # the bands cost next to nothing in size


# ---- command line (argument handling; the render itself needs a GPU: tests/test_gpu_cli.py) ----------
YAHR = os.path.join(ROOT, "bin", "yahr")


def run_cli(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([YAHR, *args], capture_output=True, text=True, env=e)


def test_cli_usage_and_errors(tmp_path):
    r = run_cli()
    assert r.returncode == 1 and "two positional" in r.stderr
    r = run_cli("/nonexistent.yahrr", str(tmp_path / "o.png"))
    assert r.returncode == 1 and "does not exist" in r.stderr
    bad = tmp_path / "bad.yahrr"
    bad.write_text("Scene { nope }")
    r = run_cli(str(bad), str(tmp_path / "o.png"))
    assert r.returncode == 1 and "no parse" in r.stderr
    good = tmp_path / "good.yahrr"
    good.write_text(scenes.scene_fixed(8, 8).to_text())
    r = run_cli(str(good), str(tmp_path / "o.png"), "-p", "bogus")
    assert r.returncode == 1 and "Non-exhaustive" in r.stderr              # main.hs:134-137
    r = run_cli(str(good), str(tmp_path / "o.png"), "--frobnicate")
    assert r.returncode == 1
    assert run_cli("--help").returncode == 0


@pytest.mark.skipif(yahr_amd.device_count() > 0, reason="checks the no-GPU error path")
def test_cli_without_gpu_fails_loudly(tmp_path):
    good = tmp_path / "good.yahrr"
    good.write_text(scenes.scene_fixed(8, 8).to_text())
    # the Blender add-on's argv shape (render_engine.py:44-47)
    r = run_cli(str(good), str(tmp_path / "o.png"), "-p", "eval", "+RTS", "-N4")
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "o.png").exists()


def test_committed_profile_belongs_to_the_committed_kernels():
    """bench.py quotes `roofline.frac` from instruction counts of a rocprofv3 PMC profile kept under profiles/ -- only while
    the fingerprint stored with it (kernel sources + compiler flag lines, yahr_amd/srchash.py) is the tree's.  This test is
    the reminder to re-profile (tools/final_profiles.sh on the GPU box) after the last kernel change of a round."""
    import json
    from yahr_amd.srchash import kernel_source_hash
    prof = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert prof["workload"] == "bunny" and prof["valu_insts_per_frame"] > 1e8
    assert prof["kernel_source_sha256"] == kernel_source_hash(), "kernel sources changed after the profile was taken: re-profile"
