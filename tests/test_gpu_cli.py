"""End-to-end command line on the GPU: reproduces the reference's compat/test_yahr.py (write a scene, run
`yahr scene out`, exit status 0 and a non-empty image) and the Blender add-on's argv (render_engine.py:44-47)."""
import os
import subprocess

import numpy as np
import pytest

import yahr_amd
from oracle_py import OracleScene
from yahr_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAHR = os.path.join(ROOT, "bin", "yahr")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_reference_e2e(gpu, tmp_path):
    out = tmp_path / "testout.png"
    r = subprocess.run([YAHR, os.path.join(GOLDEN, "test_yahr_scene.yahr"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.stat(out).st_size > 0                                                    # test_yahr.py:65-66
    assert r.stdout.splitlines()[0] == "1 threads, 1 batches, parallel sequential"    # main.hs:140-141 (10x10 -> 1 batch)


@pytest.mark.parametrize("mode", ["sequential", "eval", "par", "hip"])
def test_blender_argv_and_pixels(gpu, tmp_path, mode):
    d = scenes.scene_fixed(64, 48)
    src = tmp_path / "blender_scene.yahr"
    src.write_text(d.to_text())
    out = tmp_path / "blender_scene.pfm"
    r = subprocess.run([YAHR, str(src), str(out), "-p", mode, "+RTS", "-N4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[0] == f"4 threads, {yahr_amd.n_batches(4, 64, 48)} batches, parallel {mode}"
    data = open(out, "rb").read()
    hdr = b"PF\n64 48\n-1.0\n"
    img = np.frombuffer(data[len(hdr):], dtype=np.float32).reshape(48, 64, 3)[::-1]
    ref, _ = OracleScene.from_desc(d).build().render(threads=4)
    err = np.abs(img - ref)
    assert (err <= 1e-4 * np.maximum(np.abs(img), np.abs(ref)) + 1e-6).all()
    png = tmp_path / "o.png"
    assert subprocess.run([YAHR, str(src), str(png)], capture_output=True).returncode == 0
    from test_sampling_abi_io import decode_png
    got = decode_png(str(png))
    want = (255.0 * np.clip(img, 0, 1)).astype(np.uint8)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.999


def test_stale_scene_needs_lenient_flag(gpu, tmp_path):
    from test_loader import stale_text
    src = tmp_path / "stale.yahrr"
    src.write_text(stale_text())
    out = tmp_path / "o.ppm"
    assert subprocess.run([YAHR, str(src), str(out)], capture_output=True).returncode == 1
    env = dict(os.environ, YAHR_LENIENT="1")
    assert subprocess.run([YAHR, str(src), str(out)], capture_output=True, env=env).returncode == 0
    assert os.stat(out).st_size == len(b"P6\n8 6\n255\n") + 8 * 6 * 3
