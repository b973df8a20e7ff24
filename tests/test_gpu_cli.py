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


def test_bench_line_contract(gpu):
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's command shape) prints ONE JSON line with the contract's
    fields, the `roofline` and `cpu_baseline` objects, and a same-run `parity` object that holds every compared pixel of the
    timed batch's first frame identical to the oracle's."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--min-seconds", "0.3"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "single_frame_ms", "gather_ms_avg", "gather_exposed_ms_avg"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["value"] > 1000 and abs(d["ms_per_step"] * d["value"] * 1e3 / d["config"]["rays_per_frame"] - 1.0) < 0.02
    assert "workload" in d["config"] and d["config"]["frames_per_launch"] == 8 and d["vs_baseline"] is None and d["dtype"] == "f32"
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "survey_8d_frac", "kernel_ms_avg", "profile_matches_source"):
        assert k in rf, k
    assert rf["profile_matches_source"] is True and 0.3 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 1 and cb["cores"] >= 1 and cb["unit"] == "Mrays/s" and cb["sample"]
    pa = d["parity"]
    assert pa["pixels"] == 1920 * 1080 and pa["pixels_outside_tolerance"] == 0 and pa["bit_identical_pixels"] >= 0.9999 * pa["pixels"]
