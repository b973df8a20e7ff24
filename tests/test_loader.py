"""The .yahrr loader (yahr_amd/csrc/scene_parse.cpp) against the derived-Read grammar of the reference
(Scene.hs:42,50,58 + main.hs:117) and the `expand` rules (Scene.hs:61-86).  CPU only."""
import os

import numpy as np
import pytest

import yahr_amd
from oracle_py import OracleScene
from yahr_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

MIN_SCENE = ('Scene { integrator = WhittedIntegrator { recursionDepth = 2 }, cullingMode = BVH 16 Midpoint, '
             'camera = Camera { imW = 8, imH = 6, focalLength = 1, lookDir = Vec3 0 0 1, upDir = Vec3 0 1 0, '
             'position = Vec3 0 0 0 }, materials = [%s], lights = [%s], objects = [%s] }')
MAT = 'BlinnPhongMaterial { id = "m", ambient = Vec3 0 0 0, diffuse = Vec3 1 1 1, specular = Vec3 0 0 0, shininess = 1 }'


def parse(objects="", lights="", mats=MAT, lenient=False):
    return yahr_amd.Scene.parse(MIN_SCENE % (mats, lights, objects), lenient=lenient).build()


def test_reference_serialiser_output_parses():
    """tests/golden/test_yahr_scene.yahr was written by the reference's own compat/yahr.py (make_fixtures.py):
    bare negative literals, `(BVH 16 Midpoint)`, `%d` focal length, parenthesised constructor arguments."""
    s = yahr_amd.Scene.load(os.path.join(GOLDEN, "test_yahr_scene.yahr")).build()
    assert (s.width, s.height, s.depth) == (10, 10, 3)
    info = s.info()
    assert info["bvh_depth"] == 16 and info["split"] == yahr_amd.SPLIT_MIDPOINT and info["n_objects"] == 1
    a = s.arrays()
    # smooth = [True, False]: first triangle takes the vertex normals, second the flat normal (Scene.hs:70-78)
    orig = list(a["prim_orig"])
    i0, i1 = orig.index(0), orig.index(1)
    assert np.allclose(a["n0"][i0], [0, 1, 0]) and np.allclose(a["n1"][i0], [0, 1, 0])
    p0, e1, e2 = a["prim_p0"][i1], a["prim_e1"][i1], a["prim_e2"][i1]
    flat = np.cross(e2.astype(np.float64), e1.astype(np.float64))
    flat /= np.linalg.norm(flat)
    assert np.allclose(a["n0"][i1], flat, atol=1e-6)
    assert np.allclose(p0, [-1000, -2.2, -1000])
    # same scene through the neutral description
    b = scenes.test_yahr_scene().to_scene().build().arrays()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_text_and_programmatic_paths_agree_bitwise():
    for d in (scenes.scene_fixed(), scenes.spheres8(64, 36), scenes.pseudo_bunny(32, 18, n_lon=24, n_lat=13)):
        a = yahr_amd.Scene.parse(d.to_text()).build().arrays()
        b = d.to_scene().build().arrays()
        for k in a:
            assert np.array_equal(np.asarray(a[k]).view(np.uint32) if np.asarray(a[k]).dtype == np.float32 else a[k],
                                  np.asarray(b[k]).view(np.uint32) if np.asarray(b[k]).dtype == np.float32 else b[k]), k


def stale_text():
    """the dialect of the bundled scene.yahrr (SURVEY finding 2): PointLight with one argument and a
    TriangleMesh without the Normals/Smooth fields"""
    return MIN_SCENE % (
        MAT, 'PointLight (Vec3 4 10 -10)',
        'TriangleMesh { triangleMeshPoints = [Vec3 -1000 -2.2 -1000, Vec3 1000 -2.2 -1000, Vec3 1000 -2.2 1000], '
        'triangleMeshTriangles = [(0, 1, 2)], materialId = "m" }')


def test_stale_dialect_is_rejected_strictly_and_accepted_leniently():
    with pytest.raises(yahr_amd.YahrError) as e:
        yahr_amd.Scene.parse(stale_text())
    assert e.value.code == -2 and "no parse" in str(e.value)
    s = yahr_amd.Scene.parse(stale_text(), lenient=True).build()
    assert s.n_lights == 1 and s.flat.contents.n_prims == 1
    assert np.allclose(s.arrays()["light_spectrum"][0], [100, 100, 100])


@pytest.mark.skipif(not os.path.exists("/root/reference/scene.yahrr"), reason="reference checkout not present")
def test_bundled_scene_file_in_place():
    with pytest.raises(yahr_amd.YahrError):
        yahr_amd.Scene.load("/root/reference/scene.yahrr")
    s = yahr_amd.Scene.load("/root/reference/scene.yahrr", lenient=True).build()
    assert (s.width, s.height, s.depth) == (1024, 768, 3) and s.flat.contents.n_prims == 9
    fixed = yahr_amd.Scene.load(os.path.join(os.path.dirname(HERE), "scenes", "scene_fixed.yahrr")).build()
    a, b = s.arrays(), fixed.arrays()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_number_forms():
    s = parse('Sphere { position = Vec3 (-2.5) -0.5 1e1, radius = 2, materialId = "m" }, '
              'Sphere { position = (Vec3 (((1))) 0x10 1.5e-1), radius = (((3.25))), materialId = ("m") }')
    a = s.arrays()
    o = list(a["prim_orig"])
    assert np.array_equal(a["prim_p0"][o.index(0)], np.array([-2.5, -0.5, 10], dtype=np.float32))
    assert np.array_equal(a["prim_p0"][o.index(1)], np.array([1, 16, 0.15], dtype=np.float32))
    assert a["prim_e1"][o.index(1)][0] == np.float32(3.25)


@pytest.mark.parametrize("bad", [
    'Sphere { position = Vec3 0 0 .5, radius = 1, materialId = "m" }',           # ".5" is not a Haskell literal
    'Sphere { position = Vec3 0 0 5., radius = 1, materialId = "m" }',           # nor is "5."
    'Sphere { radius = 1, position = Vec3 0 0 5, materialId = "m" }',            # fields out of order
    'Sphere { position = Vec3 0 0 5, radius = 1 }',                               # missing field
    'Sphere { position = Vec3 0 0 5, radius = 1, materialId = "m", }',            # trailing comma
    'Sphere { position = Vec3 0 0, radius = 1, materialId = "m" }',               # too few components
    'Cube { position = Vec3 0 0 5 }',                                             # unknown constructor
    'Sphere { position = Vec3 0 0 5, radius = 1, materialId = m }',               # unquoted string
    'TriangleMesh { triangleMeshPoints = [], triangleMeshNormals = Just Vec3 0 0 1, triangleMeshTriangles = [], '
    'triangleMeshSmooth = Nothing, materialId = "m" }',                           # Just needs a list
    'TriangleMesh { triangleMeshPoints = [], triangleMeshNormals = Nothing, triangleMeshTriangles = [(0, 1.5, 2)], '
    'triangleMeshSmooth = Nothing, materialId = "m" }',                           # fractional Int
])
def test_malformed_objects_are_no_parse(bad):
    with pytest.raises(yahr_amd.YahrError) as e:
        parse(bad)
    assert e.value.code == -2


def test_precedence_rules():
    # a constructor application as a constructor ARGUMENT needs parentheses (derived Read, precedence 11)
    with pytest.raises(yahr_amd.YahrError):
        parse(lights="PointLight Vec3 0 1 0 Vec3 1 1 1")
    assert parse(lights="PointLight (Vec3 0 1 0) (Vec3 1 1 1)").n_lights == 1
    assert parse(lights="(((PointLight (Vec3 0 1 0) ((Vec3 1 1 1)))))").n_lights == 1
    # list elements and record fields are read at precedence 0: no parentheses needed
    assert parse(lights="PointLight (Vec3 0 1 0) (Vec3 1 1 1), PointLight (Vec3 0 2 0) (Vec3 1 1 1)").n_lights == 2
    for cm in ("BVH 3 Midpoint", "(BVH 3 Midpoint)", "BVH (3) (SurfaceAreaHeuristic)", "((BVH 3 SurfaceAreaHeuristic))"):
        t = (MIN_SCENE % (MAT, "", "")).replace("BVH 16 Midpoint", cm)
        assert yahr_amd.Scene.parse(t).info()["bvh_depth"] == 3
    with pytest.raises(yahr_amd.YahrError):
        yahr_amd.Scene.parse((MIN_SCENE % (MAT, "", "")) + " trailing")
    with pytest.raises(yahr_amd.YahrError):
        yahr_amd.Scene.parse("")


def test_string_escapes_and_material_map():
    mats = (MAT + ', BlinnPhongMaterial { id = "we\\"ird\\\\id", ambient = Vec3 0 0 0, diffuse = Vec3 0.5 0.25 0.125, '
            'specular = Vec3 0 0 0, shininess = 2 }, '
            # duplicate id: Data.Map.fromList keeps the LAST (main.hs:57)
            'BlinnPhongMaterial { id = "m", ambient = Vec3 0 0 0, diffuse = Vec3 0.75 0.75 0.75, specular = Vec3 0 0 0, shininess = 9 }')
    s = parse('Sphere { position = Vec3 0 0 5, radius = 1, materialId = "we\\"ird\\\\id" }, '
              'Sphere { position = Vec3 3 0 5, radius = 1, materialId = "m" }, '
              'Sphere { position = Vec3 6 0 5, radius = 1, materialId = "nope" }', mats=mats)
    a = s.arrays()
    o = list(a["prim_orig"])
    assert a["prim_mat"][o.index(0)] == 1 and a["prim_mat"][o.index(1)] == 2
    assert a["prim_mat"][o.index(2)] == -1       # unknown id: only an error if it is ever shaded (Map.!, main.hs:55)
    assert a["mat_shininess"][2] == 9


def test_expand_rules_match_oracle():
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, size=(40, 3)).astype(np.float32)
    nrm = rng.uniform(-1, 1, size=(40, 3)).astype(np.float32)
    tris = rng.integers(0, 40, size=(120, 3)).astype(np.int32)
    smooth = list(rng.integers(0, 2, size=97).astype(bool))       # shorter than the triangle list: truncates
    base = scenes.scene_fixed(16, 16)
    cases = {
        "normals+smooth(trunc)": ("mesh", pts, nrm, tris, smooth, "sphere"),
        "normals, smooth Nothing": ("mesh", pts, nrm, tris, None, "sphere"),
        "no normals, smooth given": ("mesh", pts, None, tris, smooth, "sphere"),
        "subsampled": ("subsampled", 0.303, [("mesh", pts, nrm, tris, None, "sphere"), ("sphere", (0, 0, 3), 1.0, "bottom"),
                                             ("subsampled", 0.5, [("mesh", pts, None, tris, None, "back")])]),
    }
    for name, obj in cases.items():
        d = base.with_(objects=[obj])
        for scene in (yahr_amd.Scene.parse(d.to_text()), d.to_scene()):
            a = scene.build().arrays()
            op, om = OracleScene.from_desc(d).prims()
            order = a["prim_orig"]
            assert len(order) == op.shape[0], name
            inv = np.argsort(order)
            assert np.array_equal(a["prim_kind"][inv], op[:, 0].astype(np.int32)), name
            tri = op[:, 0] == 0
            assert np.array_equal(a["prim_p0"][inv][tri].view(np.uint32), op[tri, 1:4].view(np.uint32)), name
            assert np.array_equal(a["n0"][inv][tri].view(np.uint32), op[tri, 10:13].view(np.uint32)), name
            assert np.array_equal(a["n2"][inv][tri].view(np.uint32), op[tri, 16:19].view(np.uint32)), name
            assert np.array_equal(a["prim_mat"][inv], om), name
    assert len(yahr_amd.Scene.parse(base.with_(objects=[cases["normals+smooth(trunc)"]]).to_text()).build().arrays()["prim_orig"]) == 97


def test_mesh_index_out_of_bounds_is_an_error():
    with pytest.raises(yahr_amd.YahrError):
        parse('TriangleMesh { triangleMeshPoints = [Vec3 0 0 0], triangleMeshNormals = Nothing, '
              'triangleMeshTriangles = [(0, 0, 1)], triangleMeshSmooth = Nothing, materialId = "m" }')


def test_image_size_is_floored():
    t = (MIN_SCENE % (MAT, "", "")).replace("imW = 8", "imW = 8.9").replace("imH = 6", "imH = 6.5")
    s = yahr_amd.Scene.parse(t).build()
    assert (s.width, s.height) == (8, 6) and abs(s.info()["imW"] - 8.9) < 1e-6      # main.hs:122-123


def test_missing_file():
    with pytest.raises(yahr_amd.YahrError) as e:
        yahr_amd.Scene.load("/nonexistent/scene.yahrr")
    assert e.value.code == -3


# ---- numeric literals: decimal -> binary32 must be CORRECTLY rounded (GHC reads a Rational, then fromRational) ----
def _parse_f32(lexeme):
    import ctypes as C
    from yahr_amd import _lib
    out = C.c_float()
    _lib.check(_lib.lib.yahr_parse_float32(lexeme.encode(), len(lexeme), C.byref(out)))
    return np.float32(out.value)


def _round_to_f32(frac):
    """correctly rounded binary32 of an exact Fraction > 0 (normal range), ties to even -- by exact comparison"""
    from fractions import Fraction
    f = np.float32(float(frac))                       # a nearby float (double rounding may be off by one ulp)
    cands = sorted({float(np.nextafter(f, np.float32(0))), float(f), float(np.nextafter(f, np.float32(np.inf)))})
    best = min(cands, key=lambda c: (abs(Fraction(c) - frac), int(np.float32(c).view(np.uint32)) & 1))
    return np.float32(best)


def test_float_literals_round_trip_and_round_correctly():
    from fractions import Fraction
    rng = np.random.default_rng(5)
    bits = rng.integers(0x00800000, 0x7F7FFFFF, 20000, dtype=np.uint32)       # normal positive floats, all exponents
    vals = np.concatenate([bits.view(np.float32), rng.uniform(-1000, 1000, 5000).astype(np.float32).__abs__()])
    for v in vals:
        for fmt in ("%.9g", "%.8e", "%.17g"):                                   # every one identifies v uniquely
            s = fmt % float(v)
            assert _parse_f32(s).view(np.uint32) == np.float32(v).view(np.uint32), (s, v)
    # decimals just below / at / just above the midpoint of two adjacent floats, short enough for the fast path
    for v in bits.view(np.float32)[:3000]:
        up = np.nextafter(v, np.float32(np.inf))
        mid = (Fraction(float(v)) + Fraction(float(up))) / 2
        for digits in (9, 12, 15):
            e = int(np.floor(np.log10(float(v))))
            scale = Fraction(10) ** (digits - 1 - e)
            for k in (-1, 0, 1):
                n = int(mid * scale) + k                                        # digits-long mantissa around the midpoint
                s = "%de%d" % (n, e - digits + 1)
                want = _round_to_f32(Fraction(n) / scale)
                assert _parse_f32(s).view(np.uint32) == want.view(np.uint32), (s, float(v))
    # the exact midpoint 2^24 + 1 ties to even; a hair above it rounds up
    assert _parse_f32("16777217") == np.float32(16777216.0)
    assert _parse_f32("16777217.000001") == np.float32(16777218.0)
    assert _parse_f32("0x1F") == np.float32(31) and _parse_f32("0o17") == np.float32(15)
    assert _parse_f32("1e-45") == np.float32(1e-45) and _parse_f32("1e39") == np.float32(np.inf)
    assert _parse_f32("0.000") == np.float32(0) and _parse_f32("000123.4500") == np.float32(123.45)
