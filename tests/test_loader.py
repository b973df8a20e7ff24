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


# ---- long lists parsed by several threads (scene_parse.cpp: list_into / parallel_list / scan_list_parallel) ----------
def _arrays_equal(a, b):
    for k in a:
        x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
        assert x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)), k


def _parse_with(text, threads, monkeypatch, min_bytes="1"):
    monkeypatch.setenv("YAHR_PARSE_THREADS", str(threads))
    if threads > 1:
        monkeypatch.setenv("YAHR_PARSE_PARALLEL_MIN", min_bytes)
    else:
        monkeypatch.delenv("YAHR_PARSE_PARALLEL_MIN", raising=False)
    return yahr_amd.Scene.parse(text)


def _parallel_lists():
    return yahr_amd.lib.yahr_debug_parallel_lists()


def _tricky_scene_text(rng, n_records=150):
    """objects that exercise the bracket scan: material ids with brackets, commas, escaped quotes and backslashes,
    redundant parentheses, a mesh between loose records, a nested Subsampled list"""
    ids = ['m', 'a,b', ']', '[(', 'q\\"uote', 'back\\\\slash', 'end\\\\', '{x}', 'two\\\\\\\\', 'gap\\   \\ped']
    names = ['m', 'a,b', ']', '[(', 'q"uote', 'back\\slash', 'end\\', '{x}', 'two\\\\', 'gapped']
    mats = ", ".join('BlinnPhongMaterial { id = "%s", ambient = Vec3 0 0 0, diffuse = Vec3 %g 0.5 0.25, specular = Vec3 0 0 0, '
                     'shininess = %d }' % (i, 0.1 + 0.05 * k, k + 1) for k, i in enumerate(ids))

    def v3():
        x = rng.uniform(-3, 3, 3)
        return "Vec3 %s %s %s" % tuple(("(%r)" % float(np.float32(c))) if c < 0 or rng.random() < 0.2 else repr(float(np.float32(c))) for c in x)
    objs = []
    for k in range(n_records):
        i = ids[int(rng.integers(len(ids)))]
        r = rng.random()
        if r < 0.55:
            o = 'Triangle { p0 = %s, p1 = %s, p2 = %s, n0 = %s, n1 = %s, n2 = %s, materialId = "%s" }' % (v3(), v3(), v3(), v3(), v3(), v3(), i)
        elif r < 0.8:
            o = 'Sphere { position = (%s), radius = %r, materialId = ("%s") }' % (v3(), float(np.float32(rng.uniform(0.1, 1))), i)
        elif r < 0.9:
            nv = int(rng.integers(3, 40))
            tris = ",".join("(%d, %d,%d)" % tuple(rng.integers(0, nv, 3)) for _ in range(int(rng.integers(1, 60))))
            o = ('TriangleMesh { triangleMeshPoints = [%s], triangleMeshNormals = Nothing, triangleMeshTriangles = [%s], '
                 'triangleMeshSmooth = Nothing, materialId = "%s" }' % (",".join(v3() for _ in range(nv)), tris, i))
        else:
            inner = ", ".join('Sphere { position = %s, radius = 0.25, materialId = "%s" }' % (v3(), i) for _ in range(int(rng.integers(1, 30))))
            o = 'Subsampled { subsampleSize = 0.5, subsampledObjects = [%s] }' % inner
        if rng.random() < 0.15:
            o = "((%s))" % o
        objs.append(o)
    return MIN_SCENE % (mats, "PointLight (Vec3 0 5 0) (Vec3 50 50 50)", ",\n ".join(objs)), names


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_parallel_lists_give_the_sequential_scene(seed, monkeypatch):
    rng = np.random.default_rng(4000 + seed)
    text, _ = _tricky_scene_text(rng)
    ref = _parse_with(text, 1, monkeypatch).build().arrays()
    assert _parallel_lists() == 0
    for threads in (2, 3, 5, 16):
        sc = _parse_with(text, threads, monkeypatch)
        assert _parallel_lists() >= 1                    # the objects list at least (150 elements >= 4 x threads)
        _arrays_equal(ref, sc.build().arrays())
    # one big mesh (lists of points, normals, index triples, smooth flags) and the bunny file
    d = scenes.pseudo_bunny(32, 18, n_lon=40, n_lat=21)
    ref = _parse_with(d.to_text(), 1, monkeypatch).build().arrays()
    for threads in (2, 7):
        sc = _parse_with(d.to_text(), threads, monkeypatch)
        assert _parallel_lists() >= 3                    # points, normals, index triples (, smooth flags)
        _arrays_equal(ref, sc.build().arrays())
        # a threshold between the sizes of the lists: the long ones by threads, the short ones not
        sc = _parse_with(d.to_text(), threads, monkeypatch, min_bytes="20000")
        assert 1 <= _parallel_lists() <= 3
        _arrays_equal(ref, sc.build().arrays())


def test_parallel_lists_report_the_sequential_error(monkeypatch):
    """whatever goes wrong inside a list parsed by threads, the error (message and byte offset) is the sequential
    parser's: syntax errors in the middle and at the end of a long list, an unterminated string, unbalanced brackets,
    a missing closing bracket, a string gap (which the chunked bracket scan refuses)"""
    rng = np.random.default_rng(7)
    good, _ = _tricky_scene_text(rng, 120)
    at = good.index("objects = [") + len("objects = [")
    mid = good.index("},", at + len(good[at:]) // 2) + 1
    variants = [good[:mid] + ", Sphere { position = Vec3 0 0, radius = 1, materialId = \"m\" }" + good[mid:],   # too few components
                good[:mid] + ", Cube { }" + good[mid:],
                good[:mid] + ", " + good[mid:],                                                                   # empty element
                good[:mid] + ")" + good[mid:],                                                                    # stray bracket
                good[:mid] + ", Sphere { position = Vec3 0 0 1, radius = 1, materialId = \"m }" + good[mid:],     # unterminated string
                good.rstrip()[:-3],                                                                                # no closing ] }
                good[:mid] + " \\ " + good[mid:]]                                                                 # backslash outside a string
    for text in variants:
        with pytest.raises(yahr_amd.YahrError) as e1:
            _parse_with(text, 1, monkeypatch)
        for threads in (2, 5):
            with pytest.raises(yahr_amd.YahrError) as e2:
                _parse_with(text, threads, monkeypatch)
            assert str(e2.value) == str(e1.value)


def test_mesh_index_out_of_bounds_is_reported_for_the_first_offending_triangle():
    """expand (Scene.hs:62-79) indexes the point and normal lists with `!!`: a bad index is an error at build time.
    Big meshes are expanded by several threads; the message names what the FIRST bad triangle got wrong."""
    def mesh(n_tris, bad):
        tris = [(0, 1, 2)] * n_tris
        smooth = ["False"] * n_tris
        for k, what in bad:
            if what == "points":
                tris[k] = (0, 9, 2)
            else:
                tris[k] = (0, 1, 3); smooth[k] = "True"                # index 3: a point, but no normal
        obj = ('TriangleMesh { triangleMeshPoints = [Vec3 0 0 0, Vec3 1 0 0, Vec3 0 1 0, Vec3 0 0 1], '
               'triangleMeshNormals = Just [Vec3 0 0 1, Vec3 0 0 1, Vec3 0 0 1], triangleMeshTriangles = [%s], '
               'triangleMeshSmooth = Just [%s], materialId = "m" }' % (",".join("(%d,%d,%d)" % t for t in tris), ",".join(smooth)))
        return yahr_amd.Scene.parse(MIN_SCENE % (MAT, "", obj))
    mesh(10, []).build()
    for n in (10, 200_000):                                            # sequential and threaded expansion
        for bad, word in (([(n - 1, "points")], "points"), ([(n // 2, "normals")], "normals"),
                          ([(n // 3, "normals"), (n - 2, "points")], "normals"), ([(3, "points"), (n - 1, "normals")], "points")):
            with pytest.raises(yahr_amd.YahrError) as e:
                mesh(n, bad).build()
            assert "out of bounds (%s)" % word in str(e.value), (n, bad)
        # smooth = False: the normals are not looked at (lazy `if`, Scene.hs:70-78)
        s = mesh(n, [])
        s.build()
        assert s.arrays()["prim_p0"].shape[0] == n


def test_default_settings_parse_the_bunny_file_by_threads(monkeypatch):
    """no environment knobs: the 4.8 MB pseudo-bunny text is above the megabyte threshold, so its point, normal and
    index lists go to the threaded path with the host's core count -- and must give the scene of the programmatic path"""
    d = scenes.pseudo_bunny()
    text = d.to_text()
    assert len(text) > (1 << 20)
    for k in ("YAHR_PARSE_THREADS", "YAHR_PARSE_PARALLEL_MIN"):
        monkeypatch.delenv(k, raising=False)
    sc = yahr_amd.Scene.parse(text)
    if (os.cpu_count() or 1) > 1:
        assert _parallel_lists() >= 2
    _arrays_equal(sc.build().arrays(), d.to_scene().build().arrays())
