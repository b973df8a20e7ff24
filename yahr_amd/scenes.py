"""Scene descriptions for tests and benchmarks: a neutral in-memory form (SceneDesc) that can be written as
.yahrr text (the reference's derived-Read syntax, Scene.hs:15-58), fed to the C ABI programmatically
(yahr_amd.Scene), or fed to the oracle by the tests.  Also the synthetic workloads of BASELINE.json:

    scene_fixed()      config 1: the bundled scene.yahrr, repaired (SURVEY finding 2), 256x256
    spheres8()         config 2: 8 spheres + plane + 2 point lights, 1920x1080, depth 1
    pseudo_bunny()     configs 3/4: procedural 69,696-triangle closed mesh on a plane (NO Stanford bunny
                       mesh ships with the reference; this stand-in is labelled "pseudo-bunny" everywhere)
    random_tris()      config 5: 1M random triangles, splitmix64 seed 0x5EED, BVH 20 Midpoint

All numbers are rounded to binary32 before use so that text and programmatic paths agree bit for bit.
"""
import numpy as np

F = np.float32


def _v(x):
    return tuple(float(F(c)) for c in x)


class SceneDesc:
    def __init__(self, depth, bvh_depth, split, camera, materials, lights, objects):
        self.depth = int(depth)
        self.bvh_depth = int(bvh_depth)
        self.split = split                  # "Midpoint" | "SurfaceAreaHeuristic"
        self.camera = camera                # dict imW imH focal look up pos
        self.materials = materials          # list of dict id ambient diffuse specular shininess
        self.lights = lights                # list of (pos, spectrum)
        self.objects = objects              # list of ("sphere", c, r, mat) | ("triangle", p0,p1,p2,n0,n1,n2, mat)
        #                                      | ("mesh", points, normals|None, tris, smooth|None, mat)
        #                                      | ("subsampled", size, [objects])

    # ---- .yahrr text -----------------------------------------------------------------------------
    @staticmethod
    def _num(x):
        x = float(F(x))
        if x != x:
            return "NaN"
        if x in (float("inf"), float("-inf")):
            return "Infinity" if x > 0 else "(-Infinity)"
        s = "%.9g" % x                      # round-trips binary32; forms 12, 1.5, 1e+06, 1.25e-05 all lex in Haskell
        return "(%s)" % s if s.startswith("-") else s

    @classmethod
    def _vec(cls, v):
        return "Vec3 %s %s %s" % tuple(cls._num(c) for c in v)

    @staticmethod
    def _str(s):
        return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'

    def _obj_text(self, o):
        k = o[0]
        if k == "sphere":
            return "Sphere { position = %s, radius = %s, materialId = %s }" % (self._vec(o[1]), self._num(o[2]), self._str(o[3]))
        if k == "triangle":
            names = ["p0", "p1", "p2", "n0", "n1", "n2"]
            body = ", ".join("%s = %s" % (n, self._vec(v)) for n, v in zip(names, o[1:7]))
            return "Triangle { %s, materialId = %s }" % (body, self._str(o[7]))
        if k == "mesh":
            _, pts, nrm, tris, smooth, mat = o
            pts_t = ",".join(self._vec(p) for p in np.asarray(pts, dtype=F).reshape(-1, 3))
            nrm_t = "Nothing" if nrm is None else "Just [%s]" % ",".join(self._vec(p) for p in np.asarray(nrm, dtype=F).reshape(-1, 3))
            tri_t = ",".join("(%d,%d,%d)" % tuple(t) for t in np.asarray(tris, dtype=np.int64).reshape(-1, 3))
            sm_t = "Nothing" if smooth is None else "Just [%s]" % ",".join("True" if b else "False" for b in smooth)
            return ("TriangleMesh { triangleMeshPoints = [%s], triangleMeshNormals = %s, triangleMeshTriangles = [%s], "
                    "triangleMeshSmooth = %s, materialId = %s }") % (pts_t, nrm_t, tri_t, sm_t, self._str(mat))
        if k == "subsampled":
            return "Subsampled { subsampleSize = %s, subsampledObjects = [%s] }" % (
                self._num(o[1]), ", ".join(self._obj_text(c) for c in o[2]))
        raise ValueError(k)

    def to_text(self):
        c = self.camera
        mats = ",\n    ".join(
            "BlinnPhongMaterial { id = %s, ambient = %s, diffuse = %s, specular = %s, shininess = %s }" % (
                self._str(m["id"]), self._vec(m["ambient"]), self._vec(m["diffuse"]), self._vec(m["specular"]),
                self._num(m["shininess"])) for m in self.materials)
        lights = ", ".join("PointLight (%s) (%s)" % (self._vec(p), self._vec(s)) for p, s in self.lights)
        objs = ",\n    ".join(self._obj_text(o) for o in self.objects)
        return ("Scene {\n  integrator = WhittedIntegrator { recursionDepth = %d },\n  cullingMode = BVH %d %s,\n"
                "  camera = Camera { imW = %s, imH = %s, focalLength = %s, lookDir = %s, upDir = %s, position = %s },\n"
                "  materials = [\n    %s\n  ],\n  lights = [ %s ],\n  objects = [\n    %s\n  ]\n}\n") % (
            self.depth, self.bvh_depth, self.split, self._num(c["imW"]), self._num(c["imH"]), self._num(c["focal"]),
            self._vec(c["look"]), self._vec(c["up"]), self._vec(c["pos"]), mats, lights, objs)

    # ---- programmatic path through the C ABI ---------------------------------------------------------
    def to_scene(self):
        import yahr_amd
        s = yahr_amd.Scene()
        c = self.camera
        s.set_camera(F(c["imW"]), F(c["imH"]), F(c["focal"]), _v(c["look"]), _v(c["up"]), _v(c["pos"]))
        s.set_integrator(self.depth)
        s.set_culling(self.bvh_depth, yahr_amd.SPLIT_SAH if self.split == "SurfaceAreaHeuristic" else yahr_amd.SPLIT_MIDPOINT)
        for m in self.materials:
            s.add_material(m["id"], _v(m["ambient"]), _v(m["diffuse"]), _v(m["specular"]), F(m["shininess"]))
        for p, sp in self.lights:
            s.add_light(_v(p), _v(sp))

        def add(o):
            k = o[0]
            if k == "sphere":
                s.add_sphere(_v(o[1]), F(o[2]), o[3])
            elif k == "triangle":
                s.add_triangle(*[_v(v) for v in o[1:7]], o[7])
            elif k == "mesh":
                s.add_mesh(o[1], o[2], o[3], o[4], o[5])
            elif k == "subsampled":
                s.begin_subsampled(F(o[1]))
                for ch in o[2]:
                    add(ch)
                s.end_subsampled()
        for o in self.objects:
            add(o)
        return s

    def with_camera(self, **kw):
        cam = dict(self.camera)
        cam.update(kw)
        return SceneDesc(self.depth, self.bvh_depth, self.split, cam, self.materials, self.lights, self.objects)

    def with_(self, **kw):
        d = dict(depth=self.depth, bvh_depth=self.bvh_depth, split=self.split, camera=self.camera,
                 materials=self.materials, lights=self.lights, objects=self.objects)
        d.update(kw)
        return SceneDesc(**d)


def _mat(id, ambient, diffuse, specular, shininess):
    return dict(id=id, ambient=ambient, diffuse=diffuse, specular=specular, shininess=shininess)


def plane_mesh(y, mat, half=1000.0, normals=False):
    pts = [(-half, y, -half), (half, y, -half), (half, y, half), (-half, y, half)]
    nrm = [(0, 1, 0)] * 4 if normals else None
    return ("mesh", np.array(pts, dtype=F), None if nrm is None else np.array(nrm, dtype=F),
            np.array([(0, 1, 2), (0, 2, 3)], dtype=np.int32), None, mat)


_SCENE_YAHRR_MATS = [
    _mat("sphere", (0.5, 0, 0), (0.2, 0.2, 0.2), (1, 1, 1), 10),
    _mat("back", (0, 0.3, 0), (0, 1, 0), (0, 0.1, 0), 1),
    _mat("side", (0.3, 0, 0), (1, 0, 0), (0.1, 0, 0), 1),
    _mat("bottom", (0.3, 0.3, 0.3), (1, 1, 1), (0.1, 0.1, 0.1), 1),
]
_SCENE_YAHRR_SPHERES = [(-2.4, -1.2, 20), (0, -1.2, 20), (2.4, -1.2, 20), (-1.2, -1.2, 15), (2, -1.2, 15),
                        (0, -1.2, 10), (3, -1.2, 10)]


def scene_fixed(width=256, height=256):
    """The bundled scene.yahrr (scene.yahrr:1-67) with the two stale constructs repaired: the point light
    gets an explicit spectrum -- (100,100,100) is this build's documented guess -- and the plane mesh gets
    triangleMeshNormals = Nothing / triangleMeshSmooth = Nothing.  Camera size overridden (config 1)."""
    cam = dict(imW=width, imH=height, focal=1.5, look=(0.4, -0.3, 1), up=(0, 1, 0), pos=(-4, 3, 2))
    objs = [("sphere", c, 1.0, "sphere") for c in _SCENE_YAHRR_SPHERES] + [plane_mesh(-2.2, "bottom")]
    return SceneDesc(3, 16, "Midpoint", cam, _SCENE_YAHRR_MATS, [((4, 10, -10), (100, 100, 100))], objs)


def spheres8(width=1920, height=1080, depth=1):
    """config 2: the 7 spheres of scene.yahrr + one more, plane, 2 point lights, direct lighting."""
    cam = dict(imW=width, imH=height, focal=1.5, look=(0.4, -0.3, 1), up=(0, 1, 0), pos=(-4, 3, 2))
    cs = _SCENE_YAHRR_SPHERES + [(-3, -1.2, 10)]
    objs = [("sphere", c, 1.0, "sphere") for c in cs] + [plane_mesh(-2.2, "bottom")]
    lights = [((4, 10, -10), (100, 100, 100)), ((-6, 8, 0), (100, 100, 100))]
    return SceneDesc(depth, 16, "Midpoint", cam, _SCENE_YAHRR_MATS, lights, objs)


def test_yahr_scene():
    """The scene of the reference's end-to-end smoke test (compat/test_yahr.py:11-52), as data."""
    cam = dict(imW=10, imH=10, focal=5, look=(0, -0.1, -1), up=(0, 1, 0), pos=(0, 0.2, 1))
    mesh = ("mesh", np.array([(-1000, -2.2, -1000), (1000, -2.2, -1000), (1000, -2.2, 1000), (-1000, -2.2, 1000)], dtype=F),
            np.array([(0, 1, 0)] * 4, dtype=F), np.array([(0, 1, 2), (0, 2, 3)], dtype=np.int32), [True, False], "testmat")
    return SceneDesc(3, 16, "Midpoint", cam, [_mat("testmat", (0, 0, 0), (0.8, 0.8, 0.8), (0, 0, 0), 1)],
                     [((4, 10, 10), (1, 1, 1))], [mesh])


def pseudo_bunny_mesh(n_lon=264, n_lat=133):
    """Closed star-shaped mesh, 2*n_lon*(n_lat-1) triangles (69,696 by default): a body ellipsoid with a head
    lobe and two ear lobes plus low-frequency bumps, fixed formula, no RNG.  Smooth vertex normals
    (area-weighted face normals, normalised) as the Blender exporter writes them (render_engine.py:79-88)."""
    th = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]                  # ring polar angles
    ph = np.arange(n_lon) * (2.0 * np.pi / n_lon)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], axis=-1).reshape(-1, 3)
    d = np.concatenate([[[0.0, 1.0, 0.0]], d, [[0.0, -1.0, 0.0]]], axis=0)

    def unit(v):
        v = np.asarray(v, dtype=np.float64)
        return v / np.linalg.norm(v)

    def lobe(dirv, amp, k):
        return amp * np.exp(k * (d @ unit(dirv) - 1.0))
    r = (1.0 + lobe((0.75, 0.55, 0.0), 0.55, 9.0)                 # head
         + lobe((0.55, 0.80, 0.22), 0.95, 60.0) + lobe((0.55, 0.80, -0.22), 0.95, 60.0)   # ears
         + lobe((-0.9, -0.25, 0.0), 0.18, 25.0)                   # tail
         + 0.03 * np.sin(5.0 * np.arctan2(d[:, 2], d[:, 0])) * np.sin(4.0 * np.arccos(np.clip(d[:, 1], -1, 1)))
         + 0.012 * np.sin(17.0 * d[:, 0] + 3.0) * np.sin(13.0 * d[:, 1]) * np.sin(15.0 * d[:, 2] + 1.0))
    pts = d * r[:, None] * np.array([1.25, 1.0, 0.9])              # ellipsoid body
    nv = pts.shape[0]
    tris = []
    ring = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        tris.append((0, ring(0, j + 1), ring(0, j)))
    for i in range(n_lat - 2):
        for j in range(n_lon):
            a, b, c, e = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            tris.append((a, b, e))
            tris.append((a, e, c))
    for j in range(n_lon):
        tris.append((nv - 1, ring(n_lat - 2, j), ring(n_lat - 2, j + 1)))
    tris = np.array(tris, dtype=np.int32)
    fn = np.cross(pts[tris[:, 1]] - pts[tris[:, 0]], pts[tris[:, 2]] - pts[tris[:, 0]])
    centre = pts[tris].mean(axis=1)
    flip = (fn * centre).sum(axis=1) < 0                            # make every face normal point outward
    fn[flip] *= -1.0
    vn = np.zeros_like(pts)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn)
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    return pts.astype(F), vn.astype(F), tris


def pseudo_bunny(width=1920, height=1080, depth=2, n_lon=264, n_lat=133, bvh_depth=16, split="Midpoint"):
    """configs 3/4: pseudo-bunny on the +-1000 plane, one light, framed like bunny2.png."""
    pts, vn, tris = pseudo_bunny_mesh(n_lon, n_lat)
    pts = pts.astype(np.float64)
    pts[:, 1] -= pts[:, 1].min()                                    # sit on the plane y = 0
    pts = pts.astype(F)
    mats = [_mat("bunny", (0, 0, 0), (0.55, 0.55, 0.55), (0.25, 0.25, 0.25), 12),
            _mat("floor", (0, 0, 0), (0.08, 0.3, 0.12), (0.35, 0.35, 0.35), 40)]
    cam = dict(imW=width, imH=height, focal=2, look=(0.22, -0.2, 1), up=(0, 1, 0), pos=(-1.2, 2.1, -6.2))
    mesh = ("mesh", pts, vn, tris, [True] * len(tris), "bunny")
    return SceneDesc(depth, bvh_depth, split, cam, mats, [((-5, 9, -6), (90, 90, 90))], [mesh, plane_mesh(0.0, "floor")])


def _splitmix64(n, seed):
    """n outputs of splitmix64 started at `seed` (vectorised)"""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_tris(n=1_000_000, width=1920, height=1080, depth=2, bvh_depth=20, seed=0x5EED):
    """config 5: n triangles, centre uniform in [-10,10]^3, vertices = centre + uniform
    [-0.05,0.05]^3, flat normals (triangleMeshNormals = Nothing), one material, one light."""
    u = (_splitmix64(12 * n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    u = u.reshape(n, 12)
    centre = u[:, 0:3] * 20.0 - 10.0
    verts = centre[:, None, :] + (u[:, 3:12].reshape(n, 3, 3) * 0.1 - 0.05)
    pts = verts.reshape(-1, 3).astype(F)
    tris = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    cam = dict(imW=width, imH=height, focal=1.5, look=(0, 0, 1), up=(0, 1, 0), pos=(0, 0, -30))
    mats = [_mat("tri", (0, 0, 0), (0.7, 0.7, 0.7), (0.3, 0.3, 0.3), 8)]
    mesh = ("mesh", pts, None, tris, None, "tri")
    return SceneDesc(depth, bvh_depth, "Midpoint", cam, mats, [((0, 20, -20), (400, 400, 400))], [mesh])


def spec_sphere_grid(split="Midpoint"):
    """The 200-sphere grid of the reference's BVH known-answer test (test/Spec.hs:221-262)."""
    objs = [("sphere", (i * 10.0, j * 10.0, k * 10.0), 1.0, "m") for i in range(1, 11) for j in range(1, 11) for k in (1, 2)]
    cam = dict(imW=64, imH=64, focal=1, look=(0, 0, 1), up=(0, 1, 0), pos=(55, 55, -60))
    return SceneDesc(1, 16, split, cam, [_mat("m", (0, 0, 0), (0.6, 0.6, 0.6), (0.4, 0.4, 0.4), 10)],
                     [((55, 55, -40), (3000, 3000, 3000))], objs)
