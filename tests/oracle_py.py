"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- the product never imports this).

`OracleScene.from_desc(desc)` feeds a yahr_amd.scenes.SceneDesc to the oracle directly, without going
through the product's parser or builder, so oracle and product stay independent."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
KAT_PATH = os.path.join(ORACLE_DIR, "kat")


def build_oracle():
    src = [os.path.join(ORACLE_DIR, f) for f in ("yahr_oracle.c", "yahr_oracle.h", "kat.c", "Makefile")]
    if (not os.path.exists(LIB_PATH) or not os.path.exists(KAT_PATH)
            or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)):
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


class OStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary_rays", "reflection_rays", "shadow_rays", "box_tests",
                                          "tri_tests", "sphere_tests", "shaded_hits")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        L.oracle_scene_new.restype = C.c_void_p
        sig = {
            "oracle_scene_free": (None, [C.c_void_p]),
            "oracle_scene_set_camera": (None, [C.c_void_p, C.c_float, C.c_float, C.c_float, _F, _F, _F]),
            "oracle_scene_set_integrator": (None, [C.c_void_p, C.c_int]),
            "oracle_scene_set_culling": (None, [C.c_void_p, C.c_int, C.c_int]),
            "oracle_scene_add_material": (C.c_int, [C.c_void_p, _F, _F, C.c_float]),
            "oracle_scene_add_material_raw_blinn": (C.c_int, [C.c_void_p, C.c_float]),
            "oracle_scene_add_light": (None, [C.c_void_p, _F, _F]),
            "oracle_scene_add_sphere": (C.c_int, [C.c_void_p, _F, C.c_float, C.c_int]),
            "oracle_scene_add_triangle": (C.c_int, [C.c_void_p, _F, _F, _F, _F, _F, _F, C.c_int]),
            "oracle_scene_add_mesh": (C.c_int, [C.c_void_p, C.c_int, _F, _F, C.c_int, _I, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
            "oracle_subsample_pick": (C.c_int, [C.c_int, C.c_float, _I]),
            "oracle_scene_n_prims": (C.c_int, [C.c_void_p]),
            "oracle_scene_get_prim": (None, [C.c_void_p, C.c_int, _F, _I]),
            "oracle_scene_build": (None, [C.c_void_p]),
            "oracle_set_shadow_any_hit": (None, [C.c_int]),
            "oracle_scene_size": (None, [C.c_void_p, _I, _I]),
            "oracle_render": (None, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, _F, C.POINTER(OStats)]),
            "oracle_render_rows": (None, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, _F, C.POINTER(OStats)]),
            "oracle_n_batches": (C.c_int, [C.c_int, C.c_int, C.c_int]),
            "oracle_batch_window": (None, [C.c_int, C.c_int, C.c_int, C.c_int, _I]),
            "oracle_camera_ray": (None, [C.c_void_p, C.c_float, C.c_float, _F, _F, _F]),
            "oracle_camera_matrices": (None, [C.c_void_p, _F, _F]),
            "oracle_scene_intersect": (C.c_int, [C.c_void_p, _F, _F, C.c_float, _F]),
            "oracle_scene_reachable": (C.c_int, [C.c_void_p, _F, _F]),
            "oracle_radiance": (None, [C.c_void_p, _F, _F, C.c_float, _F]),
            "oracle_bsdf_at": (None, [C.c_int, C.c_float, _F, _F, _F, _F, _F]),
            "oracle_blinn_phong_at": (None, [_F, _F, C.c_float, _F, _F, _F, _F, _F]),
            "oracle_sample_offset": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, _F, _F]),
            "oracle_tree_dump": (C.c_int, [C.c_void_p, _I, _F, _I, _I, _I, _I]),
            "oracle_collide_triangle": (C.c_int, [_F] * 8 + [C.c_float] + [_F] * 5),
            "oracle_bb_ray": (C.c_int, [_F, _F, _F, _F, C.c_float, _F]),
            "oracle_powf_array": (None, [C.c_int, _F, _F, _F]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def fp(a):
    return a.ctypes.data_as(_F)


class OracleScene:
    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_scene_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_scene_free(self.h)
            self.h = None

    @classmethod
    def from_desc(cls, d):
        s = cls()
        L, h = s.L, s.h
        c = d.camera
        L.oracle_scene_set_camera(h, np.float32(c["imW"]), np.float32(c["imH"]), np.float32(c["focal"]),
                                  f3(c["look"]), f3(c["up"]), f3(c["pos"]))
        L.oracle_scene_set_integrator(h, d.depth)
        L.oracle_scene_set_culling(h, d.bvh_depth, 1 if d.split == "SurfaceAreaHeuristic" else 0)
        mat_idx = {}
        for m in d.materials:                     # Data.Map.fromList: later duplicates win (main.hs:57)
            mat_idx[m["id"]] = L.oracle_scene_add_material(h, f3(m["diffuse"]), f3(m["specular"]), np.float32(m["shininess"]))
        for p, sp in d.lights:
            L.oracle_scene_add_light(h, f3(p), f3(sp))

        def add(o, sink):
            """append primitives of object o (after expand) to sink as tuples"""
            k = o[0]
            if k == "sphere":
                sink.append(("s", o[1], o[2], mat_idx[o[3]]))
            elif k == "triangle":
                sink.append(("t",) + tuple(o[1:7]) + (mat_idx[o[7]],))
            elif k == "mesh":
                sink.append(("m", o[1], o[2], o[3], o[4], mat_idx[o[5]]))
            elif k == "subsampled":
                # expand children into a scratch oracle scene, then keep what Scene.hs:80-85 picks
                tmp = cls()
                inner = []
                for ch in o[2]:
                    add(ch, inner)
                for it in inner:
                    tmp._emit(it)
                n = L.oracle_scene_n_prims(tmp.h)
                idx = (C.c_int32 * max(n, 1))()
                cnt = L.oracle_subsample_pick(n, np.float32(o[1]), idx)
                for q in range(cnt):
                    buf = (C.c_float * 19)()
                    mat = C.c_int32()
                    L.oracle_scene_get_prim(tmp.h, idx[q], buf, C.byref(mat))
                    b = np.array(buf, dtype=np.float32)
                    if b[0] == 1:
                        sink.append(("s", b[1:4], b[4], mat.value))
                    else:
                        sink.append(("t", b[1:4], b[4:7], b[7:10], b[10:13], b[13:16], b[16:19], mat.value))
        items = []
        for o in d.objects:
            add(o, items)
        for it in items:
            s._emit(it)
        return s

    def _emit(self, it):
        L, h = self.L, self.h
        if it[0] == "s":
            L.oracle_scene_add_sphere(h, f3(it[1]), np.float32(it[2]), it[3])
        elif it[0] == "t":
            L.oracle_scene_add_triangle(h, *[f3(v) for v in it[1:7]], it[7])
        else:
            _, pts, nrm, tris, smooth, mat = it
            pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
            nrm = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
            tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
            sm = None if smooth is None else np.ascontiguousarray(smooth, dtype=np.uint8)
            L.oracle_scene_add_mesh(h, pts.shape[0], fp(pts), None if nrm is None else fp(nrm), tris.shape[0],
                                    tris.ctypes.data_as(_I), None if sm is None else sm.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    0 if sm is None else sm.shape[0], mat)

    def build(self):
        self.L.oracle_scene_build(self.h)
        return self

    def size(self):
        w, h = C.c_int32(), C.c_int32()
        self.L.oracle_scene_size(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def render(self, spp=1, seed=0, threads=8):
        w, h = self.size()
        out = np.zeros((h, w, 3), dtype=np.float32)
        st = OStats()
        self.L.oracle_render(self.h, spp, seed, threads, fp(out), C.byref(st))
        return out, st.as_dict()

    def render_rows(self, row0, row1, spp=1, seed=0, threads=8):
        w, h = self.size()
        out = np.zeros((row1 - row0, w, 3), dtype=np.float32)
        st = OStats()
        self.L.oracle_render_rows(self.h, row0, row1, spp, seed, threads, fp(out), C.byref(st))
        return out, st.as_dict()

    def n_prims(self):
        return self.L.oracle_scene_n_prims(self.h)

    def prims(self):
        n = self.n_prims()
        out = np.zeros((n, 19), dtype=np.float32)
        mats = np.zeros(n, dtype=np.int32)
        for i in range(n):
            m = C.c_int32()
            self.L.oracle_scene_get_prim(self.h, i, fp(out[i]), C.byref(m))
            mats[i] = m.value
        return out, mats

    def tree(self):
        n = self.L.oracle_tree_dump(self.h, None, None, None, None, None, None)
        kind = np.zeros(n, dtype=np.int32)
        boxes = np.zeros((n, 6), dtype=np.float32)
        first = np.zeros(n, dtype=np.int32)
        count = np.zeros(n, dtype=np.int32)
        prim = np.zeros(max(self.n_prims(), 1), dtype=np.int32)
        npo = C.c_int32()
        self.L.oracle_tree_dump(self.h, kind.ctypes.data_as(_I), fp(boxes), first.ctypes.data_as(_I),
                                count.ctypes.data_as(_I), prim.ctypes.data_as(_I), C.byref(npo))
        return dict(kind=kind, boxes=boxes, first=first, count=count, prim=prim[:npo.value])

    def camera_ray(self, u, v):
        x0, d = (C.c_float * 3)(), (C.c_float * 3)()
        t = C.c_float()
        self.L.oracle_camera_ray(self.h, np.float32(u), np.float32(v), x0, d, C.byref(t))
        return np.array(list(x0) + list(d) + [t.value], dtype=np.float32)

    def camera_matrices(self):
        tf = np.zeros(16, dtype=np.float32)
        vtf = np.zeros(16, dtype=np.float32)
        self.L.oracle_camera_matrices(self.h, fp(tf), fp(vtf))
        return tf.reshape(4, 4), vtf.reshape(4, 4)

    def intersect(self, ray):
        ray = np.asarray(ray, dtype=np.float32)
        t = C.c_float()
        w = self.L.oracle_scene_intersect(self.h, f3(ray[0:3]), f3(ray[3:6]), np.float32(ray[6]), C.byref(t))
        return w, (t.value if w >= 0 else 0.0)

    def radiance(self, ray):
        ray = np.asarray(ray, dtype=np.float32)
        out = (C.c_float * 3)()
        self.L.oracle_radiance(self.h, f3(ray[0:3]), f3(ray[3:6]), np.float32(ray[6]), out)
        return np.array(list(out), dtype=np.float32)
