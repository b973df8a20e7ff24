"""yahr_amd -- MI355X-native render hot path of pstiasny/yahr behind a C ABI (libyahr_hip.so).

Python here is only the host-side mirror of the reference's interface for that path:

    Scene.load(path) / Scene.parse(text)      readFile + `read :: Scene`          (main.hs:115-117)
    Scene.build() / .flat                     buildCollisionModel + cull          (main.hs:41-60,118)
    square_batches(w, h, n), n_batches(..)    Sampling.squareBatches / nBatches   (Sampling.hs:5-21, main.hs:128)
    Context(scene, device).render(...)        render / renderEval / renderPar     (main.hs:68-96)
    render_image(scene, ...)                  render + samplesToImage             (main.hs:98-107)
    write_png(path, rgb)                      savePngImage (ImageRGBF img)        (main.hs:142)

All arithmetic of the path runs in hand-written HIP kernels (yahr_amd/csrc/render_kernel.hip); there is no
CPU implementation in this package.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BUILDER_AUTO, BUILDER_DEVICE, BUILDER_HOST, GATHER_AUTO, GATHER_NONE, GATHER_PEER, GATHER_RCCL, SPLIT_MIDPOINT, SPLIT_SAH, TILES_REFERENCE, TILES_WAVE, TRAVERSE_AUTO, TRAVERSE_HYBRID, TRAVERSE_STREAM, TRAVERSE_STACK, TRAVERSE_HYBRID_STACK, TRAVERSE_ORDERED,
                   TRAVERSE_LANE, TRAVERSE_PACKET, TUNE_CENTRE_FIRST, TUNE_ONE_PIXEL_PER_LANE, TUNE_REGROUP, TUNE_XCD_BANDS, RenderOpts, Stats,
                   YahrError, check, lib)

__all__ = ["Scene", "Context", "MultiContext", "GATHER_AUTO", "GATHER_RCCL", "GATHER_PEER", "GATHER_NONE", "device_mem_info", "render_image", "square_batches", "batch_window", "n_batches", "shard_rows",
           "shard_global_rows", "write_png", "write_ppm", "write_pfm", "device_count", "YahrError",
           "SPLIT_MIDPOINT", "SPLIT_SAH", "TRAVERSE_AUTO", "TRAVERSE_LANE", "TRAVERSE_PACKET", "TRAVERSE_HYBRID", "TRAVERSE_STREAM", "TRAVERSE_STACK", "TRAVERSE_HYBRID_STACK", "TRAVERSE_ORDERED",
           "TILES_WAVE", "TILES_REFERENCE", "TUNE_XCD_BANDS", "TUNE_CENTRE_FIRST", "TUNE_REGROUP", "TUNE_ONE_PIXEL_PER_LANE"]


def _f3(v):
    a = (C.c_float * 3)(*[float(np.float32(x)) for x in v])
    return a


def _fptr(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_float))


def device_count():
    return int(lib.yahr_device_count())


def device_mem_info(device=0):
    """(free, total) bytes of a device's HBM"""
    f, t = C.c_uint64(), C.c_uint64()
    check(lib.yahr_device_mem_info(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def batch_window(w, h, num, count):
    """Sampling.batchWindow (Sampling.hs:9-21) -> (x0, y0, x1, y1)."""
    out = (C.c_int32 * 4)()
    lib.yahr_batch_window(w, h, num, count, out)
    return tuple(out)


def square_batches(w, h, count):
    """Sampling.squareBatches (Sampling.hs:5-7): list of batches, each a list of (u, v), u-major."""
    res = []
    for i in range(count):
        x0, y0, x1, y1 = batch_window(w, h, i, count)
        res.append([(u, v) for u in range(x0, x1) for v in range(y0, y1)])
    return res


def n_batches(num_threads, width, height):
    """nBatches of main.hs:128-130."""
    return int(lib.yahr_n_batches(num_threads, width, height))


def shard_rows(height, n_shards, shard, strip_rows=8):
    return int(lib.yahr_shard_rows(height, n_shards, shard, strip_rows))


def shard_global_rows(height, n_shards, shard, strip_rows=8):
    """global image row of every local row of a shard (numpy int64 array)"""
    n = shard_rows(height, n_shards, shard, strip_rows)
    return np.array([lib.yahr_shard_global_row(i, n_shards, shard, strip_rows) for i in range(n)], dtype=np.int64)


class Scene:
    """Host scene: Scene.hs data + expand + BVH build (C++ in libyahr_hip.so)."""

    def __init__(self, handle=None):
        if handle is None:
            h = C.c_void_p()
            check(lib.yahr_scene_new(C.byref(h)))
            handle = h
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:           # lib is None during interpreter shutdown
            lib.yahr_scene_free(h)

    @classmethod
    def load(cls, path, lenient=False):
        h = C.c_void_p()
        check(lib.yahr_scene_load(str(path).encode(), int(lenient), C.byref(h)))
        return cls(h)

    @classmethod
    def parse(cls, text, lenient=False):
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        check(lib.yahr_scene_parse(data, len(data), int(lenient), C.byref(h)))
        return cls(h)

    # ---- programmatic assembly (what a foreign host marshals instead of text)
    def set_camera(self, imW, imH, focal, look, up, pos):
        check(lib.yahr_scene_set_camera(self._h, imW, imH, focal, _f3(look), _f3(up), _f3(pos)))

    def set_integrator(self, depth):
        check(lib.yahr_scene_set_integrator(self._h, depth))

    def set_culling(self, max_depth, split=SPLIT_MIDPOINT):
        check(lib.yahr_scene_set_culling(self._h, max_depth, split))

    def set_builder(self, builder):
        """who builds the BVH: _lib.BUILDER_AUTO / BUILDER_HOST / BUILDER_DEVICE (identical trees)"""
        check(lib.yahr_scene_set_builder(self._h, builder))
        return self

    def builder_used(self):
        """(BUILDER_HOST or BUILDER_DEVICE, device milliseconds) of the last build"""
        ms = C.c_double()
        return int(lib.yahr_scene_builder_used(self._h, C.byref(ms))), float(ms.value)

    def add_material(self, id, ambient, diffuse, specular, shininess):
        check(lib.yahr_scene_add_material(self._h, id.encode(), _f3(ambient), _f3(diffuse), _f3(specular), shininess))

    def add_light(self, pos, spectrum):
        check(lib.yahr_scene_add_light(self._h, _f3(pos), _f3(spectrum)))

    def add_sphere(self, centre, radius, material_id):
        check(lib.yahr_scene_add_sphere(self._h, _f3(centre), radius, material_id.encode()))

    def add_triangle(self, p0, p1, p2, n0, n1, n2, material_id):
        check(lib.yahr_scene_add_triangle(self._h, _f3(p0), _f3(p1), _f3(p2), _f3(n0), _f3(n1), _f3(n2), material_id.encode()))

    def add_mesh(self, points, normals, tris, smooth, material_id):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        sm = None if smooth is None else np.ascontiguousarray(smooth, dtype=np.uint8)
        check(lib.yahr_scene_add_mesh(
            self._h, pts.shape[0], _fptr(pts), None if nrm is None else _fptr(nrm), tri.shape[0],
            tri.ctypes.data_as(C.POINTER(C.c_int32)),
            None if sm is None else sm.ctypes.data_as(C.POINTER(C.c_uint8)), 0 if sm is None else sm.shape[0],
            material_id.encode()))

    def begin_subsampled(self, size):
        check(lib.yahr_scene_begin_subsampled(self._h, size))

    def end_subsampled(self):
        check(lib.yahr_scene_end_subsampled(self._h))

    # ---- build + flattened view
    def build(self):
        check(lib.yahr_scene_build(self._h))
        return self

    @property
    def flat(self):
        p = lib.yahr_scene_flat(self._h)
        if not p:
            raise YahrError(-1, lib.yahr_last_error().decode())
        return p

    @property
    def width(self):
        return int(self.flat.contents.width)

    @property
    def height(self):
        return int(self.flat.contents.height)

    @property
    def depth(self):
        return int(self.flat.contents.depth)

    @property
    def n_lights(self):
        return int(self.flat.contents.n_lights)

    def info(self):
        imW, imH = C.c_float(), C.c_float()
        d, s, n = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.yahr_scene_info(self._h, C.byref(imW), C.byref(imH), C.byref(d), C.byref(s), C.byref(n)))
        return dict(imW=imW.value, imH=imH.value, bvh_depth=d.value, split=s.value, n_objects=n.value)

    def arrays(self):
        """numpy copies of the flattened arrays (for tests and diagnostics)"""
        f = self.flat.contents

        def grab(ptr, n, words, dt=np.float32):
            if n == 0:
                return np.zeros((0, words), dtype=dt)
            buf = C.cast(ptr, C.POINTER(C.c_uint32 * (n * words))).contents
            return np.frombuffer(buf, dtype=np.uint32).copy().reshape(n, words)

        nodes = grab(f.nodes, f.n_nodes, 8)
        prims = grab(f.prims, f.n_prims, 12)
        normals = grab(f.normals, f.n_prims, 12)
        mats = grab(f.mats, f.n_mats, 8)
        lights = grab(f.lights, f.n_lights, 8)
        fl = lambda a: a.view(np.float32)
        il = lambda a: a.view(np.int32)
        return dict(
            node_lo=fl(nodes[:, 0:3]), node_a=il(nodes[:, 3]), node_hi=fl(nodes[:, 4:7]), node_b=il(nodes[:, 7]),
            prim_p0=fl(prims[:, 0:3]), prim_mat=il(prims[:, 3]), prim_e1=fl(prims[:, 4:7]), prim_kind=il(prims[:, 7]),
            prim_e2=fl(prims[:, 8:11]), prim_orig=il(prims[:, 11]),
            n0=fl(normals[:, 0:3]), n1=fl(normals[:, 4:7]), n2=fl(normals[:, 8:11]),
            mat_diffuse=fl(mats[:, 0:3]), mat_shininess=fl(mats[:, 3]), mat_specular=fl(mats[:, 4:7]),
            light_pos=fl(lights[:, 0:3]), light_spectrum=fl(lights[:, 4:7]),
            cam_tf=np.array(list(f.cam_tf), dtype=np.float32).reshape(4, 4),
            cam_vtf=np.array(list(f.cam_vtf), dtype=np.float32).reshape(4, 4), focal=np.float32(f.focal))


def make_opts(spp=1, seed=0, traverse=TRAVERSE_AUTO, tiles=TILES_WAVE, n_batches=0, collect_stats=False,
              n_shards=1, shard=0, strip_rows=8, tuning=0, frames=1, frame_rows=0, stream_walker=0):
    o = RenderOpts()
    lib.yahr_default_opts(C.byref(o))
    o.spp, o.seed, o.traverse, o.tiles, o.n_batches = spp, seed, traverse, tiles, n_batches
    o.collect_stats = int(collect_stats)
    o.n_shards, o.shard, o.strip_rows = n_shards, shard, strip_rows
    o.tuning = tuning
    o.frames = frames
    o.frame_rows = frame_rows
    o.stream_walker = stream_walker
    return o


class Context:
    """A scene resident in one GPU's HBM (yahr_ctx)."""

    def __init__(self, scene, device=0):
        self.scene = scene
        self.device = device
        h = C.c_void_p()
        check(lib.yahr_ctx_create(scene.flat, device, C.byref(h)))
        self._h = h
        self.width, self.height = scene.width, scene.height

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.yahr_ctx_destroy(h)

    def __del__(self):
        self.close()

    def render_into(self, d_ptr, row0=0, row1=None, opts=None, stream=0):
        """Asynchronous render into DEVICE memory at integer address d_ptr on HIP stream `stream`."""
        if row1 is None:
            row1 = self.height
        o = opts if opts is not None else make_opts()
        check(lib.yahr_ctx_render(self._h, row0, row1, C.byref(o), C.c_void_p(d_ptr), C.c_void_p(stream)))

    def set_cameras(self, cameras):
        """cameras for frame batches: rows of (imW, imH, focal, look[3], up[3], pos[3]); None/[] clears them"""
        cams = np.zeros((0, 12), dtype=np.float32) if cameras is None else \
            np.ascontiguousarray(cameras, dtype=np.float32).reshape(-1, 12)
        check(lib.yahr_ctx_set_cameras(self._h, cams.shape[0], _fptr(cams) if cams.shape[0] else None))

    def stats(self):
        st = Stats()
        check(lib.yahr_ctx_stats(self._h, C.byref(st)))
        return st.as_dict()

    def render(self, row0=0, row1=None, **kw):
        """Render rows (or a shard) into a new torch CUDA tensor [rows, width, 3] float32."""
        import torch
        o = make_opts(**kw)
        if row1 is None:
            row1 = self.height
        rows = shard_rows(self.height, o.n_shards, o.shard, o.strip_rows) if o.n_shards > 1 else row1 - row0
        dev = torch.device("cuda", self.device)
        shape = (rows, self.width, 3) if o.frames <= 1 else (o.frames, rows, self.width, 3)
        out = torch.zeros(shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            self.render_into(out.data_ptr(), row0, row1, o, torch.cuda.current_stream().cuda_stream)
        return out

    # diagnostic ray queries (KAT checks on the device)
    def trace(self, rays, traverse=TRAVERSE_PACKET):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        check(lib.yahr_ctx_trace(self._h, n, _fptr(rays), traverse, prim.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(t)))
        return prim, t

    def occluded(self, rays, traverse=TRAVERSE_PACKET):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        blocked = np.zeros(n, dtype=np.int32)
        check(lib.yahr_ctx_occluded(self._h, n, _fptr(rays), traverse, blocked.ctypes.data_as(C.POINTER(C.c_int32))))
        return blocked

    def radiance(self, rays, traverse=TRAVERSE_AUTO):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        rgb = np.zeros((n, 3), dtype=np.float32)
        check(lib.yahr_ctx_radiance(self._h, n, _fptr(rays), traverse, _fptr(rgb)))
        return rgb

    def camera_rays(self, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        rays = np.zeros((uv.shape[0], 7), dtype=np.float32)
        check(lib.yahr_ctx_camera_rays(self._h, uv.shape[0], _fptr(uv), _fptr(rays)))
        return rays


class MultiContext:
    """All GPUs of the node from one process (yahr_multi): resident scenes, one RCCL / peer-DMA gather per frame."""

    def __init__(self, scene, n_devices=1, devices=None, gather=GATHER_AUTO, strip_rows=8):
        self.scene = scene
        h = C.c_void_p()
        devs = None if devices is None else (C.c_int * len(devices))(*devices)
        self._pinned, self._inflight = [], []
        self._pool, self._owner = {}, {}          # page-locked buffers: free ones per shape; address -> (pointer, shape) of those handed out
        check(lib.yahr_multi_create(scene.flat, n_devices, devs, gather, strip_rows, C.byref(h)))
        self._h = h

    @property
    def transport(self):
        return int(lib.yahr_multi_transport(self._h))

    @property
    def last_gather_ms(self):
        return float(lib.yahr_multi_last_gather_ms(self._h))

    def _out(self, frames, pinned):
        shape = (self.scene.height, self.scene.width, 3) if frames <= 1 else (frames, self.scene.height, self.scene.width, 3)
        if not pinned:
            return np.zeros(shape, dtype=np.float32)
        # page-locked buffers are RECYCLED: a pool per shape, fed by release(); a submit / wait loop that releases what it
        # has consumed holds two buffers (the two submissions in flight), whatever its length
        n = int(np.prod(shape))
        pool = self._pool.setdefault(shape, [])
        if pool:
            p = pool.pop()
        else:
            p = lib.yahr_pinned_alloc(4 * n)
            if not p:
                raise MemoryError("yahr_pinned_alloc")
            self._pinned.append(p)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).reshape(shape)
        self._owner[arr.ctypes.data] = (p, shape)
        return arr

    def release(self, arr):
        """hand a page-locked array returned by submit() / wait() back for reuse (no-op for pageable arrays).  The
        array must not be touched afterwards: the next submit() of that shape overwrites it."""
        key = arr.ctypes.data if isinstance(arr, np.ndarray) else None
        ent = self._owner.pop(key, None)
        if ent is not None:
            self._pool.setdefault(ent[1], []).append(ent[0])

    def render(self, **kw):
        """one frame (or a batch: frames=F) -> (numpy [h, w, 3] or [F, h, w, 3], stats dict)"""
        o = make_opts(**kw)
        out = self._out(o.frames, False)
        st = Stats()
        check(lib.yahr_multi_render(self._h, C.byref(o), _fptr(out), C.byref(st)))
        return out, st.as_dict()

    def set_cameras(self, cameras):
        cams = np.zeros((0, 12), dtype=np.float32) if cameras is None else \
            np.ascontiguousarray(cameras, dtype=np.float32).reshape(-1, 12)
        check(lib.yahr_multi_set_cameras(self._h, cams.shape[0], _fptr(cams) if cams.shape[0] else None))

    def submit(self, pinned=True, **kw):
        """asynchronous: queue a frame batch; returns the (page-locked) host array the matching wait() fills.  Page-locked
        arrays come from a pool: release() the ones you are done with and a submit / wait loop allocates two in all."""
        o = make_opts(**kw)
        out = self._out(o.frames, pinned)
        try:
            check(lib.yahr_multi_submit(self._h, C.byref(o), _fptr(out)))
        except Exception:
            self.release(out)                  # a failed submission keeps no buffer
            raise
        self._inflight.append(out)
        return out

    def wait(self):
        """complete the oldest submission -> (its host array, its stats).  A page-locked array is a VIEW of memory this
        object owns: copy it (or finish with it) before close(), and give it back with release() in a long loop."""
        st = Stats()
        check(lib.yahr_multi_wait(self._h, C.byref(st)))
        return self._inflight.pop(0), st.as_dict()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.yahr_multi_destroy(h)
            for p in getattr(self, "_pinned", []):
                lib.yahr_pinned_free(p)
            self._pinned, self._pool, self._owner = [], {}, {}

    def __del__(self):
        self.close()


def render_image(scene, device=0, n_devices=1, **kw):
    """One-shot render of the whole image to a numpy array [h, w, 3] (yahr_render_rows / yahr_render_multi)."""
    o = make_opts(**kw)
    f = scene.flat
    h, w = scene.height, scene.width
    out = np.zeros((h, w, 3), dtype=np.float32)
    st = Stats()
    if n_devices > 1:
        check(lib.yahr_render_multi(f, n_devices, C.byref(o), _fptr(out), C.byref(st)))
    else:
        check(lib.yahr_render_rows(f, 0, h, device, C.byref(o), _fptr(out), C.byref(st)))
    return out, st.as_dict()


def _write(fn, path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    check(fn(str(path).encode(), _fptr(rgb), w, h))


def write_png(path, rgb):
    _write(lib.yahr_write_png, path, rgb)


def write_ppm(path, rgb):
    _write(lib.yahr_write_ppm, path, rgb)


def write_pfm(path, rgb):
    _write(lib.yahr_write_pfm, path, rgb)
