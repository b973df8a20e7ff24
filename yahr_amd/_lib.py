"""ctypes binding of libyahr_hip.so (include/yahr_hip.h).  No fallback: if the library is missing the
import of yahr_amd fails loudly; if no gfx950 device is present every device entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyahr_hip.so")

YAHR_OK = 0
ERR_NAMES = {-1: "YAHR_E_ARG", -2: "YAHR_E_PARSE", -3: "YAHR_E_IO", -4: "YAHR_E_HIP", -5: "YAHR_E_NOMEM",
             -6: "YAHR_E_NODEVICE", -7: "YAHR_E_MATERIAL"}

SPLIT_MIDPOINT, SPLIT_SAH = 0, 1
TRAVERSE_AUTO, TRAVERSE_LANE, TRAVERSE_PACKET, TRAVERSE_HYBRID, TRAVERSE_STREAM = 0, 1, 2, 3, 4
TRAVERSE_STACK, TRAVERSE_HYBRID_STACK, TRAVERSE_ORDERED = 5, 6, 7
BUILDER_AUTO, BUILDER_HOST, BUILDER_DEVICE = 0, 1, 2
TILES_WAVE, TILES_REFERENCE = 0, 1
GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_NONE = 0, 1, 2, 3
TUNE_XCD_BANDS, TUNE_CENTRE_FIRST, TUNE_REGROUP, TUNE_ONE_PIXEL_PER_LANE = 1, 2, 4, 8      # yahr_render_opts.tuning bits (diagnostics)


class YahrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Node(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("a", C.c_int32), ("hi", C.c_float * 3), ("b", C.c_int32)]


class Prim(C.Structure):
    _fields_ = [("p0", C.c_float * 3), ("mat", C.c_int32), ("e1", C.c_float * 3), ("kind", C.c_int32),
                ("e2", C.c_float * 3), ("orig", C.c_int32)]


class PrimNormals(C.Structure):
    _fields_ = [("n0", C.c_float * 3), ("pad0", C.c_float), ("n1", C.c_float * 3), ("pad1", C.c_float),
                ("n2", C.c_float * 3), ("pad2", C.c_float)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("shininess", C.c_float), ("specular", C.c_float * 3), ("pad", C.c_float)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("pad0", C.c_float), ("spectrum", C.c_float * 3), ("pad1", C.c_float)]


class FlatScene(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32),
                ("cam_tf", C.c_float * 16), ("cam_vtf", C.c_float * 16), ("focal", C.c_float),
                ("n_nodes", C.c_int32), ("nodes", C.POINTER(Node)),
                ("n_prims", C.c_int32), ("prims", C.POINTER(Prim)), ("normals", C.POINTER(PrimNormals)),
                ("n_mats", C.c_int32), ("mats", C.POINTER(Material)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light))]


class RenderOpts(C.Structure):
    _fields_ = [("spp", C.c_int32), ("seed", C.c_uint32), ("traverse", C.c_int32), ("tiles", C.c_int32),
                ("n_batches", C.c_int32), ("collect_stats", C.c_int32),
                ("n_shards", C.c_int32), ("shard", C.c_int32), ("strip_rows", C.c_int32),
                ("tuning", C.c_int32), ("frames", C.c_int32), ("frame_rows", C.c_int32), ("stream_walker", C.c_int32), ("reserved", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("reflection_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("wave_node_steps", C.c_uint64), ("wave_prim_steps", C.c_uint64),
                ("bad_material", C.c_uint64), ("kernel_ms", C.c_double), ("upload_ms", C.c_double),
                ("max_wave_cycles", C.c_uint64), ("sum_wave_cycles", C.c_uint64),
                ("wave_stage1_steps", C.c_uint64), ("wave_stage2_steps", C.c_uint64),
                ("tri_tests_culled", C.c_uint64), ("wave_cull_steps", C.c_uint64), ("tri_cull_violations", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/yahr_hip.h declares: name -> (restype, argtypes)
_F3 = C.POINTER(C.c_float)
_VP = C.c_void_p
SIGNATURES = {
    "yahr_scene_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_VP)]),
    "yahr_scene_parse": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(_VP)]),
    "yahr_parse_float32": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_float)]),
    "yahr_scene_new": (C.c_int, [C.POINTER(_VP)]),
    "yahr_scene_set_camera": (C.c_int, [_VP, C.c_float, C.c_float, C.c_float, _F3, _F3, _F3]),
    "yahr_scene_set_integrator": (C.c_int, [_VP, C.c_int32]),
    "yahr_scene_set_culling": (C.c_int, [_VP, C.c_int32, C.c_int32]),
    "yahr_scene_set_builder": (C.c_int, [_VP, C.c_int32]),
    "yahr_scene_builder_used": (C.c_int32, [_VP, C.POINTER(C.c_double)]),
    "yahr_scene_add_material": (C.c_int, [_VP, C.c_char_p, _F3, _F3, _F3, C.c_float]),
    "yahr_scene_add_light": (C.c_int, [_VP, _F3, _F3]),
    "yahr_scene_add_sphere": (C.c_int, [_VP, _F3, C.c_float, C.c_char_p]),
    "yahr_scene_add_triangle": (C.c_int, [_VP, _F3, _F3, _F3, _F3, _F3, _F3, C.c_char_p]),
    "yahr_scene_add_mesh": (C.c_int, [_VP, C.c_int32, _F3, _F3, C.c_int32, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_uint8), C.c_int32, C.c_char_p]),
    "yahr_scene_begin_subsampled": (C.c_int, [_VP, C.c_float]),
    "yahr_scene_end_subsampled": (C.c_int, [_VP]),
    "yahr_scene_build": (C.c_int, [_VP]),
    "yahr_scene_flat": (C.POINTER(FlatScene), [_VP]),
    "yahr_scene_info": (C.c_int, [_VP, _F3, _F3, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "yahr_scene_free": (None, [_VP]),
    "yahr_batch_window": (None, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "yahr_n_batches": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "yahr_default_opts": (None, [C.POINTER(RenderOpts)]),
    "yahr_device_count": (C.c_int, []),
    "yahr_device_warmup": (C.c_int, [C.c_int]),
    "yahr_device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "yahr_debug_fail_alloc": (None, [C.c_int32]),
    "yahr_debug_pair_records": (C.c_int32, [C.POINTER(FlatScene), _VP, C.c_int32, C.POINTER(C.c_int32)]),
    "yahr_debug_box_flags": (C.c_int32, [C.POINTER(FlatScene)]),
    "yahr_debug_powf": (C.c_int, [C.c_int, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "yahr_debug_powf_host": (C.c_int, [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "yahr_debug_parallel_lists": (C.c_int32, []),
    "yahr_ctx_create": (C.c_int, [C.POINTER(FlatScene), C.c_int, C.POINTER(_VP)]),
    "yahr_ctx_destroy": (None, [_VP]),
    "yahr_shard_rows": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "yahr_shard_global_row": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "yahr_ctx_render": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(RenderOpts), _VP, _VP]),
    "yahr_ctx_set_cameras": (C.c_int, [_VP, C.c_int32, _F3]),
    "yahr_ctx_stats": (C.c_int, [_VP, C.POINTER(Stats)]),
    "yahr_ctx_trace": (C.c_int, [_VP, C.c_int32, _F3, C.c_int32, C.POINTER(C.c_int32), _F3]),
    "yahr_ctx_occluded": (C.c_int, [_VP, C.c_int32, _F3, C.c_int32, C.POINTER(C.c_int32)]),
    "yahr_ctx_radiance": (C.c_int, [_VP, C.c_int32, _F3, C.c_int32, _F3]),
    "yahr_ctx_camera_rays": (C.c_int, [_VP, C.c_int32, _F3, _F3]),
    "yahr_render_rows": (C.c_int, [C.POINTER(FlatScene), C.c_int32, C.c_int32, C.c_int, C.POINTER(RenderOpts), _F3, C.POINTER(Stats)]),
    "yahr_render_multi": (C.c_int, [C.POINTER(FlatScene), C.c_int, C.POINTER(RenderOpts), _F3, C.POINTER(Stats)]),
    "yahr_multi_create": (C.c_int, [C.POINTER(FlatScene), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int32, C.POINTER(_VP)]),
    "yahr_multi_render": (C.c_int, [_VP, C.POINTER(RenderOpts), _F3, C.POINTER(Stats)]),
    "yahr_multi_set_cameras": (C.c_int, [_VP, C.c_int32, _F3]),
    "yahr_multi_submit": (C.c_int, [_VP, C.POINTER(RenderOpts), _F3]),
    "yahr_multi_wait": (C.c_int, [_VP, C.POINTER(Stats)]),
    "yahr_pinned_alloc": (_VP, [C.c_size_t]),
    "yahr_pinned_free": (None, [_VP]),
    "yahr_multi_transport": (C.c_int, [_VP]),
    "yahr_multi_last_gather_ms": (C.c_double, [_VP]),
    "yahr_multi_destroy": (None, [_VP]),
    "yahr_write_png": (C.c_int, [C.c_char_p, _F3, C.c_int32, C.c_int32]),
    "yahr_write_ppm": (C.c_int, [C.c_char_p, _F3, C.c_int32, C.c_int32]),
    "yahr_write_pfm": (C.c_int, [C.c_char_p, _F3, C.c_int32, C.c_int32]),
    "yahr_last_error": (C.c_char_p, []),
    "yahr_version": (C.c_char_p, []),
}


def load():
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch first
    # makes libyahr_hip.so's NEEDED libamdhip64.so.7 resolve to that already-loaded copy, so torch tensors,
    # torch streams and RCCL share the runtime our kernels are launched on.  Without torch installed the
    # library binds the system ROCm runtime (/opt/rocm), as the `yahr` command line does.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C yahr_amd/csrc`.  yahr_amd has no pure-Python or CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch; fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(rc):
    if rc != YAHR_OK:
        raise YahrError(rc, lib.yahr_last_error().decode("utf-8", "replace"))
