/*
 * yahr_hip.h -- C ABI of libyahr_hip.so: the MI355X (gfx950) render hot path of pstiasny/yahr.
 *
 * The reference (Haskell) has no FFI; the seam this library replaces is internal to main.hs:
 *
 *     render / renderEval / renderPar :: Int -> Int -> [[SampleCoordinates]]
 *                                     -> (Float -> Float -> Ray) -> (Ray -> Spectrum) -> [Sample]
 *                                                                   (main.hs:68-96, selected at :133-137)
 *
 * with the two closures (caster = Cameras.computeInitialRay camera, main.hs:121; li = Integrators.radiance
 * integ lights collider, main.hs:125-126) replaced by plain data: `yahr_scene`.  Everything is extern "C",
 * plain pointers and sizes, no exceptions, no callbacks, no torch types.  INTEGRATION.md shows the
 * `foreign import ccall` stub a yahr maintainer would add.
 *
 * Layers (each usable on its own):
 *   1. yahr_scene            flattened scene: camera matrices, BVH nodes in reference DFS order,
 *                            leaf-ordered primitive records, materials, lights.   <- what the kernel eats
 *   2. yahr_host_scene       C++ host mirror of Scene.hs / main.hs:41-60 / Culling.hs: .yahrr loader
 *                            (derived-Read grammar), expand, BVH build -> produces (1)
 *   3. yahr_ctx              a scene resident in one GPU's HBM + render launches on a HIP stream
 *   4. one-shot conveniences yahr_render_rows / yahr_render_multi (host buffers in, host image out)
 *   4b. yahr_multi           all GPUs of the node from one process: resident scenes, one RCCL / peer-DMA gather per frame
 *
 * Conventions: return 0 (YAHR_OK) on success, negative code otherwise; yahr_last_error() gives the message
 * (thread-local).  Inputs are caller-owned and read-only for the duration of the call.  A yahr_ctx is
 * thread-compatible: one call at a time per context.
 */
#ifndef YAHR_HIP_H
#define YAHR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes size or meaning; yahr_scene.abi_version carries it and every entry point
 * that takes a yahr_scene refuses another value, so a host compiled against an older header fails loudly instead of having
 * a larger yahr_stats written over its smaller one.
 *   1 -> 2: yahr_stats grew by wave_stage1_steps / wave_stage2_steps (16 bytes); yahr_render_opts.reserved[0] became
 *           stream_walker (same size); YAHR_TRAVERSE_STACK / _HYBRID_STACK / _ORDERED; yahr_multi_submit / _wait /
 *           _set_cameras, yahr_pinned_alloc / _free.
 *   2 -> 3: yahr_stats grew by tri_tests_culled / wave_cull_steps / tri_cull_violations (24 bytes). */
#define YAHR_ABI_VERSION 3

enum {
  YAHR_OK = 0,
  YAHR_E_ARG = -1,       /* bad argument / malformed yahr_scene */
  YAHR_E_PARSE = -2,     /* .yahrr text rejected ("Prelude.read: no parse", main.hs:117) */
  YAHR_E_IO = -3,        /* file could not be read / written */
  YAHR_E_HIP = -4,       /* HIP runtime error */
  YAHR_E_NOMEM = -5,
  YAHR_E_NODEVICE = -6,  /* no usable gfx950 device / kernel image missing */
  YAHR_E_MATERIAL = -7   /* a ray shaded an object whose materialId is not in the map (Map.!, main.hs:55) */
};

/* ---------------------------------------------------------------------------------------------
 * 1. Flattened scene
 * ------------------------------------------------------------------------------------------- */

/* BVH node, 32 B.  Nodes are stored in the reference's DFS (pre)order: Culling.hs:38 visits the left
 * child, then the right one, so the left child of interior node i is node i+1.
 *   interior : b == -1, a = index of the first node after this node's subtree ("skip" link)
 *   leaf     : b >= 0 = index of its first primitive record, a = number of primitives
 *              (a == 1: `wrapCollider col bb`, Culling.hs:33;  a > 1: multiLeaf, Culling.hs:52-54;
 *               a == 0: the null collider of Culling.hs:32, box = AABBs.empty)
 * lo/hi = BoundingBox min/max (AABBs.hs:8).
 * Any boxes are rendered with the reference's semantics.  Two properties every tree of Culling.hs:27-54 has are CHECKED
 * when a context is created, and the packet walker's shortcuts are taken only where they hold: all boxes finite with
 * lo <= hi (NaN-free slab test), and every child's box inside its parent's (Culling.hs:39: a parent's box is the join of
 * its children's -- then a ray that fails a node fails everything below it, and the walk keeps no per-ray memory of the
 * subtrees a ray has left). */
typedef struct yahr_node {
  float lo[3]; int32_t a;
  float hi[3]; int32_t b;
} yahr_node;

/* Primitive record in LEAF order, 48 B (so a leaf's primitives are contiguous).
 *   kind 0 triangle (Shapes.hs:36-59): p0, e1 = p1 - p0, e2 = p2 - p0 (computed in binary32)
 *   kind 1 sphere   (Shapes.hs:13-29): p0 = centre, e1[0] = radius
 * mat  = index into mats, or -1 for an unknown materialId.  orig = index in `objects >>= expand` order
 * (main.hs:44). */
typedef struct yahr_prim {
  float p0[3]; int32_t mat;
  float e1[3]; int32_t kind;
  float e2[3]; int32_t orig;
} yahr_prim;

/* Triangle vertex normals n0,n1,n2 (Scene.hs:20-28) for primitive record i; unused for spheres. 48 B. */
typedef struct yahr_prim_normals {
  float n0[3]; float pad0;
  float n1[3]; float pad1;
  float n2[3]; float pad2;
} yahr_prim_normals;

/* blinnPhong ambient diffuse specular shininess (Shaders.hs:12-14); ambient is ignored there. 32 B. */
typedef struct yahr_material {
  float diffuse[3]; float shininess;
  float specular[3]; float pad;
} yahr_material;

/* PointLight position spectrum (Lights.hs:7). 32 B. */
typedef struct yahr_light {
  float pos[3]; float pad0;
  float spectrum[3]; float pad1;
} yahr_light;

typedef struct yahr_scene {
  int32_t abi_version;          /* YAHR_ABI_VERSION */
  int32_t width, height;        /* floor imW, floor imH (main.hs:122-123) */
  int32_t depth;                /* WhittedIntegrator recursionDepth (Integrators.hs:18-20) */
  float cam_tf[16];             /* cameraToWorld, row-major (Cameras.hs:58-65) */
  float cam_vtf[16];            /* tf !*! rasterToCamera (Cameras.hs:81) */
  float focal;                  /* focalLength */
  int32_t n_nodes;  const yahr_node *nodes;
  int32_t n_prims;  const yahr_prim *prims; const yahr_prim_normals *normals;
  int32_t n_mats;   const yahr_material *mats;
  int32_t n_lights; const yahr_light *lights;
} yahr_scene;

/* ---------------------------------------------------------------------------------------------
 * 2. Host scene: loader, expand, BVH build  (Scene.hs, main.hs:41-60, Culling.hs:21-112)
 * ------------------------------------------------------------------------------------------- */

typedef struct yahr_host_scene yahr_host_scene;

enum { YAHR_SPLIT_MIDPOINT = 0, YAHR_SPLIT_SAH = 1 };   /* Culling.hs:18 */

/* readFile + `read :: Scene` (main.hs:115-117).  lenient != 0 additionally accepts the stale bundled
 * scene.yahrr dialect (PointLight with one argument, TriangleMesh without the Normals/Smooth fields). */
int yahr_scene_load(const char *path, int lenient, yahr_host_scene **out);
int yahr_scene_parse(const char *text, size_t len, int lenient, yahr_host_scene **out);
/* The loader's numeric-literal conversion on its own (tests): decimal / 0x / 0o lexeme -> binary32, correctly
 * rounded as GHC's `read :: Float` does (Scene.hs derives Read). */
int yahr_parse_float32(const char *lexeme, size_t len, float *out);
/* Texts of a megabyte and more are parsed by several threads (YAHR_PARSE_THREADS, default: the host's cores, at most
 * 16): long lists are cut at their top-level commas and the runs parsed side by side; any error falls back to the
 * sequential code, so results and error messages do not depend on the thread count.  Diagnostics (tests): how many
 * lists of the calling thread's last yahr_scene_parse / yahr_scene_load went that way. */
int32_t yahr_debug_parallel_lists(void);

/* programmatic assembly: what a foreign host marshals instead of text */
int yahr_scene_new(yahr_host_scene **out);
int yahr_scene_set_camera(yahr_host_scene *s, float imW, float imH, float focal,
                          const float look[3], const float up[3], const float pos[3]);   /* Cameras.hs:54-56 */
int yahr_scene_set_integrator(yahr_host_scene *s, int32_t recursion_depth);
int yahr_scene_set_culling(yahr_host_scene *s, int32_t max_depth, int32_t split_mode);
int yahr_scene_add_material(yahr_host_scene *s, const char *id, const float ambient[3],
                            const float diffuse[3], const float specular[3], float shininess);
int yahr_scene_add_light(yahr_host_scene *s, const float pos[3], const float spectrum[3]);
int yahr_scene_add_sphere(yahr_host_scene *s, const float centre[3], float radius, const char *material_id);
int yahr_scene_add_triangle(yahr_host_scene *s, const float p0[3], const float p1[3], const float p2[3],
                            const float n0[3], const float n1[3], const float n2[3], const char *material_id);
/* TriangleMesh (Scene.hs:29-36): normals NULL = Nothing, smooth NULL = Nothing.  tris = n_tris*3 indices. */
int yahr_scene_add_mesh(yahr_host_scene *s, int32_t n_points, const float *points, const float *normals,
                        int32_t n_tris, const int32_t *tris, const uint8_t *smooth, int32_t n_smooth,
                        const char *material_id);
/* Subsampled (Scene.hs:37-40): objects added between begin/end are its subsampledObjects (nestable). */
int yahr_scene_begin_subsampled(yahr_host_scene *s, float subsample_size);
int yahr_scene_end_subsampled(yahr_host_scene *s);

/* Who builds the BVH.  The device builder (midpoint split, finite boxes) works level by level on the GPU and yields the
 * IDENTICAL node array and leaf order as the host builder; AUTO uses it for scenes of 32768 primitives and more when a
 * device is visible and falls back to the host silently; DEVICE fails the build if it cannot be used.
 * SIDE EFFECT: with AUTO (the default) or DEVICE, yahr_scene_build / yahr_scene_flat of such a scene INITIALISE HIP and
 * launch kernels on the device that is CURRENT in the calling thread -- not necessarily the one later given to
 * yahr_ctx_create.  A host that builds scenes in a launcher process before it forks or spawns its GPU workers must keep
 * the build on the CPU: yahr_scene_set_builder(s, YAHR_BUILDER_HOST), or YAHR_BVH_BUILDER=host in the environment. */
enum { YAHR_BUILDER_AUTO = 0, YAHR_BUILDER_HOST = 1, YAHR_BUILDER_DEVICE = 2 };
int yahr_scene_set_builder(yahr_host_scene *s, int32_t builder);
/* after yahr_scene_build: YAHR_BUILDER_HOST or YAHR_BUILDER_DEVICE; *device_ms = device time of the build (0 on the host) */
int32_t yahr_scene_builder_used(const yahr_host_scene *s, double *device_ms_or_null);

/* `objects >>= expand`, buildCollisionModel, cull: fills the flattened arrays.  Idempotent. */
int yahr_scene_build(yahr_host_scene *s);
/* view of the flattened scene (valid until the host scene is modified or freed); builds if needed */
const yahr_scene *yahr_scene_flat(yahr_host_scene *s);
/* scene-file fields that are not in yahr_scene */
int yahr_scene_info(const yahr_host_scene *s, float *imW, float *imH, int32_t *bvh_depth,
                    int32_t *split_mode, int32_t *n_objects);
void yahr_scene_free(yahr_host_scene *s);

/* Sampling.hs:9-21 batchWindow -> out = {x0, y0, x1, y1};  main.hs:128-130 nBatches */
void yahr_batch_window(int32_t w, int32_t h, int32_t num, int32_t count, int32_t out[4]);
int32_t yahr_n_batches(int32_t num_threads, int32_t width, int32_t height);

/* ---------------------------------------------------------------------------------------------
 * 3. Device context and render
 * ------------------------------------------------------------------------------------------- */

typedef struct yahr_ctx yahr_ctx;

enum {
  YAHR_TRAVERSE_AUTO = 0,      /* chosen per scene (yahr_ctx_create) and launch: PACKET below 150 k primitives; from there
                                  on STREAM once a frame holds enough samples (3 M; 2 M on trees of 400 k primitives and
                                  more with leaves of more than 16), else the fused PACKET or, for big thin trees, HYBRID kernel --
                                  a policy derived from sweeps (profiles/r03_auto_sweep*.csv) for scenes
                                  whose rays do NOT stay together (triangle soups).  Whether they do is measured: for
                                  scenes of 150 k primitives and more yahr_ctx_create renders four 8-row strips with the
                                  counted packet kernel; where its lanes are 30 % busy or more (tessellated meshes:
                                  45-80 %, soups 5-20 %) AUTO never leaves the fused kernels
                                  (profiles/r03_mesh_sweep.csv) */
  YAHR_TRAVERSE_LANE = 1,      /* every lane walks the node array on its own (vector loads) */
  YAHR_TRAVERSE_PACKET = 2,    /* the wavefront walks the union of its lanes' paths (scalar node loads) */
  YAHR_TRAVERSE_HYBRID = 3,    /* adaptive: a packet while the rays stay coherent, then every lane on its own */
  YAHR_TRAVERSE_STREAM = 4,    /* wavefront form: camera rays as packets, then ONE LAUNCH PER RECURSION LEVEL with the
                                  surviving paths compacted into a queue in HBM between launches (ballot + prefix),
                                  so every wavefront of a level starts with 64 live rays, and the level launches run in
                                  Morton order of the hit points (counting sort on the device); same image, same
                                  counters.  Needs 142 + 32 x depth bytes of device memory per sample of a band of
                                  rows (allocated inside the context on first use; bands are sized to an eighth of
                                  the free memory), twice that when a launch holds several passes -- frames of a batch,
                                  bands of a frame -- which then overlap on two internal streams */
  YAHR_TRAVERSE_STACK = 5,     /* STREAM whose secondary rays run on the stack walker: every lane on its own over child-PAIR
                                  records (one fetch = both children's boxes) with an LDS stack of pending right siblings;
                                  the reference's tests in the reference's order (same image, same counters) */
  YAHR_TRAVERSE_HYBRID_STACK = 6, /* STREAM, secondary rays: a packet while coherent, then the stack walker */
  YAHR_TRAVERSE_ORDERED = 7    /* OPT-IN, not the reference's order: STREAM whose secondary rays visit the NEARER child
                                  first (stack walker); an exact tie in t still goes to the primitive that is later in
                                  the reference's order.  Same closest hit unless a hit lies in front of its own box's
                                  computed entry distance (rounding); test COUNTS differ by design */
};
enum {
  YAHR_TILES_WAVE = 0,         /* 8x8-pixel tile per wavefront */
  YAHR_TILES_REFERENCE = 1     /* the reference's squareBatches windows (Sampling.hs:5-21), one per wavefront */
};

typedef struct yahr_render_opts {
  int32_t spp;                 /* 1 = the reference's sampling (pixel corner, main.hs:73). >1: build extension */
  uint32_t seed;               /* sampler seed (spp > 1 only) */
  int32_t traverse;            /* YAHR_TRAVERSE_* */
  int32_t tiles;               /* YAHR_TILES_* */
  int32_t n_batches;           /* for YAHR_TILES_REFERENCE; 0 = yahr_n_batches(1, w, h) */
  int32_t collect_stats;       /* != 0: also count box/primitive tests (slower kernel variant) */
  /* row sharding: the image is cut into strips of strip_rows rows; this call renders the strips s with
   * s % n_shards == shard, compacted top to bottom into the output.  n_shards <= 1: rows [row0,row1). */
  int32_t n_shards, shard, strip_rows;
  int32_t tuning;              /* 0 = defaults; bit 0: XCD band swizzle of the tile order (diagnostics: slower);
                                  bit 1: dispatch centre tile rows first (diagnostics: measured slower);
                                  bit 2 (YAHR_TUNE_REGROUP): workgroups of 4 waves (2x2 tiles) that pack their
                                  active rays into the fewest waves before every traversal (diagnostics:
                                  same image, measured slower -- the walkers are latency bound);
                                  bit 3 (YAHR_TUNE_ONE_PIXEL_PER_LANE): with spp a power of two the samples of a
                                  pixel normally sit in neighbouring lanes; this bit restores the serial loop */
  /* Frame batch: frames > 1 renders that many frames in ONE launch (tiles of all frames interleaved, so the
   * tail of one frame overlaps the others).  Frame f lands at d_rgb + f * frame_rows * width * 3 and uses camera f of
   * yahr_ctx_set_cameras (the scene camera when none were set).  0 and 1 mean a single frame. */
  int32_t frames;
  int32_t frame_rows;          /* rows between consecutive frames in d_rgb; 0 = packed (the rows rendered) */
  int32_t stream_walker;       /* diagnostics: walker of the stream form's secondary rays (YAHR_TRAVERSE_LANE / _PACKET /
                                  _HYBRID / _STACK / _HYBRID_STACK / _ORDERED); 0 = the context's choice */
  int32_t reserved[3];
} yahr_render_opts;

#define YAHR_TUNE_XCD_BANDS 1
#define YAHR_TUNE_CENTRE_FIRST 2
#define YAHR_TUNE_REGROUP 4
#define YAHR_TUNE_ONE_PIXEL_PER_LANE 8   /* spp > 1: keep one pixel per lane with a serial sample loop (diagnostics) */

typedef struct yahr_stats {
  uint64_t primary_rays, reflection_rays, shadow_rays;   /* rays actually cast */
  uint64_t box_tests, tri_tests, sphere_tests;           /* per-ray counts (collect_stats only) */
  uint64_t shaded_hits;
  uint64_t wave_node_steps, wave_prim_steps;             /* wavefront-level loop trips (collect_stats only) */
  uint64_t bad_material;                                 /* shaded hits on mat == -1 */
  double kernel_ms;                                      /* device time of the render launch(es), HIP events */
  double upload_ms;                                      /* one-shot calls: host->device scene copy */
  uint64_t max_wave_cycles, sum_wave_cycles;             /* shader clocks of the longest wavefront / of all (collect_stats only) */
  uint64_t wave_stage1_steps, wave_stage2_steps;         /* packet walker: primitive steps that ran the triangle's IEEE divide
                                                            (not rejected division-free by every live lane) / its second half */
  uint64_t tri_tests_culled, wave_cull_steps;            /* packet walker, fat multiLeafs: triangle tests (part of tri_tests) whose
                                                            outcome -- every lane far -- was proven for the whole packet at once
                                                            instead of being executed / packet tests run (one per <= 64 triangles) */
  uint64_t tri_cull_violations;                          /* collect_stats builds execute every skipped test after all and count the
                                                            lanes that contradict the proof: always 0 */
} yahr_stats;

void yahr_default_opts(yahr_render_opts *o);
int yahr_device_count(void);
/* Bring the HIP runtime, the device and this library's kernel image up (~90 ms the first time in a process): a host may
 * call it from a second thread while it still parses its scene.  Idempotent; every other entry point works without it. */
int yahr_device_warmup(int device);
/* free / total bytes of a device's HBM (hipMemGetInfo after a device synchronise) */
int yahr_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);
/* Test hook: make the nth device allocation from now on (1 = the next one) fail with out-of-memory, once;
 * 0 disarms.  Used to check that no error path of this library leaks device memory. */
void yahr_debug_fail_alloc(int32_t nth);

/* Test hook (host only): the child-pair records of the stack walkers for this scene -- two yahr_node-shaped child records
 * per interior node, in DFS order of the interior nodes: a leaf child as in `nodes`; an interior child with a = rank of
 * its own pair and b = -1 - its DFS index.  Returns their number (0: the scene has none -- the root is a leaf, a box is
 * not finite, or a node's box is not exactly the join of its children's) and copies at most `cap` of them to `out`;
 * *levels = entries of the per-lane stack (depth of the deepest node). */
int32_t yahr_debug_pair_records(const yahr_scene *scene, yahr_node *out, int32_t cap, int32_t *levels);
/* Test hook (host only): what a context would let the packet walker assume about this scene's arrays -- bit 0: every node
 * box finite with lo <= hi; bit 2: ... and every child's box inside its parent's; bit 1: every triangle record's p0, e1, e2
 * within +-2^32 (see yahr_node above).  Negative: an error code. */
int32_t yahr_debug_box_flags(const yahr_scene *scene);
/* Test hook: the device's `**` on Float (BSDF.hs:25) -- glibc's powf restated (yahr_amd/csrc/device_math.hpp: glibc_powf) --
 * evaluated on `device` for n host operand pairs; the tests compare it with the host C library's powf bit for bit. */
int yahr_debug_powf(int device, int32_t n, const float *x, const float *y, float *out);
/* ... and the same source compiled for the host (no GPU needed). */
int yahr_debug_powf_host(int32_t n, const float *x, const float *y, float *out);

/* Upload a flattened scene to `device` (index as seen by this process).  The context owns device memory. */
int yahr_ctx_create(const yahr_scene *scene, int device, yahr_ctx **out);
void yahr_ctx_destroy(yahr_ctx *ctx);

/* number of image rows shard `shard` of `n_shards` owns with strips of strip_rows rows */
int32_t yahr_shard_rows(int32_t height, int32_t n_shards, int32_t shard, int32_t strip_rows);
/* global image row of local row `local` of that shard */
int32_t yahr_shard_global_row(int32_t local, int32_t n_shards, int32_t shard, int32_t strip_rows);

/* Render rows [row0,row1) (or this shard's strips when opts->n_shards > 1; row0/row1 then ignored) into
 * DEVICE memory d_rgb: float RGB, index ((local_row)*width + u)*3 -- the reference's `w*v+u` scatter
 * (main.hs:100) restricted to the rows rendered.  Asynchronous on `hip_stream` (a hipStream_t, NULL =
 * default stream); no host synchronisation, no allocation: usable inside hipGraph capture. */
int yahr_ctx_render(yahr_ctx *ctx, int32_t row0, int32_t row1, const yahr_render_opts *opts,
                    float *d_rgb, void *hip_stream);
/* Cameras for frame batches (opts->frames > 1): n cameras of 12 floats each -- imW imH focalLength lookDir[3]
 * upDir[3] position[3] (Cameras.hs:54-56); floor imW/imH must equal the scene's image size.  n = 0 clears them.
 * Synchronous; call it outside stream capture. */
int yahr_ctx_set_cameras(yahr_ctx *ctx, int32_t n, const float *cameras);
/* Counters of the launches since the last call (synchronises the device). */
int yahr_ctx_stats(yahr_ctx *ctx, yahr_stats *out);

/* Test/diagnostic entry points: run the device functions on arbitrary rays (host buffers, synchronous).
 * rays: n*7 floats (x0, u, tMax).  closest hit -> prim_orig (or -1) and t;  radiance -> n*3 floats.
 * yahr_ctx_trace / yahr_ctx_occluded run a COUNTING build: their box / triangle / sphere tests are added to the context's
 * counters (yahr_ctx_stats), so a test can see what a box test decided. */
int yahr_ctx_trace(yahr_ctx *ctx, int32_t n, const float *rays, int32_t traverse, int32_t *prim_orig, float *t);
int yahr_ctx_occluded(yahr_ctx *ctx, int32_t n, const float *rays, int32_t traverse, int32_t *blocked);
int yahr_ctx_radiance(yahr_ctx *ctx, int32_t n, const float *rays, int32_t traverse, float *rgb);
int yahr_ctx_camera_rays(yahr_ctx *ctx, int32_t n, const float *uv, float *rays);

/* ---------------------------------------------------------------------------------------------
 * 4. One-shot conveniences (host memory in, host memory out, synchronous)
 * ------------------------------------------------------------------------------------------- */

/* rgb_out: (rows rendered)*width*3 floats on the HOST. */
int yahr_render_rows(const yahr_scene *scene, int32_t row0, int32_t row1, int device,
                     const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats_or_null);
/* ---------------------------------------------------------------------------------------------
 * 4b. Multi-GPU: one process, every GPU of the node  (replaces renderEval / renderPar, main.hs:75-96)
 * ------------------------------------------------------------------------------------------- */

/* The scene is resident in the HBM of each device (one context per device, created once); the image is cut into
 * strips of strip_rows rows dealt round-robin to the devices; each device renders its strips on its own stream and
 * ONE exchange step per frame brings them over xGMI to device 0, where the image is assembled, followed by one
 * device-to-host copy.  Transports of the exchange step: */
enum {
  YAHR_GATHER_AUTO = 0,   /* RCCL if librccl can be loaded (lazily, on first use), else PEER; nothing with 1 device */
  YAHR_GATHER_RCCL = 1,   /* grouped ncclSend/ncclRecv (ncclCommInitAll, single process) + strided placement on device 0 */
  YAHR_GATHER_PEER = 2,   /* one strided peer DMA per device (hipMemcpy2DAsync) straight into place on device 0 */
  YAHR_GATHER_NONE = 3    /* reported by yahr_multi_transport for a one-device object: no exchange */
};
typedef struct yahr_multi yahr_multi;
/* devices: n_devices indices, or NULL for 0..n_devices-1 (the first is the root).  gather: YAHR_GATHER_AUTO/RCCL/PEER;
 * naming a transport explicitly runs it even with one device (self send / self copy).  strip_rows <= 0: 8.
 * With YAHR_GATHER_PEER named explicitly a device may be listed several times (each entry gets its own resident
 * context and streams): a rehearsal of the N-device plan on fewer GPUs. */
int yahr_multi_create(const yahr_scene *scene, int n_devices, const int *devices, int gather, int32_t strip_rows,
                      yahr_multi **out);
/* renders opts->frames frames (opts: spp, seed, traverse, collect_stats, frames; sharding fields are set by the object)
 * into the HOST buffer rgb_out (frames*height*width*3 floats); synchronous = yahr_multi_submit + yahr_multi_wait.
 * Repeated calls reuse contexts, buffers and communicators. */
int yahr_multi_render(yahr_multi *m, const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats_or_null);
/* cameras of frame batches, on every device (see yahr_ctx_set_cameras); not while submissions are in flight */
int yahr_multi_set_cameras(yahr_multi *m, int32_t n, const float *cameras);
/* Asynchronous pair.  submit queues the renders of a frame batch on every device, the exchange step and the download
 * into rgb_out (which must stay valid until the matching wait; memory from yahr_pinned_alloc makes the download
 * asynchronous too) and returns; at most TWO submissions may be in flight (two buffer sets), so that the exchange and
 * download of submission i overlap the render of submission i+1.  wait completes the OLDEST submission in flight and
 * returns exactly its counters.  A submission that fails half way drains every stream and leaves the object unusable
 * (every later call fails; destroy it). */
int yahr_multi_submit(yahr_multi *m, const yahr_render_opts *opts, float *rgb_out);
int yahr_multi_wait(yahr_multi *m, yahr_stats *stats_or_null);
/* page-locked host memory for image buffers (hipHostMalloc); NULL on failure */
void *yahr_pinned_alloc(size_t bytes);
void yahr_pinned_free(void *p);
int yahr_multi_transport(const yahr_multi *m);           /* YAHR_GATHER_RCCL / _PEER / _NONE actually in use */
double yahr_multi_last_gather_ms(const yahr_multi *m);   /* device time of the last frame's exchange + placement */
void yahr_multi_destroy(yahr_multi *m);
/* one-shot: create + render + destroy (peer-DMA transport: no communicator is set up for a single frame; one device and
 * one frame: exactly yahr_render_rows over the whole image on device 0) */
int yahr_render_multi(const yahr_scene *scene, int n_devices, const yahr_render_opts *opts,
                      float *rgb_out, yahr_stats *stats_or_null);

/* Image output (main.hs:142 savePngImage (ImageRGBF img)); 8-bit rule: clamp to [0,1], *255, truncate. */
int yahr_write_png(const char *path, const float *rgb, int32_t width, int32_t height);
int yahr_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height);
int yahr_write_pfm(const char *path, const float *rgb, int32_t width, int32_t height);

const char *yahr_last_error(void);
const char *yahr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* YAHR_HIP_H */
