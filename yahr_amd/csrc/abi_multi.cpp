// abi_multi.cpp -- the multi-GPU form of the render path behind the C ABI (include/yahr_hip.h section 4b).
//
// The reference parallelises over independent pixel tiles inside one process (main.hs:75-96: renderEval /
// renderPar).  Here ONE process drives every GPU of the node: the scene is resident in each GPU's HBM (one
// yahr_ctx per device, created once), the image is cut into strips of `strip_rows` rows dealt round-robin to
// the devices, every device renders its strips on its own stream, and there is exactly ONE exchange step per
// frame: the strips travel over xGMI to device 0, where the image is assembled, and one device-to-host copy
// hands it to the caller.  Two transports for that step:
//   * RCCL   grouped ncclSend / ncclRecv (ncclCommInitAll, single process) into a staging buffer on device 0,
//            then strided device-local copies put the strips in place.  librccl is loaded LAZILY (dlopen) the first
//            time a multi-device object asks for it: it is a 570 MB library and one-GPU hosts (and the `yahr`
//            command line on them) never pay for it.
//   * PEER   every device pushes its strips with ONE strided peer DMA (hipMemcpy2DAsync, device to device) straight
//            to their rows of the image on device 0 -- gather and assembly in a single copy, no kernels.
// AUTO uses RCCL when the library can be loaded and falls back to PEER.  With one device there is no exchange at
// all unless the caller forces the transport (tests do, to exercise both legs on a one-GPU box).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "host_scene.hpp"

using yahr::set_error;

namespace {

#define HIP_TRYM(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                        \
      return e_ == hipErrorOutOfMemory ? YAHR_E_NOMEM : YAHR_E_HIP;                        \
    }                                                                                      \
  } while (0)

// ---- RCCL, resolved at run time -------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;                                         // why loading failed
};

Rccl g_rccl;
Rccl *rccl() {
  static std::once_flag once;
  std::call_once(once, [] {
    Rccl &r = g_rccl;
    // a copy that is already mapped (PyTorch-ROCm bundles one under the same soname) is reused
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) { r.why = std::string("dlopen(librccl.so.1): ") + dlerror(); return; }
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.why.empty()) r.why = std::string("missing symbol ") + n; return p; };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.why.empty()) { dlclose(r.handle); r.handle = nullptr; }
  });
  return g_rccl.handle ? &g_rccl : nullptr;
}

#define NCCL_TRY(expr)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      set_error(std::string(#expr) + ": " + rccl()->GetErrorString(r_));                   \
      return YAHR_E_HIP;                                                                   \
    }                                                                                      \
  } while (0)

}  // namespace

struct yahr_multi {
  int n = 0;
  int32_t width = 0, height = 0, strip = 8;
  int transport = YAHR_GATHER_NONE;
  std::vector<int> dev;
  std::vector<yahr_ctx *> ctx;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> done;
  std::vector<float *> d_local;            // per device: its strips, compacted top to bottom
  std::vector<int32_t> rows;               // rows each device owns
  int32_t max_rows = 0;
  float *d_stage = nullptr;                // device 0: n x max_rows x width x 3 (RCCL receive buffers)
  float *d_image = nullptr;                // device 0: height x width x 3
  std::vector<ncclComm_t> comm;
  double last_gather_ms = 0;
  hipEvent_t g0 = nullptr, g1 = nullptr;   // device 0: around the exchange step
};

namespace {

// strips of device r -> their rows of the image: one strided copy (+ one for a ragged last strip).
// `src` holds the device's strips compacted; runs on `st` (any device's stream: the copy engine of that device).
int place_strips(const yahr_multi *m, int r, const float *src, hipStream_t st) {
  const size_t row_bytes = (size_t)m->width * 3 * sizeof(float);
  const size_t strip_bytes = row_bytes * (size_t)m->strip;
  const int32_t rows = m->rows[(size_t)r];
  const int32_t full = rows / m->strip, rest = rows - full * m->strip;
  char *dst0 = (char *)m->d_image + (size_t)r * strip_bytes;              // first strip of device r is strip r
  if (full > 0)
    HIP_TRYM(hipMemcpy2DAsync(dst0, strip_bytes * (size_t)m->n, src, strip_bytes, strip_bytes, (size_t)full,
                              hipMemcpyDeviceToDevice, st));
  if (rest > 0)
    HIP_TRYM(hipMemcpyAsync(dst0 + (size_t)full * strip_bytes * (size_t)m->n, (const char *)src + (size_t)full * strip_bytes,
                            row_bytes * (size_t)rest, hipMemcpyDeviceToDevice, st));
  return YAHR_OK;
}

}  // namespace

extern "C" {

void yahr_multi_destroy(yahr_multi *m) {
  if (!m) return;
  for (int d = 0; d < (int)m->dev.size(); ++d) {
    (void)hipSetDevice(m->dev[(size_t)d]);
    if ((size_t)d < m->comm.size() && m->comm[(size_t)d] && rccl()) (void)rccl()->CommDestroy(m->comm[(size_t)d]);
    if ((size_t)d < m->ctx.size()) yahr_ctx_destroy(m->ctx[(size_t)d]);
    if ((size_t)d < m->d_local.size()) (void)hipFree(m->d_local[(size_t)d]);
    if ((size_t)d < m->done.size() && m->done[(size_t)d]) (void)hipEventDestroy(m->done[(size_t)d]);
    if ((size_t)d < m->stream.size() && m->stream[(size_t)d]) (void)hipStreamDestroy(m->stream[(size_t)d]);
  }
  if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
  (void)hipFree(m->d_stage); (void)hipFree(m->d_image);
  if (m->g0) (void)hipEventDestroy(m->g0);
  if (m->g1) (void)hipEventDestroy(m->g1);
  delete m;
}

int yahr_multi_create(const yahr_scene *scene, int n_devices, const int *devices, int gather, int32_t strip_rows,
                      yahr_multi **out) {
  if (!out) { set_error("null argument"); return YAHR_E_ARG; }
  *out = nullptr;
  if (!scene || n_devices < 1 || gather < YAHR_GATHER_AUTO || gather > YAHR_GATHER_PEER) { set_error("bad arguments"); return YAHR_E_ARG; }
  const int have = yahr_device_count();
  if (have < 1) { set_error("no HIP device available (libyahr_hip has no CPU fallback)"); return YAHR_E_NODEVICE; }
  struct Owner { yahr_multi *m; ~Owner() { if (m) yahr_multi_destroy(m); } } own{new yahr_multi()};
  yahr_multi *m = own.m;
  m->n = n_devices; m->width = scene->width; m->height = scene->height; m->strip = strip_rows > 0 ? strip_rows : 8;
  for (int d = 0; d < n_devices; ++d) {
    const int id = devices ? devices[d] : d;
    if (id < 0 || id >= have) { set_error("more devices requested than visible"); return YAHR_E_NODEVICE; }
    for (int e = 0; e < d; ++e) if (m->dev[(size_t)e] == id) { set_error("device listed twice"); return YAHR_E_ARG; }
    m->dev.push_back(id);
  }
  const size_t row_floats = (size_t)m->width * 3;
  for (int d = 0; d < n_devices; ++d) {
    yahr_ctx *c = nullptr;
    int rc = yahr_ctx_create(scene, m->dev[(size_t)d], &c);       // scene replicated: resident in every GPU's HBM
    if (rc) return rc;
    m->ctx.push_back(c);
    HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
    hipStream_t s = nullptr; HIP_TRYM(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); m->stream.push_back(s);
    hipEvent_t e = nullptr; HIP_TRYM(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m->done.push_back(e);
    const int32_t rows = yahr_shard_rows(m->height, n_devices, d, m->strip);
    m->rows.push_back(rows);
    if (rows > m->max_rows) m->max_rows = rows;
    float *p = nullptr;
    HIP_TRYM(hipMalloc((void **)&p, sizeof(float) * row_floats * (size_t)(rows > 0 ? rows : 1)));
    m->d_local.push_back(p);
  }
  HIP_TRYM(hipSetDevice(m->dev[0]));
  HIP_TRYM(hipEventCreate(&m->g0)); HIP_TRYM(hipEventCreate(&m->g1));
  // transport
  const bool exchange = n_devices > 1 || gather != YAHR_GATHER_AUTO;      // one device: only when forced (tests)
  if (!exchange) { m->transport = YAHR_GATHER_NONE; *out = m; own.m = nullptr; return YAHR_OK; }
  HIP_TRYM(hipMalloc((void **)&m->d_image, sizeof(float) * row_floats * (size_t)(m->height > 0 ? m->height : 1)));
  int want = gather;
  if (want == YAHR_GATHER_AUTO) want = rccl() ? YAHR_GATHER_RCCL : YAHR_GATHER_PEER;
  if (want == YAHR_GATHER_RCCL) {
    Rccl *r = rccl();
    if (!r) { set_error("RCCL transport requested but librccl could not be loaded: " + g_rccl.why); return YAHR_E_HIP; }
    m->comm.assign((size_t)n_devices, nullptr);
    NCCL_TRY(r->CommInitAll(m->comm.data(), n_devices, m->dev.data()));
    HIP_TRYM(hipSetDevice(m->dev[0]));
    HIP_TRYM(hipMalloc((void **)&m->d_stage, sizeof(float) * row_floats * (size_t)(m->max_rows > 0 ? m->max_rows : 1) * (size_t)n_devices));
  } else {
    // peer DMA: device d writes into device 0's memory
    for (int d = 1; d < n_devices; ++d) {
      int can = 0;
      HIP_TRYM(hipDeviceCanAccessPeer(&can, m->dev[(size_t)d], m->dev[0]));
      if (!can) continue;                                   // hipMemcpy*Async still works, staged by the runtime
      HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
      hipError_t e = hipDeviceEnablePeerAccess(m->dev[0], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); return YAHR_E_HIP; }
      (void)hipGetLastError();
    }
  }
  m->transport = want;
  *out = m; own.m = nullptr;
  return YAHR_OK;
}

int yahr_multi_transport(const yahr_multi *m) { return m ? m->transport : YAHR_GATHER_NONE; }

int yahr_multi_render(yahr_multi *m, const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats) {
  if (!m || !rgb_out) { set_error("null argument"); return YAHR_E_ARG; }
  yahr_render_opts base; if (opts) base = *opts; else yahr_default_opts(&base);
  if (base.frames > 1 || base.tiles != YAHR_TILES_WAVE) { set_error("yahr_multi_render renders one frame in wave tiles"); return YAHR_E_ARG; }
  const size_t row_floats = (size_t)m->width * 3;
  // 1. every device renders its strips on its own stream (asynchronous: all GPUs run concurrently)
  for (int d = 0; d < m->n; ++d) {
    yahr_render_opts o = base;
    o.n_shards = m->n; o.shard = d; o.strip_rows = m->strip;
    int rc = yahr_ctx_render(m->ctx[(size_t)d], 0, m->height, &o, m->d_local[(size_t)d], m->stream[(size_t)d]);
    if (rc) return rc;
  }
  // 2. the one exchange step: strips -> device 0, in place
  const float *image = m->d_local[0];
  HIP_TRYM(hipSetDevice(m->dev[0]));
  if (m->transport != YAHR_GATHER_NONE) {
    image = m->d_image;
    if (m->transport == YAHR_GATHER_RCCL) {
      Rccl *r = rccl();
      HIP_TRYM(hipEventRecord(m->g0, m->stream[0]));
      NCCL_TRY(r->GroupStart());
      for (int d = 0; d < m->n; ++d) {
        const size_t cnt = row_floats * (size_t)m->rows[(size_t)d];
        if (cnt == 0) continue;
        NCCL_TRY(r->Send(m->d_local[(size_t)d], cnt, ncclFloat, 0, m->comm[(size_t)d], m->stream[(size_t)d]));
        NCCL_TRY(r->Recv(m->d_stage + (size_t)d * row_floats * (size_t)m->max_rows, cnt, ncclFloat, d, m->comm[0], m->stream[0]));
      }
      NCCL_TRY(r->GroupEnd());
      HIP_TRYM(hipSetDevice(m->dev[0]));
      for (int d = 0; d < m->n; ++d) {
        int rc = place_strips(m, d, m->d_stage + (size_t)d * row_floats * (size_t)m->max_rows, m->stream[0]);
        if (rc) return rc;
      }
      HIP_TRYM(hipEventRecord(m->g1, m->stream[0]));
    } else {
      HIP_TRYM(hipEventRecord(m->g0, m->stream[0]));
      for (int d = 0; d < m->n; ++d) {
        HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
        int rc = place_strips(m, d, m->d_local[(size_t)d], m->stream[(size_t)d]);     // pushed by the owner of the strips
        if (rc) return rc;
        if (d > 0) HIP_TRYM(hipEventRecord(m->done[(size_t)d], m->stream[(size_t)d]));
      }
      HIP_TRYM(hipSetDevice(m->dev[0]));
      for (int d = 1; d < m->n; ++d) HIP_TRYM(hipStreamWaitEvent(m->stream[0], m->done[(size_t)d], 0));
      HIP_TRYM(hipEventRecord(m->g1, m->stream[0]));
    }
  }
  // 3. one device-to-host copy of the assembled image
  HIP_TRYM(hipMemcpyAsync(rgb_out, image, sizeof(float) * row_floats * (size_t)m->height, hipMemcpyDeviceToHost, m->stream[0]));
  HIP_TRYM(hipStreamSynchronize(m->stream[0]));
  for (int d = 1; d < m->n; ++d) { HIP_TRYM(hipSetDevice(m->dev[(size_t)d])); HIP_TRYM(hipStreamSynchronize(m->stream[(size_t)d])); }
  HIP_TRYM(hipSetDevice(m->dev[0]));
  if (m->transport != YAHR_GATHER_NONE) { float ms = 0; HIP_TRYM(hipEventElapsedTime(&ms, m->g0, m->g1)); m->last_gather_ms = ms; }
  // 4. counters: summed over the devices, kernel time = the slowest device
  yahr_stats total; std::memset(&total, 0, sizeof total);
  int bad = YAHR_OK; std::string bad_msg;
  for (int d = 0; d < m->n; ++d) {
    yahr_stats s; int rc = yahr_ctx_stats(m->ctx[(size_t)d], &s);
    if (rc && rc != YAHR_E_MATERIAL) return rc;
    if (rc == YAHR_E_MATERIAL) { bad = rc; bad_msg = yahr_last_error(); }
    total.primary_rays += s.primary_rays; total.reflection_rays += s.reflection_rays; total.shadow_rays += s.shadow_rays;
    total.box_tests += s.box_tests; total.tri_tests += s.tri_tests; total.sphere_tests += s.sphere_tests;
    total.shaded_hits += s.shaded_hits; total.wave_node_steps += s.wave_node_steps; total.wave_prim_steps += s.wave_prim_steps;
    total.wave_stage1_steps += s.wave_stage1_steps; total.wave_stage2_steps += s.wave_stage2_steps;
    total.bad_material += s.bad_material;
    if (s.kernel_ms > total.kernel_ms) total.kernel_ms = s.kernel_ms;
    if (s.upload_ms > total.upload_ms) total.upload_ms = s.upload_ms;
    if (s.max_wave_cycles > total.max_wave_cycles) total.max_wave_cycles = s.max_wave_cycles;
    total.sum_wave_cycles += s.sum_wave_cycles;
  }
  if (stats) *stats = total;
  if (bad) { set_error(bad_msg); return bad; }
  return YAHR_OK;
}

double yahr_multi_last_gather_ms(const yahr_multi *m) { return m ? m->last_gather_ms : 0.0; }

int yahr_render_multi(const yahr_scene *scene, int n_devices, const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats) {
  if (!scene || !rgb_out || n_devices < 1) { set_error("bad arguments"); return YAHR_E_ARG; }
  yahr_multi *m = nullptr;
  int rc = yahr_multi_create(scene, n_devices, nullptr, YAHR_GATHER_AUTO, opts ? opts->strip_rows : 8, &m);
  if (rc) return rc;
  rc = yahr_multi_render(m, opts, rgb_out, stats);
  std::string keep = rc ? yahr_last_error() : "";
  yahr_multi_destroy(m);
  if (rc) set_error(keep);
  return rc;
}

}  // extern "C"
