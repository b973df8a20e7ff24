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
//
// A submission renders a BATCH of frames (opts->frames, cameras of yahr_multi_set_cameras) and is asynchronous:
// yahr_multi_submit queues the renders (one stream per device), the exchange and the device-to-host copy (a second
// stream per device) and returns; yahr_multi_wait completes the oldest submission.  Two buffer sets, so the exchange
// and download of submission i run while submission i+1 renders -- what bench.py's ShardPipeline does across
// processes (yahr_amd/dist.py), here inside one.  Counters are copied out stream-ordered after every render, so each
// submission reports exactly its own.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
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

// internal entry points of abi_device.cpp (not part of the C ABI)
hipError_t yahr_internal_dev_alloc(void **p, size_t bytes);
int yahr_internal_ctx_counters_async(yahr_ctx *c, unsigned long long *host_pinned, void *hip_stream);
void yahr_internal_counters_to_stats(const unsigned long long *h, yahr_stats *out);
int yahr_internal_counter_words(void);

struct yahr_multi {
  int n = 0;
  int32_t width = 0, height = 0, strip = 8;
  int transport = YAHR_GATHER_NONE;
  bool poisoned = false;                   // a submission failed half way: streams were drained, the object refuses further work
  std::vector<int> dev;
  std::vector<yahr_ctx *> ctx;
  std::vector<hipStream_t> rs, cs;         // per device: render stream, copy (exchange / download) stream
  std::vector<int32_t> rows;               // rows each device owns
  int32_t max_rows = 0;
  int frames_cap = 0;                      // frames the buffers below hold
  struct Slot {
    std::vector<float *> d_local;          // per device: its strips of every frame, [frames][rows_d][width][3]
    float *d_stage = nullptr;              // device 0, RCCL: [n][frames][max_rows][width][3] receive buffers
    float *d_image = nullptr;              // device 0: [frames][height][width][3]
    std::vector<hipEvent_t> t0, t1, c1, sent;   // per device: around its render; after its counter copy; after its part of the exchange
    hipEvent_t g0 = nullptr, g1 = nullptr, done = nullptr;   // device 0: around the exchange step; after the download
    std::vector<unsigned long long *> h_ctr;   // per device: pinned copy of its counters after the render
    bool busy = false;
    int frames = 0;
  } slot[2];
  unsigned long long submitted = 0, completed = 0;
  std::vector<ncclComm_t> comm;
  double last_gather_ms = 0;
};

namespace {

// strips of device r (one frame) -> their rows of the image: one strided copy (+ one for a ragged last strip).
// `src` holds the device's strips compacted; runs on `st` (any device's stream: the copy engine of that device).
int place_strips(const yahr_multi *m, int r, const float *src, float *image, hipStream_t st) {
  const size_t row_bytes = (size_t)m->width * 3 * sizeof(float);
  const size_t strip_bytes = row_bytes * (size_t)m->strip;
  const int32_t rows = m->rows[(size_t)r];
  const int32_t full = rows / m->strip, rest = rows - full * m->strip;
  char *dst0 = (char *)image + (size_t)r * strip_bytes;                   // first strip of device r is strip r
  if (full > 0)
    HIP_TRYM(hipMemcpy2DAsync(dst0, strip_bytes * (size_t)m->n, src, strip_bytes, strip_bytes, (size_t)full,
                              hipMemcpyDeviceToDevice, st));
  if (rest > 0)
    HIP_TRYM(hipMemcpyAsync(dst0 + (size_t)full * strip_bytes * (size_t)m->n, (const char *)src + (size_t)full * strip_bytes,
                            row_bytes * (size_t)rest, hipMemcpyDeviceToDevice, st));
  return YAHR_OK;
}

void free_buffers(yahr_multi *m) {
  for (auto &s : m->slot) {
    for (size_t d = 0; d < s.d_local.size(); ++d) { (void)hipSetDevice(m->dev[d]); (void)hipFree(s.d_local[d]); }
    s.d_local.clear();
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    (void)hipFree(s.d_stage); (void)hipFree(s.d_image);
    s.d_stage = nullptr; s.d_image = nullptr;
  }
  m->frames_cap = 0;
}

// buffers of both slots for batches of up to `frames` frames (only ever called with nothing in flight)
int ensure_buffers(yahr_multi *m, int frames) {
  if (frames <= m->frames_cap) return YAHR_OK;
  free_buffers(m);
  const size_t row_floats = (size_t)m->width * 3;
  for (auto &s : m->slot) {
    s.d_local.assign((size_t)m->n, nullptr);
    for (int d = 0; d < m->n; ++d) {
      HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
      const size_t rows = (size_t)(m->rows[(size_t)d] > 0 ? m->rows[(size_t)d] : 1);
      HIP_TRYM(yahr_internal_dev_alloc((void **)&s.d_local[(size_t)d], sizeof(float) * row_floats * rows * (size_t)frames));
    }
    HIP_TRYM(hipSetDevice(m->dev[0]));
    if (m->transport != YAHR_GATHER_NONE)
      HIP_TRYM(yahr_internal_dev_alloc((void **)&s.d_image, sizeof(float) * row_floats * (size_t)(m->height > 0 ? m->height : 1) * (size_t)frames));
    if (m->transport == YAHR_GATHER_RCCL)
      HIP_TRYM(yahr_internal_dev_alloc((void **)&s.d_stage, sizeof(float) * row_floats * (size_t)(m->max_rows > 0 ? m->max_rows : 1) *
                                                               (size_t)m->n * (size_t)frames));
  }
  m->frames_cap = frames;
  return YAHR_OK;
}

// a submission went wrong half way: leave nothing queued behind, then refuse further work
int poison(yahr_multi *m, int rc, bool group_open) {
  const std::string keep = yahr_last_error();
  if (group_open && rccl()) (void)rccl()->GroupEnd();
  for (int d = 0; d < m->n; ++d) {
    (void)hipSetDevice(m->dev[(size_t)d]);
    (void)hipStreamSynchronize(m->rs[(size_t)d]); (void)hipStreamSynchronize(m->cs[(size_t)d]);
  }
  (void)hipGetLastError();
  m->poisoned = true;
  for (auto &s : m->slot) s.busy = false;
  m->completed = m->submitted;
  set_error(keep + " (the yahr_multi object is unusable after a failed submission: destroy it)");
  return rc;
}

}  // namespace

extern "C" {

void *yahr_pinned_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc failed"); return nullptr; }
  return p;
}
void yahr_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

void yahr_multi_destroy(yahr_multi *m) {
  if (!m) return;
  for (int d = 0; d < (int)m->dev.size(); ++d) {           // nothing of ours may still be running
    (void)hipSetDevice(m->dev[(size_t)d]);
    if ((size_t)d < m->rs.size() && m->rs[(size_t)d]) (void)hipStreamSynchronize(m->rs[(size_t)d]);
    if ((size_t)d < m->cs.size() && m->cs[(size_t)d]) (void)hipStreamSynchronize(m->cs[(size_t)d]);
  }
  free_buffers(m);
  for (auto &s : m->slot) {
    for (size_t d = 0; d < m->dev.size(); ++d) {
      (void)hipSetDevice(m->dev[d]);
      if (d < s.t0.size() && s.t0[d]) (void)hipEventDestroy(s.t0[d]);
      if (d < s.t1.size() && s.t1[d]) (void)hipEventDestroy(s.t1[d]);
      if (d < s.sent.size() && s.sent[d]) (void)hipEventDestroy(s.sent[d]);
      if (d < s.c1.size() && s.c1[d]) (void)hipEventDestroy(s.c1[d]);
      if (d < s.h_ctr.size() && s.h_ctr[d]) (void)hipHostFree(s.h_ctr[d]);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (s.g0) (void)hipEventDestroy(s.g0);
    if (s.g1) (void)hipEventDestroy(s.g1);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (int d = 0; d < (int)m->dev.size(); ++d) {
    (void)hipSetDevice(m->dev[(size_t)d]);
    if ((size_t)d < m->comm.size() && m->comm[(size_t)d] && rccl()) (void)rccl()->CommDestroy(m->comm[(size_t)d]);
    if ((size_t)d < m->ctx.size()) yahr_ctx_destroy(m->ctx[(size_t)d]);
    if ((size_t)d < m->rs.size() && m->rs[(size_t)d]) (void)hipStreamDestroy(m->rs[(size_t)d]);
    if ((size_t)d < m->cs.size() && m->cs[(size_t)d]) (void)hipStreamDestroy(m->cs[(size_t)d]);
  }
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
    // A device may be listed more than once ONLY with the peer-DMA transport named explicitly: a rehearsal of the
    // N-device plan (strip placement, cross-stream events, per-device counters) on fewer GPUs -- each entry gets its
    // own resident context and streams.  RCCL refuses two ranks on one device.
    for (int e = 0; e < d; ++e)
      if (m->dev[(size_t)e] == id && gather != YAHR_GATHER_PEER) { set_error("device listed twice (allowed with YAHR_GATHER_PEER only)"); return YAHR_E_ARG; }
    m->dev.push_back(id);
  }
  const int words = yahr_internal_counter_words();
  for (int d = 0; d < n_devices; ++d) {
    yahr_ctx *c = nullptr;
    int rc = yahr_ctx_create(scene, m->dev[(size_t)d], &c);       // scene replicated: resident in every GPU's HBM
    if (rc) return rc;
    m->ctx.push_back(c);
    HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
    hipStream_t s = nullptr;
    HIP_TRYM(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); m->rs.push_back(s);
    s = nullptr;
    HIP_TRYM(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); m->cs.push_back(s);
    for (auto &sl : m->slot) {
      hipEvent_t e = nullptr;
      HIP_TRYM(hipEventCreate(&e)); sl.t0.push_back(e); e = nullptr;
      HIP_TRYM(hipEventCreate(&e)); sl.t1.push_back(e); e = nullptr;
      HIP_TRYM(hipEventCreateWithFlags(&e, hipEventDisableTiming)); sl.sent.push_back(e); e = nullptr;
      HIP_TRYM(hipEventCreateWithFlags(&e, hipEventDisableTiming)); sl.c1.push_back(e);
      unsigned long long *h = nullptr;
      HIP_TRYM(hipHostMalloc((void **)&h, sizeof(unsigned long long) * (size_t)words, hipHostMallocDefault)); sl.h_ctr.push_back(h);
    }
    const int32_t rows = yahr_shard_rows(m->height, n_devices, d, m->strip);
    m->rows.push_back(rows);
    if (rows > m->max_rows) m->max_rows = rows;
  }
  HIP_TRYM(hipSetDevice(m->dev[0]));
  for (auto &sl : m->slot) { HIP_TRYM(hipEventCreate(&sl.g0)); HIP_TRYM(hipEventCreate(&sl.g1)); HIP_TRYM(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming)); }
  // transport
  const bool exchange = n_devices > 1 || gather != YAHR_GATHER_AUTO;      // one device: only when forced (tests)
  int want = YAHR_GATHER_NONE;
  if (exchange) {
    want = gather;
    if (want == YAHR_GATHER_AUTO) want = rccl() ? YAHR_GATHER_RCCL : YAHR_GATHER_PEER;
    if (want == YAHR_GATHER_RCCL) {
      Rccl *r = rccl();
      if (!r) { set_error("RCCL transport requested but librccl could not be loaded: " + g_rccl.why); return YAHR_E_HIP; }
      m->comm.assign((size_t)n_devices, nullptr);
      NCCL_TRY(r->CommInitAll(m->comm.data(), n_devices, m->dev.data()));
    } else {
      // peer DMA: device d writes into device 0's memory
      for (int d = 1; d < n_devices; ++d) {
        if (m->dev[(size_t)d] == m->dev[0]) continue;
        int can = 0;
        HIP_TRYM(hipDeviceCanAccessPeer(&can, m->dev[(size_t)d], m->dev[0]));
        if (!can) continue;                                   // hipMemcpy*Async still works, staged by the runtime
        HIP_TRYM(hipSetDevice(m->dev[(size_t)d]));
        hipError_t e = hipDeviceEnablePeerAccess(m->dev[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); return YAHR_E_HIP; }
        (void)hipGetLastError();
      }
    }
  }
  m->transport = want;
  int rc = ensure_buffers(m, 1);
  if (rc) return rc;
  *out = m; own.m = nullptr;
  return YAHR_OK;
}

int yahr_multi_transport(const yahr_multi *m) { return m ? m->transport : YAHR_GATHER_NONE; }

int yahr_multi_set_cameras(yahr_multi *m, int32_t n, const float *cameras) {
  if (!m) { set_error("null argument"); return YAHR_E_ARG; }
  if (m->submitted != m->completed) { set_error("yahr_multi_set_cameras with submissions in flight"); return YAHR_E_ARG; }
  for (int d = 0; d < m->n; ++d) {
    int rc = yahr_ctx_set_cameras(m->ctx[(size_t)d], n, cameras);
    if (rc) return rc;
  }
  return YAHR_OK;
}

int yahr_multi_submit(yahr_multi *m, const yahr_render_opts *opts, float *rgb_out) {
  if (!m || !rgb_out) { set_error("null argument"); return YAHR_E_ARG; }
  if (m->poisoned) { set_error("this yahr_multi object failed in an earlier submission: destroy it"); return YAHR_E_HIP; }
  yahr_render_opts base; if (opts) base = *opts; else yahr_default_opts(&base);
  if (base.tiles != YAHR_TILES_WAVE) { set_error("yahr_multi renders in wave tiles"); return YAHR_E_ARG; }
  if (m->submitted - m->completed >= 2) { set_error("two submissions are already in flight: yahr_multi_wait first"); return YAHR_E_ARG; }
  const int frames = base.frames > 1 ? base.frames : 1;
  if (frames > m->frames_cap) {
    if (m->submitted != m->completed) { set_error("a larger frame batch needs new buffers: yahr_multi_wait for the submission in flight first"); return YAHR_E_ARG; }
    int rc = ensure_buffers(m, frames);
    if (rc) return rc;                                        // nothing was queued: the object stays usable
  }
  yahr_multi::Slot &s = m->slot[m->submitted & 1];
  const size_t row_floats = (size_t)m->width * 3;
  const size_t frame_floats = row_floats * (size_t)m->height;
  // 1. every device renders its strips of every frame on its own stream (asynchronous: all GPUs run concurrently),
  //    then hands out its counters, stream-ordered, so that they are exactly this submission's
  for (int d = 0; d < m->n; ++d) {
    yahr_render_opts o = base;
    o.n_shards = m->n; o.shard = d; o.strip_rows = m->strip; o.frames = frames; o.frame_rows = 0;
    if (hipSetDevice(m->dev[(size_t)d]) != hipSuccess || hipEventRecord(s.t0[(size_t)d], m->rs[(size_t)d]) != hipSuccess) { set_error("hipEventRecord failed"); return poison(m, YAHR_E_HIP, false); }
    int rc = yahr_ctx_render(m->ctx[(size_t)d], 0, m->height, &o, s.d_local[(size_t)d], m->rs[(size_t)d]);
    if (rc) return d == 0 && rc == YAHR_E_ARG ? rc : poison(m, rc, false);     // bad options are caught on the first device, before anything is queued
    if (hipSetDevice(m->dev[(size_t)d]) != hipSuccess || hipEventRecord(s.t1[(size_t)d], m->rs[(size_t)d]) != hipSuccess) { set_error("hipEventRecord failed"); return poison(m, YAHR_E_HIP, false); }
    rc = yahr_internal_ctx_counters_async(m->ctx[(size_t)d], s.h_ctr[(size_t)d], m->rs[(size_t)d]);
    if (rc) return poison(m, rc, false);
    if (hipEventRecord(s.c1[(size_t)d], m->rs[(size_t)d]) != hipSuccess) { set_error("hipEventRecord failed"); return poison(m, YAHR_E_HIP, false); }
    if (hipStreamWaitEvent(m->cs[(size_t)d], s.t1[(size_t)d], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return poison(m, YAHR_E_HIP, false); }
  }
  // 2. the one exchange step per submission, on the copy streams: strips of every frame -> device 0, in place
  const float *image = s.d_local[0];
  bool group_open = false;
#define STEP(expr) do { if ((expr) != hipSuccess) { set_error(std::string(#expr) + " failed"); return poison(m, YAHR_E_HIP, group_open); } } while (0)
  STEP(hipSetDevice(m->dev[0]));
  if (m->transport != YAHR_GATHER_NONE) {
    image = s.d_image;
    STEP(hipEventRecord(s.g0, m->cs[0]));
    if (m->transport == YAHR_GATHER_RCCL) {
      Rccl *r = rccl();
      if (r->GroupStart() != ncclSuccess) { set_error("ncclGroupStart failed"); return poison(m, YAHR_E_HIP, false); }
      group_open = true;
      for (int d = 0; d < m->n; ++d) {
        const size_t cnt = row_floats * (size_t)m->rows[(size_t)d] * (size_t)frames;
        if (cnt == 0) continue;
        float *stage = s.d_stage + (size_t)d * row_floats * (size_t)m->max_rows * (size_t)frames;
        if (r->Send(s.d_local[(size_t)d], cnt, ncclFloat, 0, m->comm[(size_t)d], m->cs[(size_t)d]) != ncclSuccess ||
            r->Recv(stage, cnt, ncclFloat, d, m->comm[0], m->cs[0]) != ncclSuccess) { set_error("ncclSend / ncclRecv failed"); return poison(m, YAHR_E_HIP, true); }
      }
      group_open = false;
      if (r->GroupEnd() != ncclSuccess) { set_error("ncclGroupEnd failed"); return poison(m, YAHR_E_HIP, false); }
      STEP(hipSetDevice(m->dev[0]));
      for (int d = 0; d < m->n; ++d)
        for (int f = 0; f < frames; ++f) {
          const float *stage = s.d_stage + (size_t)d * row_floats * (size_t)m->max_rows * (size_t)frames;
          if (place_strips(m, d, stage + (size_t)f * row_floats * (size_t)m->rows[(size_t)d], s.d_image + (size_t)f * frame_floats, m->cs[0]))
            return poison(m, YAHR_E_HIP, false);
        }
    } else {
      for (int d = 0; d < m->n; ++d) {
        STEP(hipSetDevice(m->dev[(size_t)d]));
        for (int f = 0; f < frames; ++f)                        // pushed by the owner of the strips
          if (place_strips(m, d, s.d_local[(size_t)d] + (size_t)f * row_floats * (size_t)m->rows[(size_t)d], s.d_image + (size_t)f * frame_floats, m->cs[(size_t)d]))
            return poison(m, YAHR_E_HIP, false);
        if (d > 0) STEP(hipEventRecord(s.sent[(size_t)d], m->cs[(size_t)d]));
      }
      STEP(hipSetDevice(m->dev[0]));
      for (int d = 1; d < m->n; ++d) STEP(hipStreamWaitEvent(m->cs[0], s.sent[(size_t)d], 0));
    }
    STEP(hipEventRecord(s.g1, m->cs[0]));
  }
  // 3. one device-to-host copy of the assembled frames (asynchronous when rgb_out is pinned: yahr_pinned_alloc)
  STEP(hipMemcpyAsync(rgb_out, image, sizeof(float) * frame_floats * (size_t)frames, hipMemcpyDeviceToHost, m->cs[0]));
  STEP(hipEventRecord(s.done, m->cs[0]));
  if (m->transport == YAHR_GATHER_RCCL)                         // the senders' part of the group
    for (int d = 1; d < m->n; ++d) { STEP(hipSetDevice(m->dev[(size_t)d])); STEP(hipEventRecord(s.sent[(size_t)d], m->cs[(size_t)d])); }
#undef STEP
  (void)hipSetDevice(m->dev[0]);
  s.busy = true; s.frames = frames;
  ++m->submitted;
  return YAHR_OK;
}

int yahr_multi_wait(yahr_multi *m, yahr_stats *stats) {
  if (!m) { set_error("null argument"); return YAHR_E_ARG; }
  if (m->poisoned) { set_error("this yahr_multi object failed in an earlier submission: destroy it"); return YAHR_E_HIP; }
  if (m->submitted == m->completed) { set_error("yahr_multi_wait without a submission in flight"); return YAHR_E_ARG; }
  yahr_multi::Slot &s = m->slot[m->completed & 1];
  bool ok = hipSetDevice(m->dev[0]) == hipSuccess && hipEventSynchronize(s.done) == hipSuccess;
  for (int d = 0; d < m->n && ok; ++d) {                        // renders (and counter copies) are done; so is every sender
    // (events, not stream synchronisation: the render of the NEXT submission may already be queued behind them)
    ok = hipSetDevice(m->dev[(size_t)d]) == hipSuccess && hipEventSynchronize(s.c1[(size_t)d]) == hipSuccess;
    if (ok && d > 0 && m->transport != YAHR_GATHER_NONE) ok = hipEventSynchronize(s.sent[(size_t)d]) == hipSuccess;
  }
  if (!ok) { set_error("a device reported an error while completing a submission"); return poison(m, YAHR_E_HIP, false); }
  (void)hipSetDevice(m->dev[0]);
  if (m->transport != YAHR_GATHER_NONE) { float ms = 0; if (hipEventElapsedTime(&ms, s.g0, s.g1) == hipSuccess) m->last_gather_ms = ms; }
  // counters: summed over the devices, kernel time = the slowest device
  yahr_stats total; std::memset(&total, 0, sizeof total);
  for (int d = 0; d < m->n; ++d) {
    yahr_stats q; yahr_internal_counters_to_stats(s.h_ctr[(size_t)d], &q);
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.t0[(size_t)d], s.t1[(size_t)d]) == hipSuccess) q.kernel_ms = ms;
    total.primary_rays += q.primary_rays; total.reflection_rays += q.reflection_rays; total.shadow_rays += q.shadow_rays;
    total.box_tests += q.box_tests; total.tri_tests += q.tri_tests; total.sphere_tests += q.sphere_tests;
    total.shaded_hits += q.shaded_hits; total.wave_node_steps += q.wave_node_steps; total.wave_prim_steps += q.wave_prim_steps;
    total.wave_stage1_steps += q.wave_stage1_steps; total.wave_stage2_steps += q.wave_stage2_steps;
    total.tri_tests_culled += q.tri_tests_culled; total.wave_cull_steps += q.wave_cull_steps; total.tri_cull_violations += q.tri_cull_violations;
    total.bad_material += q.bad_material;
    if (q.kernel_ms > total.kernel_ms) total.kernel_ms = q.kernel_ms;
    if (q.max_wave_cycles > total.max_wave_cycles) total.max_wave_cycles = q.max_wave_cycles;
    total.sum_wave_cycles += q.sum_wave_cycles;
  }
  (void)hipGetLastError();
  s.busy = false;
  ++m->completed;
  if (stats) *stats = total;
  if (total.bad_material) {
    set_error("Map.!: given key is not an element in the map (an object with an unknown materialId was shaded)");
    return YAHR_E_MATERIAL;
  }
  return YAHR_OK;
}

int yahr_multi_render(yahr_multi *m, const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats) {
  if (m && m->submitted != m->completed) { set_error("yahr_multi_render with submissions in flight: yahr_multi_wait first"); return YAHR_E_ARG; }
  int rc = yahr_multi_submit(m, opts, rgb_out);
  if (rc) return rc;
  return yahr_multi_wait(m, stats);
}

double yahr_multi_last_gather_ms(const yahr_multi *m) { return m ? m->last_gather_ms : 0.0; }

int yahr_render_multi(const yahr_scene *scene, int n_devices, const yahr_render_opts *opts, float *rgb_out, yahr_stats *stats) {
  if (!scene || !rgb_out || n_devices < 1) { set_error("bad arguments"); return YAHR_E_ARG; }
  yahr_multi *m = nullptr;
  // one-shot: the peer-DMA transport -- a communicator (and loading the 570 MB RCCL library) costs more than the one
  // frame it would carry; the long-lived yahr_multi object is where RCCL pays
  const bool timing = std::getenv("YAHR_TIMING") && std::atoi(std::getenv("YAHR_TIMING")) != 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  const auto t0 = now();
  // One device, one frame: the plain one-shot call.  (A yahr_multi object brings two streams of its own per device -- on
  // this runtime a stream's first use costs ~8 ms, its hardware queue -- and buffers for an exchange that does not happen.)
  if (n_devices == 1 && (!opts || (opts->frames <= 1 && opts->tiles == YAHR_TILES_WAVE))) {
    yahr_render_opts o; if (opts) o = *opts; else yahr_default_opts(&o);
    o.n_shards = 1; o.shard = 0;
    int rc1 = yahr_render_rows(scene, 0, scene->height, 0, &o, rgb_out, stats);
    if (timing) std::fprintf(stderr, "yahr-hip timing: yahr_render_multi on one device: %.1f ms (create, render, download, destroy)\n", ms_since(t0));
    return rc1;
  }
  int rc = yahr_multi_create(scene, n_devices, nullptr, n_devices > 1 ? YAHR_GATHER_PEER : YAHR_GATHER_AUTO, opts ? opts->strip_rows : 8, &m);
  if (rc) return rc;
  const double ms_create = ms_since(t0);
  const auto t1 = now();
  rc = yahr_multi_render(m, opts, rgb_out, stats);
  const double ms_render = ms_since(t1);
  const auto t2 = now();
  std::string keep = rc ? yahr_last_error() : "";
  yahr_multi_destroy(m);
  if (timing) std::fprintf(stderr, "yahr-hip timing: yahr_render_multi: create %.1f ms, render + gather + download %.1f ms, destroy %.1f ms\n",
                           ms_create, ms_render, ms_since(t2));
  if (rc) set_error(keep);
  return rc;
}

}  // extern "C"
