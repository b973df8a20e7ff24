// Host-side AddressSanitizer run of the device half of the ABI (abi_device.cpp, abi_multi.cpp: contexts, queue layout
// of the wavefront form, bands, row shards, frame batches, ray queries): every traversal form on a scene file, images
// compared bit for bit.  Built by tools/sanitize/run_device.sh with the sanitizer on the HOST code only.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "yahr_hip.h"
static int fails = 0;
#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { std::printf("FAIL %s -> %d %s\n", #x, rc_, yahr_last_error()); ++fails; } } while (0)
int main(int argc, char **argv) {
  for (int i = 1; i < argc; ++i) {
    yahr_host_scene *hs = nullptr;
    CHECK(yahr_scene_load(argv[i], 0, &hs));
    if (!hs) continue;
    CHECK(yahr_scene_build(hs));
    const yahr_scene *f = yahr_scene_flat(hs);
    const size_t n = (size_t)f->width * f->height * 3;
    for (int spp : {1, 4, 16}) {
      std::vector<float> ref;
      for (int mode = 1; mode <= 7; ++mode) {               // lane, packet, adaptive, wavefront; its stack walkers; ordered
        for (int bands = 0; bands < (mode == 4 ? 2 : 1); ++bands) {
          if (bands) setenv("YAHR_STREAM_BUDGET_KB", "6000", 1); else unsetenv("YAHR_STREAM_BUDGET_KB");
          yahr_render_opts o; yahr_default_opts(&o); o.spp = spp; o.seed = 7; o.traverse = mode; o.collect_stats = spp == 4;
          std::vector<float> img(n, -1.0f); yahr_stats st;
          CHECK(yahr_render_rows(f, 0, f->height, 0, &o, img.data(), &st));
          if (ref.empty()) ref = img;
          else if (std::memcmp(ref.data(), img.data(), n * sizeof(float)) != 0) { std::printf("FAIL image of mode %d spp %d differs\n", mode, spp); ++fails; }
        }
      }
      unsetenv("YAHR_STREAM_BUDGET_KB");
      // the multi-device object on one device (both transports) and row shards
      yahr_render_opts o; yahr_default_opts(&o); o.spp = spp; o.seed = 7;
      std::vector<float> img(n, -1.0f); yahr_stats st;
      CHECK(yahr_render_multi(f, 1, &o, img.data(), &st));
      if (std::memcmp(ref.data(), img.data(), n * sizeof(float)) != 0) { std::printf("FAIL yahr_render_multi spp %d differs\n", spp); ++fails; }
      // the N-device plan rehearsed on one device (three entries, peer transport), frame batches, two submissions in flight
      const int devs[3] = {0, 0, 0};
      yahr_multi *m = nullptr;
      CHECK(yahr_multi_create(f, 3, devs, YAHR_GATHER_PEER, 5, &m));
      if (m) {
        o.frames = 2;
        std::vector<float> a(2 * n, -1.0f);
        float *b = (float *)yahr_pinned_alloc(2 * n * sizeof(float));
        CHECK(yahr_multi_submit(m, &o, a.data()));
        CHECK(yahr_multi_submit(m, &o, b));
        CHECK(yahr_multi_wait(m, &st));
        CHECK(yahr_multi_wait(m, &st));
        for (int fr = 0; fr < 2; ++fr)
          if (std::memcmp(ref.data(), a.data() + fr * n, n * sizeof(float)) != 0 || std::memcmp(ref.data(), b + fr * n, n * sizeof(float)) != 0) {
            std::printf("FAIL yahr_multi frame %d spp %d differs\n", fr, spp); ++fails;
          }
        yahr_pinned_free(b);
        yahr_multi_destroy(m);
      }
    }
    // the BVH built on the device against the host's
    {
      double ms = 0;
      const int32_t used = yahr_scene_builder_used(hs, &ms);
      yahr_host_scene *h2 = nullptr;
      CHECK(yahr_scene_load(argv[i], 0, &h2));
      if (h2) {
        CHECK(yahr_scene_set_builder(h2, YAHR_BUILDER_HOST));
        CHECK(yahr_scene_build(h2));
        const yahr_scene *g = yahr_scene_flat(h2);
        if (g->n_nodes != f->n_nodes || std::memcmp(g->nodes, f->nodes, sizeof(yahr_node) * (size_t)f->n_nodes) != 0 ||
            std::memcmp(g->prims, f->prims, sizeof(yahr_prim) * (size_t)f->n_prims) != 0) { std::printf("FAIL device-built tree differs from the host's\n"); ++fails; }
        std::printf("builder used: %s (%.2f ms on the device)\n", used == YAHR_BUILDER_DEVICE ? "device" : "host", ms);
        yahr_scene_free(h2);
      }
    }
    std::printf("%s: %dx%d, %d prims done\n", argv[i], f->width, f->height, f->n_prims);
    yahr_scene_free(hs);
  }
  std::printf("%s\n", fails ? "FAILED" : "all equal, no sanitizer report");
  return fails ? 1 : 0;
}
