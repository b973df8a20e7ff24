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
      for (int mode = 1; mode <= 4; ++mode) {
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
    }
    std::printf("%s: %dx%d, %d prims done\n", argv[i], f->width, f->height, f->n_prims);
    yahr_scene_free(hs);
  }
  std::printf("%s\n", fails ? "FAILED" : "all equal, no sanitizer report");
  return fails ? 1 : 0;
}
