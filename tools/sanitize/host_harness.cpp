#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include "yahr_hip.h"
#include <string>
#include <vector>
// this harness links the host code without any GPU object: the device BVH builder (bvh_build.hip) is simply "not
// possible here", which is what the host code must cope with anyway
namespace yahr { bool device_build_bvh(const float *, int32_t, int, std::vector<yahr_node> &, std::vector<int32_t> &, double *, std::string *why) { if (why) *why = "no device in the host harness"; return false; } }
int main(int argc, char **argv) {
  for (int i = 1; i < argc; ++i) {
    yahr_host_scene *hs = nullptr;
    int rc = yahr_scene_load(argv[i], 0, &hs);
    if (rc) { printf("%s: rc %d %s\n", argv[i], rc, yahr_last_error()); continue; }
    rc = yahr_scene_build(hs);
    const yahr_scene *f = yahr_scene_flat(hs);
    printf("%s: build rc %d, %d prims, %d nodes, parallel lists %d\n", argv[i], rc, f ? f->n_prims : -1, f ? f->n_nodes : -1, yahr_debug_parallel_lists());
    if (f) {
      std::vector<float> img((size_t)640 * 400 * 3);
      for (size_t k = 0; k < img.size(); ++k) img[k] = (float)((k * 2654435761u) % 1000) / 999.0f;
      rc = yahr_write_png("o.png", img.data(), 640, 400);
      printf("png rc %d\n", rc);
    }
    yahr_scene_free(hs);
  }
}
