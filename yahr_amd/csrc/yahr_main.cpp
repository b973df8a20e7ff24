// yahr_main.cpp -- drop-in command line of the reference executable (main.hs:28-38,112-145):
//
//     yahr INPUT OUTPUT [--parallel-mode MODE | -p MODE] [+RTS -N<k> ... [-RTS]]
//
// INPUT is a .yahrr scene (derived-Read text), OUTPUT an image (PNG like the reference; .ppm / .pfm by
// extension).  MODE keeps the reference's values sequential | eval | par (main.hs:134-137) -- they only
// chose a CPU scheduling strategy there and all select the HIP path here -- plus the explicit `hip`.
// `+RTS ... -RTS` (appended by the Blender add-on, render_engine.py:44-47) is accepted; -N<k> is echoed
// in the status line.  Exit status: 0 on success, 1 on any error with the message on stderr, like an
// uncaught Haskell exception.  Environment: YAHR_DEVICES=<n> renders on n GPUs (default: one),
// YAHR_LENIENT=1 accepts the stale bundled scene.yahrr dialect, YAHR_SPP=<n> supersampling extension.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/yahr_hip.h"

static int die(const std::string &msg) {
  std::fprintf(stderr, "yahr: %s\n", msg.c_str());
  return 1;
}

static bool ends_with(const std::string &s, const char *suf) {
  size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char **argv) {
  std::vector<std::string> pos;
  std::string mode = "sequential";                       // optFlag "sequential" "parallel-mode" (main.hs:38)
  int threads = 1;
  bool in_rts = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "+RTS") { in_rts = true; continue; }
    if (a == "-RTS") { in_rts = false; continue; }
    if (in_rts) {
      if (a.rfind("-N", 0) == 0 && a.size() > 2) threads = std::atoi(a.c_str() + 2);
      continue;
    }
    if (a == "-p" || a == "--parallel-mode") {
      if (i + 1 >= argc) return die("option " + a + " needs a value");
      mode = argv[++i]; continue;
    }
    if (a.rfind("--parallel-mode=", 0) == 0) { mode = a.substr(16); continue; }
    if (a == "-h" || a == "--help") {
      std::printf("usage: yahr input output [-p|--parallel-mode sequential|eval|par|hip] [+RTS -N<k> -RTS]\n");
      return 0;
    }
    if (!a.empty() && a[0] == '-') return die("unknown option " + a);
    pos.push_back(a);
  }
  if (pos.size() != 2) return die("usage: yahr input output [-p MODE]   (two positional arguments required)");
  if (mode != "sequential" && mode != "eval" && mode != "par" && mode != "hip")
    return die("Non-exhaustive patterns in case (parallel mode \"" + mode + "\")");   // main.hs:134-137
  if (threads < 1) threads = 1;

  const bool timing = std::getenv("YAHR_TIMING") && std::atoi(std::getenv("YAHR_TIMING")) != 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  const char *len_env = std::getenv("YAHR_LENIENT");
  // The HIP runtime and the kernel image take ~90 ms to come up: bring them up on a thread of their own while this one
  // reads and parses the scene.  (Errors are not this thread's business: the render call reports them.)
  std::thread warm([] { (void)yahr_device_warmup(0); });
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
  yahr_host_scene *hs = nullptr;
  if (yahr_scene_load(pos[0].c_str(), len_env && std::atoi(len_env) != 0, &hs) != YAHR_OK) return die(yahr_last_error());
  const double ms_parse = since(t_start);
  warm.join();                                          // the BVH of a big scene is built on the GPU (yahr_scene_set_builder)
  const yahr_scene *flat = yahr_scene_flat(hs);
  if (!flat) { std::string e = yahr_last_error(); yahr_scene_free(hs); return die(e); }

  const double ms_load = since(t_start);               // read + parse + expand + BVH build + flatten
  int n_dev = yahr_device_count();
  if (n_dev < 1) { yahr_scene_free(hs); return die("no HIP device available (this build has no CPU path)"); }
  // One process renders ONE image: a second GPU costs a second HIP context and a second copy of the scene (tens of
  // milliseconds each) before it has traced a ray, more than any frame of the BASELINE configs takes on one GPU (1 ms ..
  // 93 ms).  So one GPU unless YAHR_DEVICES=<n> asks for more (row strips over n GPUs, one peer-DMA gather: yahr_multi).
  { int want = 1; if (const char *e = std::getenv("YAHR_DEVICES")) want = std::atoi(e); if (want < 1) want = 1; if (want < n_dev) n_dev = want; }

  yahr_render_opts opts; yahr_default_opts(&opts);
  if (const char *e = std::getenv("YAHR_SPP")) { int s = std::atoi(e); if (s >= 1) opts.spp = s; }
  if (const char *e = std::getenv("YAHR_STATS")) opts.collect_stats = std::atoi(e) != 0;
  const int n_batches = yahr_n_batches(threads, flat->width, flat->height);
  // main.hs:140-141 prints this line before the (lazy) render is forced
  std::printf("%d threads, %d batches, parallel %s\n", threads, n_batches, mode.c_str());
  std::fflush(stdout);

  std::vector<float> rgb((size_t)flat->width * flat->height * 3, 0.0f);
  yahr_stats st;
  const auto t_render = std::chrono::steady_clock::now();
  int rc = yahr_render_multi(flat, n_dev, &opts, rgb.data(), &st);
  const double ms_render = since(t_render);            // HIP start-up + upload + kernel + download
  if (rc != YAHR_OK) { std::string e = yahr_last_error(); yahr_scene_free(hs); return die(e); }
  const double rays = (double)(st.primary_rays + st.reflection_rays + st.shadow_rays);
  std::fprintf(stderr, "yahr-hip: %d x gfx950, %dx%d, %.3f ms kernel, %.1f Mpixels/s", n_dev, flat->width, flat->height,
               st.kernel_ms, st.kernel_ms > 0 ? (double)flat->width * flat->height / st.kernel_ms / 1e3 : 0.0);
  if (opts.collect_stats)   // ray counts exist in the counted (slower) kernel build only: YAHR_STATS=1
    std::fprintf(stderr, ", %.0f rays, %.1f Mrays/s", rays, st.kernel_ms > 0 ? rays / st.kernel_ms / 1e3 : 0.0);
  std::fprintf(stderr, "\n");

  if (flat->width <= 0 || flat->height <= 0) { yahr_scene_free(hs); return die("image has zero size"); }
  const auto t_write = std::chrono::steady_clock::now();
  if (ends_with(pos[1], ".ppm")) rc = yahr_write_ppm(pos[1].c_str(), rgb.data(), flat->width, flat->height);
  else if (ends_with(pos[1], ".pfm")) rc = yahr_write_pfm(pos[1].c_str(), rgb.data(), flat->width, flat->height);
  else rc = yahr_write_png(pos[1].c_str(), rgb.data(), flat->width, flat->height);   // savePngImage (main.hs:142)
  if (timing)
    std::fprintf(stderr, "yahr-hip timing: load+build %.1f ms (read+parse %.1f, HIP start-up on a second thread meanwhile), device "
                         "(upload+render+download) %.1f ms, encode+write %.1f ms\n", ms_load, ms_parse, ms_render, since(t_write));
  yahr_scene_free(hs);
  if (rc != YAHR_OK) return die(yahr_last_error());
  return 0;
}
