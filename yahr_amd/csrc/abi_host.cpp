// abi_host.cpp -- the host-only part of the C ABI (include/yahr_hip.h sections 2 and image output):
// scene loading/assembly, Sampling.hs tile windows, PNG/PPM/PFM writers.  No HIP calls here.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <cstring>
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <algorithm>
#include <cstdlib>

#include "host_scene.hpp"

namespace yahr {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
}  // namespace yahr

using yahr::HostScene;
using yahr::set_error;

struct yahr_host_scene { std::unique_ptr<HostScene> hs; };

namespace {
inline yahr::Vec3 ld3(const float p[3]) { yahr::Vec3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; }

template <class F> int guarded(F f) {
  try { return f(); }
  catch (const yahr::ParseError &e) {
    set_error("Prelude.read: no parse (" + e.msg + " at byte " + std::to_string(e.offset) + ")");
    return YAHR_E_PARSE;
  } catch (const std::bad_alloc &) { set_error("out of memory"); return YAHR_E_NOMEM; }
  catch (const std::exception &e) { set_error(e.what()); return YAHR_E_ARG; }
}

std::vector<yahr::SceneObject> &target_list(HostScene &hs) {
  return hs.open_subsampled.empty() ? hs.objects : hs.open_subsampled.back()->children;
}
}  // namespace

extern "C" {

const char *yahr_last_error(void) { return yahr::g_last_error.c_str(); }
const char *yahr_version(void) { return "yahr_hip 0.2 (gfx950)"; }

int yahr_scene_parse(const char *text, size_t len, int lenient, yahr_host_scene **out) {
  if (!text || !out) { set_error("null argument"); return YAHR_E_ARG; }
  return guarded([&] {
    auto hs = yahr::parse_scene(text, len, lenient != 0);
    *out = new yahr_host_scene{std::move(hs)};
    return (int)YAHR_OK;
  });
}

int yahr_parse_float32(const char *lexeme, size_t len, float *out) {
  if (!lexeme || !out || len == 0) { set_error("null argument"); return YAHR_E_ARG; }
  *out = yahr::parse_float_literal(lexeme, len);
  return YAHR_OK;
}

int32_t yahr_debug_parallel_lists(void) { return (int32_t)yahr::parse_parallel_lists(); }

int yahr_scene_load(const char *path, int lenient, yahr_host_scene **out) {
  if (!path || !out) { set_error("null argument"); return YAHR_E_ARG; }
  // a regular file is mapped (a 150 MB scene is not copied twice through a stream buffer); anything else -- a pipe, a
  // device -- is read to its end
  const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) {
    set_error(std::string(path) + ": openFile: " + (errno == ENOENT ? "does not exist (No such file or directory)" : std::strerror(errno)));
    return YAHR_E_IO;
  }
  struct stat sb;
  if (::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
    void *m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      ::close(fd);
      (void)::madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
      const int rc = yahr_scene_parse((const char *)m, (size_t)sb.st_size, lenient, out);
      ::munmap(m, (size_t)sb.st_size);
      return rc;
    }
  }
  std::string text;
  char buf[1 << 16];
  for (;;) {
    const ssize_t n = ::read(fd, buf, sizeof buf);
    if (n < 0) {
      if (errno == EINTR) continue;
      set_error(std::string(path) + ": " + std::strerror(errno)); ::close(fd); return YAHR_E_IO;
    }
    if (n == 0) break;
    text.append(buf, (size_t)n);
  }
  ::close(fd);
  return yahr_scene_parse(text.data(), text.size(), lenient, out);
}

int yahr_scene_new(yahr_host_scene **out) {
  if (!out) { set_error("null argument"); return YAHR_E_ARG; }
  *out = new yahr_host_scene{std::make_unique<HostScene>()};
  return YAHR_OK;
}

void yahr_scene_free(yahr_host_scene *s) { delete s; }

int yahr_scene_set_camera(yahr_host_scene *s, float imW, float imH, float focal,
                          const float look[3], const float up[3], const float pos[3]) {
  if (!s || !look || !up || !pos) { set_error("null argument"); return YAHR_E_ARG; }
  auto &c = s->hs->camera; c.imW = imW; c.imH = imH; c.focal = focal; c.look = ld3(look); c.up = ld3(up); c.pos = ld3(pos);
  s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_set_integrator(yahr_host_scene *s, int32_t d) {
  if (!s) { set_error("null argument"); return YAHR_E_ARG; }
  s->hs->depth = d; s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_set_culling(yahr_host_scene *s, int32_t max_depth, int32_t split_mode) {
  if (!s || (split_mode != YAHR_SPLIT_MIDPOINT && split_mode != YAHR_SPLIT_SAH)) { set_error("bad culling mode"); return YAHR_E_ARG; }
  s->hs->bvh_depth = max_depth; s->hs->split = split_mode; s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_set_builder(yahr_host_scene *s, int32_t builder) {
  if (!s || builder < YAHR_BUILDER_AUTO || builder > YAHR_BUILDER_DEVICE) { set_error("bad builder"); return YAHR_E_ARG; }
  s->hs->builder = builder; s->hs->invalidate(); return YAHR_OK;
}
int32_t yahr_scene_builder_used(const yahr_host_scene *s, double *device_ms) {
  if (!s || !s->hs->built) { if (device_ms) *device_ms = 0; return 0; }
  if (device_ms) *device_ms = s->hs->builder_device_ms;
  return s->hs->builder_used;
}
int yahr_scene_add_material(yahr_host_scene *s, const char *id, const float ambient[3], const float diffuse[3],
                            const float specular[3], float shininess) {
  if (!s || !id || !diffuse || !specular) { set_error("null argument"); return YAHR_E_ARG; }
  yahr::Material m; m.id = id; if (ambient) m.ambient = ld3(ambient);
  m.diffuse = ld3(diffuse); m.specular = ld3(specular); m.shininess = shininess;
  s->hs->materials.push_back(std::move(m)); s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_add_light(yahr_host_scene *s, const float pos[3], const float spectrum[3]) {
  if (!s || !pos || !spectrum) { set_error("null argument"); return YAHR_E_ARG; }
  yahr::Light l; l.pos = ld3(pos); l.spectrum = ld3(spectrum);
  s->hs->lights.push_back(l); s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_add_sphere(yahr_host_scene *s, const float c[3], float r, const char *mid) {
  if (!s || !c || !mid) { set_error("null argument"); return YAHR_E_ARG; }
  yahr::SceneObject o; o.kind = yahr::SceneObject::Sphere; o.position = ld3(c); o.radius = r; o.material_id = mid;
  target_list(*s->hs).push_back(std::move(o)); s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_add_triangle(yahr_host_scene *s, const float p0[3], const float p1[3], const float p2[3],
                            const float n0[3], const float n1[3], const float n2[3], const char *mid) {
  if (!s || !p0 || !p1 || !p2 || !n0 || !n1 || !n2 || !mid) { set_error("null argument"); return YAHR_E_ARG; }
  yahr::SceneObject o; o.kind = yahr::SceneObject::Triangle;
  o.p0 = ld3(p0); o.p1 = ld3(p1); o.p2 = ld3(p2); o.n0 = ld3(n0); o.n1 = ld3(n1); o.n2 = ld3(n2); o.material_id = mid;
  target_list(*s->hs).push_back(std::move(o)); s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_add_mesh(yahr_host_scene *s, int32_t n_points, const float *points, const float *normals,
                        int32_t n_tris, const int32_t *tris, const uint8_t *smooth, int32_t n_smooth, const char *mid) {
  if (!s || !mid || n_points < 0 || n_tris < 0 || (n_points && !points) || (n_tris && !tris)) { set_error("bad mesh arguments"); return YAHR_E_ARG; }
  return guarded([&] {
    yahr::SceneObject o; o.kind = yahr::SceneObject::TriangleMesh; o.material_id = mid;
    o.points.resize((size_t)n_points);
    for (int32_t i = 0; i < n_points; ++i) o.points[(size_t)i] = ld3(points + 3 * (size_t)i);
    if (normals) {
      o.has_normals = true; o.normals.resize((size_t)n_points);
      for (int32_t i = 0; i < n_points; ++i) o.normals[(size_t)i] = ld3(normals + 3 * (size_t)i);
    }
    o.tris.assign(tris, tris + 3 * (size_t)n_tris);
    if (smooth) { o.has_smooth = true; o.smooth.assign(smooth, smooth + (size_t)(n_smooth < 0 ? 0 : n_smooth)); }
    target_list(*s->hs).push_back(std::move(o)); s->hs->invalidate();
    return (int)YAHR_OK;
  });
}
int yahr_scene_begin_subsampled(yahr_host_scene *s, float size) {
  if (!s) { set_error("null argument"); return YAHR_E_ARG; }
  yahr::SceneObject o; o.kind = yahr::SceneObject::Subsampled; o.subsample_size = size;
  auto &lst = target_list(*s->hs);
  lst.push_back(std::move(o));
  s->hs->open_subsampled.push_back(&lst.back());
  s->hs->invalidate(); return YAHR_OK;
}
int yahr_scene_end_subsampled(yahr_host_scene *s) {
  if (!s || s->hs->open_subsampled.empty()) { set_error("no open Subsampled"); return YAHR_E_ARG; }
  s->hs->open_subsampled.pop_back(); return YAHR_OK;
}

int yahr_scene_build(yahr_host_scene *s) {
  if (!s) { set_error("null argument"); return YAHR_E_ARG; }
  if (!s->hs->open_subsampled.empty()) { set_error("unterminated Subsampled"); return YAHR_E_ARG; }
  return guarded([&] { s->hs->build(); return (int)YAHR_OK; });
}
const yahr_scene *yahr_scene_flat(yahr_host_scene *s) {
  if (!s) { set_error("null argument"); return nullptr; }
  if (yahr_scene_build(s) != YAHR_OK) return nullptr;
  return &s->hs->flat;
}
int yahr_scene_info(const yahr_host_scene *s, float *imW, float *imH, int32_t *bvh_depth, int32_t *split_mode, int32_t *n_objects) {
  if (!s) { set_error("null argument"); return YAHR_E_ARG; }
  if (imW) *imW = s->hs->camera.imW;
  if (imH) *imH = s->hs->camera.imH;
  if (bvh_depth) *bvh_depth = s->hs->bvh_depth;
  if (split_mode) *split_mode = s->hs->split;
  if (n_objects) *n_objects = (int32_t)s->hs->objects.size();
  return YAHR_OK;
}

// Sampling.hs:9-21
void yahr_batch_window(int32_t w, int32_t h, int32_t num, int32_t count, int32_t out[4]) {
  long nx = count, ny = 1;
  while (nx % 2 == 0 && 2L * w * ny < (long)h * nx) { nx /= 2; ny *= 2; }
  long j = num / nx, i = num % nx;
  out[0] = (int32_t)std::floor((double)i / (double)nx * (double)w);
  out[1] = (int32_t)std::floor((double)j / (double)ny * (double)h);
  out[2] = (int32_t)std::floor(((double)i + 1.0) / (double)nx * (double)w);
  out[3] = (int32_t)std::floor(((double)j + 1.0) / (double)ny * (double)h);
}
// main.hs:109-110,128-130: roundUpPow2 (((max (32*numThreads) width) * height) `div` 256)
int32_t yahr_n_batches(int32_t num_threads, int32_t width, int32_t height) {
  long m = 32L * num_threads > width ? 32L * num_threads : (long)width;
  long x = (m * height) / 256;
  if (x < 1) x = 1;
  double e = std::ceil(std::log((double)x) / std::log(2.0));
  return (int32_t)std::llround(std::pow(2.0, e));
}

// ---- image output ---------------------------------------------------------------------------------
// JuicyPixels `savePngImage path (ImageRGBF img)` (main.hs:142) converts each component with
// truncate . (255 *) . min 1 . max 0 (Codec.Picture.Saving, restated from memory: JuicyPixels is not in
// the container, so the byte-level PNG is "parity unpinned"; parity is defined on the float buffer).
static inline uint8_t to_ldr(float c) {
  float m = (0.0f <= c) ? c : 0.0f;       // max 0 c   (Haskell max x y = if x <= y then y else x)
  if (c != c) m = 0.0f;                   // max 0 NaN = 0 under that definition
  float n = (1.0f <= m) ? 1.0f : m;       // min 1 m
  return (uint8_t)(255.0f * n);           // truncate
}

static void put_be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void png_chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
  put_be32(out, (uint32_t)data.size());
  size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
  put_be32(out, crc);
}

// One zlib stream made by several threads (a 1080p image: 95 -> ~10 ms): the rows are cut into bands, every band is
// deflated on its own (raw deflate, ended with a sync flush -- byte aligned, no final block -- except the last, which
// is finished), and the pieces are put behind one zlib header with the Adler-32 of the whole (adler32_combine).
// A band cannot refer back into the one before it; at 64 rows and more per band the size of the file hardly changes.
static bool zlib_compress_bands(const std::vector<uint8_t> &raw, size_t row_bytes, size_t rows, std::vector<uint8_t> &out) {
  size_t T = std::min<size_t>(16, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
  if (const char *e = std::getenv("YAHR_PNG_THREADS")) T = (size_t)std::max(1, std::atoi(e));
  T = std::max<size_t>(1, std::min(T, rows / 64));
  std::vector<std::vector<uint8_t>> piece(T);
  std::vector<uLong> adler(T, 0);
  std::vector<char> ok(T, 1);
  auto band = [&](size_t t) {
    const size_t r0 = rows * t / T, r1 = rows * (t + 1) / T;
    const uint8_t *src = raw.data() + r0 * row_bytes;
    const size_t len = (r1 - r0) * row_bytes;
    z_stream z{};
    if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok[t] = 0; return; }
    piece[t].resize(deflateBound(&z, (uLong)len) + 16);
    z.next_in = const_cast<Bytef *>(src); z.avail_in = (uInt)len;
    z.next_out = piece[t].data(); z.avail_out = (uInt)piece[t].size();
    const int rc = deflate(&z, t + 1 == T ? Z_FINISH : Z_SYNC_FLUSH);
    if ((t + 1 == T && rc != Z_STREAM_END) || (t + 1 != T && (rc != Z_OK || z.avail_in != 0 || z.avail_out == 0))) ok[t] = 0;
    piece[t].resize(piece[t].size() - z.avail_out);
    deflateEnd(&z);
    adler[t] = adler32(adler32(0L, Z_NULL, 0), src, (uInt)len);
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < T; ++t) pool.emplace_back(band, t);
  band(0);
  for (auto &th : pool) th.join();
  for (size_t t = 0; t < T; ++t) if (!ok[t]) return false;
  out.clear();
  out.push_back(0x78); out.push_back(0x9C);                           // deflate, 32 K window, default level, no dictionary
  uLong a = adler[0];
  for (size_t t = 0; t < T; ++t) {
    out.insert(out.end(), piece[t].begin(), piece[t].end());
    if (t) a = adler32_combine(a, adler[t], (z_off_t)((rows * (t + 1) / T - rows * t / T) * row_bytes));
  }
  out.push_back((uint8_t)(a >> 24)); out.push_back((uint8_t)(a >> 16)); out.push_back((uint8_t)(a >> 8)); out.push_back((uint8_t)a);
  return true;
}

int yahr_write_png(const char *path, const float *rgb, int32_t width, int32_t height) {
  if (!path || !rgb || width <= 0 || height <= 0) { set_error("bad image arguments"); return YAHR_E_ARG; }
  std::vector<uint8_t> raw((size_t)height * ((size_t)width * 3 + 1));
  for (int32_t v = 0; v < height; ++v) {
    uint8_t *row = raw.data() + (size_t)v * ((size_t)width * 3 + 1);
    row[0] = 0;  // filter: none
    for (int32_t u = 0; u < width * 3; ++u) row[1 + u] = to_ldr(rgb[(size_t)v * width * 3 + u]);
  }
  std::vector<uint8_t> comp;
  if (!zlib_compress_bands(raw, (size_t)width * 3 + 1, (size_t)height, comp)) { set_error("zlib compress failed"); return YAHR_E_IO; }
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr; put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
  png_chunk(out, "IHDR", ihdr); png_chunk(out, "IDAT", comp); png_chunk(out, "IEND", {});
  FILE *f = std::fopen(path, "wb");
  if (!f) { set_error(std::string(path) + ": cannot open for writing"); return YAHR_E_IO; }
  size_t w = std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  if (w != out.size()) { set_error("short write"); return YAHR_E_IO; }
  return YAHR_OK;
}

int yahr_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height) {
  if (!path || !rgb || width <= 0 || height <= 0) { set_error("bad image arguments"); return YAHR_E_ARG; }
  FILE *f = std::fopen(path, "wb");
  if (!f) { set_error(std::string(path) + ": cannot open for writing"); return YAHR_E_IO; }
  std::fprintf(f, "P6\n%d %d\n255\n", width, height);
  std::vector<uint8_t> row((size_t)width * 3);
  for (int32_t v = 0; v < height; ++v) {
    for (int32_t u = 0; u < width * 3; ++u) row[(size_t)u] = to_ldr(rgb[(size_t)v * width * 3 + u]);
    std::fwrite(row.data(), 1, row.size(), f);
  }
  std::fclose(f); return YAHR_OK;
}

int yahr_write_pfm(const char *path, const float *rgb, int32_t width, int32_t height) {
  if (!path || !rgb || width <= 0 || height <= 0) { set_error("bad image arguments"); return YAHR_E_ARG; }
  FILE *f = std::fopen(path, "wb");
  if (!f) { set_error(std::string(path) + ": cannot open for writing"); return YAHR_E_IO; }
  std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);    // little-endian; PFM rows go bottom to top
  for (int32_t v = height - 1; v >= 0; --v) std::fwrite(rgb + (size_t)v * width * 3, sizeof(float), (size_t)width * 3, f);
  std::fclose(f); return YAHR_OK;
}

}  // extern "C"
