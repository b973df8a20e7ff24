// host_scene.hpp -- host-side mirror of Scene.hs / main.hs:41-60 / Culling.hs for libyahr_hip.so.
// Plain C++17, no HIP.  See include/yahr_hip.h for the C ABI built on top of it.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/yahr_hip.h"

namespace yahr {

struct Vec3 { float x = 0, y = 0, z = 0; };

// Scene.hs:15-42
struct SceneObject {
  enum Kind { Sphere, Triangle, TriangleMesh, Subsampled } kind = Sphere;
  // Sphere
  Vec3 position; float radius = 0;
  // Triangle
  Vec3 p0, p1, p2, n0, n1, n2;
  // TriangleMesh
  std::vector<Vec3> points;
  bool has_normals = false; std::vector<Vec3> normals;
  std::vector<int32_t> tris;              // 3 per triangle
  bool has_smooth = false; std::vector<uint8_t> smooth;
  // Subsampled
  float subsample_size = 0; std::vector<SceneObject> children;
  std::string material_id;
};

struct Material {                          // Scene.hs:45-50
  std::string id; Vec3 ambient, diffuse, specular; float shininess = 0;
};
struct Light { Vec3 pos, spectrum; };      // Lights.hs:7

struct Camera {                            // Cameras.hs:54-56
  float imW = 0, imH = 0, focal = 0; Vec3 look, up, pos;
};

// a primitive after `objects >>= expand` (Scene.hs:61-86), with its material resolved (main.hs:55-60)
struct Primitive {
  int kind = 0;                            // 0 triangle, 1 sphere
  Vec3 p0, p1, p2, n0, n1, n2;             // sphere: p0 = centre
  float radius = 0;
  int32_t mat = -1;
};

struct HostScene {
  int depth = 1;                           // WhittedIntegrator recursionDepth
  int bvh_depth = 16; int split = YAHR_SPLIT_MIDPOINT;
  Camera camera;
  std::vector<Material> materials;
  std::vector<Light> lights;
  std::vector<SceneObject> objects;
  std::vector<SceneObject *> open_subsampled;   // programmatic begin/end nesting

  int builder = 0;                         // YAHR_BUILDER_*: who builds the BVH (yahr_scene_set_builder)

  // built state
  bool built = false;
  int builder_used = 0; double builder_device_ms = 0; std::string builder_note;   // what did, device time, why not the GPU
  std::vector<Primitive> prims;            // expand order
  std::vector<yahr_node> nodes;
  std::vector<yahr_prim> prim_recs;
  std::vector<yahr_prim_normals> prim_normals;
  std::vector<yahr_material> mat_recs;
  std::vector<yahr_light> light_recs;
  yahr_scene flat{};

  void invalidate() { built = false; }
  void build();                            // throws std::runtime_error
};

// parser: throws ParseError
struct ParseError { std::string msg; size_t offset; };
std::unique_ptr<HostScene> parse_scene(const char *text, size_t len, bool lenient);
int parse_parallel_lists();      // lists of this thread's last parse_scene that were parsed by several threads
// one numeric lexeme (decimal, hexadecimal, octal) -> binary32, correctly rounded like GHC's read (scene_parse.cpp)
float parse_float_literal(const char *s, size_t len);

// Scene.hs:61-86 (+ material resolution of main.hs:55-60): appends the expanded primitives of `o`
void expand_object(const SceneObject &o, const std::function<int32_t(const std::string &)> &mat_of,
                   std::vector<Primitive> &out);
// Cameras.hs:58-81: fills tf and vtf (row-major)
void camera_matrices(const Camera &c, float tf[16], float vtf[16]);

// bvh_build.hip: the midpoint BVH built on the GPU with the identical result (node array and leaf order); false = not
// possible here (no device, out of memory, ...), nothing written
bool device_build_bvh(const float *boxes, int32_t n, int max_depth, int split, std::vector<yahr_node> &nodes, std::vector<int32_t> &order,
                      double *device_ms, std::string *why);

void set_error(const std::string &msg);

}  // namespace yahr
