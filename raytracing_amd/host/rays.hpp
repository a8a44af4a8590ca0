// rays.hpp -- host-side mirror of the reference crate's scene / object / trace API
// (src/objects.rs, src/opt.rs, src/main.rs), written in C++ because no Rust toolchain exists in
// the build image. Names, argument meaning and data ownership follow the reference; the search
// and the frame run on the GPU through the C ABI of include/rays_hip.h.
//
// Differences a reference user will notice:
//  - pack_model uploads the object set and returns nothing per object; World::pack() replaces
//    the `objects.into_iter().map(pack_model)` line of src/main.rs:454 because the device scene
//    is built from all objects at once.
//  - trace_ray / draw take batches: one call per frame instead of one call per ray.
//  - Errors are C++ exceptions of type rays::Error carrying rt_last_error(), where the reference
//    panics (Cargo.toml:10, panic = "abort").
#ifndef RAYS_HOST_HPP
#define RAYS_HOST_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rays_hip.h"

namespace rays {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct Vec3 {
    float x = 0, y = 0, z = 0;
};

// src/objects.rs:76-81
struct Color {
    float r = 0, g = 0, b = 0;
    static Color from_u8(uint8_t r, uint8_t g, uint8_t b) { return Color{(float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f}; }
};

// src/objects.rs:176-190. `albedo` is stored divided by PI, as Material::new does.
struct Material {
    float albedo = 0, reflectivity = 0, roughness = 0;
    static Material make(float albedo, float reflectivity, float roughness);
};

// src/objects.rs:193-204
struct Triangle {
    uint32_t a = 0, b = 0, c = 0;
};
struct TriangleData {
    Vec3 normal;
    Color color;
};

// src/objects.rs:64-74. inv_dir is implied (dir.recip()).
struct Ray {
    Vec3 origin, dir;
};

// src/objects.rs:241-245
struct BoxShape {
    Vec3 min, max;
};

// src/objects.rs:5-7
Vec3 calculate_normal(Vec3 a, Vec3 b, Vec3 c);
// src/objects.rs:206-220
void new_triangle(uint32_t a, uint32_t b, uint32_t c, const std::vector<Vec3>& verts, Color color, Triangle* tri,
                  TriangleData* data);

// src/objects.rs:25-62
struct Object {
    std::vector<Triangle> tris;
    std::vector<TriangleData> tri_data;
    std::vector<Vec3> verts;
    BoxShape bounding_box;
    Material material;
    static Object from(std::vector<Triangle> tris, std::vector<TriangleData> tri_data, std::vector<Vec3> verts,
                       Material mat);
};

// src/main.rs:38-72: "v x y z" and "f a b c" lines only, 1-based triangle indices.
void read_obj(const std::string& filename, Vec3 offset, Color color, std::vector<Triangle>* tris,
              std::vector<TriangleData>* tri_data, std::vector<Vec3>* verts);

// src/opt.rs:11-36: stable sort by (smallest vertex index ascending, largest descending);
// tri_data is permuted alike. Host-side and idempotent; the device build applies the same order.
void optimize_model(Object& obj);

// Frame constants of src/main.rs:29-36.
struct FrameParams {
    uint32_t width = 800, height = 800;
    uint32_t grid = 2;
    uint32_t reflect_depth = 4, diffuse_depth = 2, diffuse_scatt = 10;
    uint32_t seed = 0;
};

// src/objects.rs:18-22. Owns the device scene once pack() has run.
class World {
   public:
    explicit World(int device = 0);
    ~World();
    World(const World&) = delete;
    World& operator=(const World&) = delete;

    std::vector<Object> objects;
    Vec3 sun;

    // optimize_model + pack_model for every object (src/main.rs:450-454).
    void pack();
    // trace_ray's nearest-hit search for a batch of rays (src/main.rs:93-171).
    std::vector<rt_hit> trace_closest(const std::vector<Ray>& rays);
    // trace_ray's shadow search (src/main.rs:185-233).
    std::vector<uint8_t> trace_any(const std::vector<Ray>& rays);
    // draw (src/main.rs:313-379): frame is width * height * 4 bytes of RGBA8.
    void draw(std::vector<uint8_t>& frame, const FrameParams& p);

    uint32_t num_tris() const;

   private:
    rt_ctx* ctx_ = nullptr;
    rt_scene* scene_ = nullptr;
};

// The animated sun of src/main.rs:476-484 at t seconds after start: a circle of radius 100,000 about
// the y axis at height 25,000.
Vec3 sun_at(float t);

// The default scene of src/main.rs:406-448: two teapots and a two-triangle floor.
void default_scene(const std::string& teapot_obj, World* world);

}  // namespace rays

#endif
