// rays_host.cpp -- implementation of rays.hpp, plus a few C entry points so that Python tools can
// use the same OBJ loader and scene description. Build with -ffp-contract=off: the vector helpers
// follow glam's scalar definitions, which round every operation separately.

#include "rays.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rays {
namespace {

inline Vec3 sub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 add(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline Vec3 normalize(Vec3 a) {
    const float r = 1.0f / std::sqrt(dot(a, a));
    return Vec3{a.x * r, a.y * r, a.z * r};
}
inline float fmin_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float fmax_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

[[noreturn]] void raise(int code) { throw Error(code, rt_last_error()); }

// One token of a "v" / "f" line, as str::parse would accept it: the whole token must convert.
bool parse_f32(const std::string& tok, float* out) {
    if (tok.empty() || std::isspace((unsigned char)tok[0])) return false;
    char* end = nullptr;
    const float v = std::strtof(tok.c_str(), &end);
    if (end != tok.c_str() + tok.size()) return false;
    if (tok.find_first_of("xX") != std::string::npos) return false;  // no hex floats in Rust
    *out = v;
    return true;
}
bool parse_usize(const std::string& tok, uint64_t* out) {
    if (tok.empty()) return false;
    size_t i = (tok[0] == '+') ? 1 : 0;
    if (i == tok.size()) return false;
    uint64_t v = 0;
    for (; i < tok.size(); ++i) {
        if (tok[i] < '0' || tok[i] > '9') return false;
        v = v * 10 + (uint64_t)(tok[i] - '0');
    }
    *out = v;
    return true;
}
std::vector<std::string> split_spaces(const std::string& s) {
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(' ', b);
        out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}

}  // namespace

Material Material::make(float albedo, float reflectivity, float roughness) {
    return Material{albedo / 3.14159265358979323846f, reflectivity, roughness};
}

Vec3 calculate_normal(Vec3 a, Vec3 b, Vec3 c) { return normalize(cross(sub(b, a), sub(c, a))); }

void new_triangle(uint32_t a, uint32_t b, uint32_t c, const std::vector<Vec3>& verts, Color color, Triangle* tri,
                  TriangleData* data) {
    if (a >= verts.size() || b >= verts.size() || c >= verts.size())
        throw Error(RT_ERR_INVALID, "new_triangle: vertex index out of range");
    *tri = Triangle{a, b, c};
    data->normal = calculate_normal(verts[a], verts[b], verts[c]);
    data->color = color;
}

Object Object::from(std::vector<Triangle> tris, std::vector<TriangleData> tri_data, std::vector<Vec3> verts,
                    Material mat) {
    if (verts.empty()) throw Error(RT_ERR_INVALID, "Object::from: no vertices");
    if (tris.size() != tri_data.size()) throw Error(RT_ERR_INVALID, "Object::from: tris / tri_data differ in length");
    Vec3 mn = verts[0], mx = verts[0];
    for (const Vec3& t : verts) {
        mn = Vec3{fmin_rs(mn.x, t.x), fmin_rs(mn.y, t.y), fmin_rs(mn.z, t.z)};
        mx = Vec3{fmax_rs(mx.x, t.x), fmax_rs(mx.y, t.y), fmax_rs(mx.z, t.z)};
    }
    Object o;
    o.tris = std::move(tris);
    o.tri_data = std::move(tri_data);
    o.verts = std::move(verts);
    o.bounding_box = BoxShape{mn, mx};
    o.material = mat;
    return o;
}

void read_obj(const std::string& filename, Vec3 offset, Color color, std::vector<Triangle>* tris,
              std::vector<TriangleData>* tri_data, std::vector<Vec3>* verts) {
    std::ifstream in(filename);
    if (!in) throw Error(RT_ERR_INVALID, "read_obj: cannot open " + filename);
    tris->clear();
    tri_data->clear();
    verts->clear();
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.rfind("v ", 0) == 0) {
            float v[3];
            int n = 0;
            for (const std::string& tok : split_spaces(line.substr(2))) {
                float f;
                if (parse_f32(tok, &f)) {
                    if (n < 3) v[n] = f;
                    ++n;
                }
            }
            if (n != 3) throw Error(RT_ERR_INVALID, "read_obj: a 'v' line does not hold exactly three numbers");
            verts->push_back(add(Vec3{v[0], v[1], v[2]}, offset));
        } else if (line.rfind("f ", 0) == 0) {
            uint64_t v[3];
            int n = 0;
            for (const std::string& tok : split_spaces(line.substr(2))) {
                uint64_t u;
                if (parse_usize(tok, &u)) {
                    if (n < 3) v[n] = u;
                    ++n;
                }
            }
            if (n != 3) throw Error(RT_ERR_INVALID, "read_obj: an 'f' line does not hold exactly three indices");
            if (v[0] == 0 || v[1] == 0 || v[2] == 0) throw Error(RT_ERR_INVALID, "read_obj: index 0 in an 'f' line");
            Triangle t;
            TriangleData d;
            new_triangle((uint32_t)(v[0] - 1), (uint32_t)(v[1] - 1), (uint32_t)(v[2] - 1), *verts, color, &t, &d);
            tris->push_back(t);
            tri_data->push_back(d);
        }
    }
}

void optimize_model(Object& obj) {
    const size_t n = obj.tris.size();
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t ia, uint32_t ib) {
        const Triangle &p = obj.tris[ia], &q = obj.tris[ib];
        const uint32_t pmin = std::min({p.a, p.b, p.c}), qmin = std::min({q.a, q.b, q.c});
        if (pmin != qmin) return pmin < qmin;
        return std::max({p.a, p.b, p.c}) > std::max({q.a, q.b, q.c});
    });
    std::vector<Triangle> tris(n);
    std::vector<TriangleData> data(n);
    for (size_t i = 0; i < n; ++i) {
        tris[i] = obj.tris[idx[i]];
        data[i] = obj.tri_data[idx[i]];
    }
    obj.tris.swap(tris);
    obj.tri_data.swap(data);
}

World::World(int device) {
    int dev = device;
    if (rt_ctx_create(&dev, 1, &ctx_) != RT_OK) raise(RT_ERR_DEVICE);
}

World::~World() {
    rt_scene_destroy(scene_);
    rt_ctx_destroy(ctx_);
}

void World::pack() {
    rt_scene_destroy(scene_);
    scene_ = nullptr;
    std::vector<rt_object_desc> descs(objects.size());
    for (size_t i = 0; i < objects.size(); ++i) {
        const Object& o = objects[i];
        rt_object_desc& d = descs[i];
        static_assert(sizeof(Vec3) == 12 && sizeof(Triangle) == 12 && sizeof(TriangleData) == 24, "layout");
        d.verts = reinterpret_cast<const float*>(o.verts.data());
        d.n_verts = (uint32_t)o.verts.size();
        d.tris = reinterpret_cast<const uint32_t*>(o.tris.data());
        d.n_tris = (uint32_t)o.tris.size();
        d.tri_data = reinterpret_cast<const float*>(o.tri_data.data());
        d.albedo_over_pi = o.material.albedo;
        d.reflectivity = o.material.reflectivity;
        d.roughness = o.material.roughness;
    }
    int rc = rt_scene_build(ctx_, descs.data(), (uint32_t)descs.size(), &scene_);
    if (rc != RT_OK) raise(rc);
    // Leave the host copy in the order the device uses, as optimize_model does in the reference.
    for (Object& o : objects) optimize_model(o);
}

std::vector<rt_hit> World::trace_closest(const std::vector<Ray>& rays) {
    static_assert(sizeof(Ray) == sizeof(rt_ray), "layout");
    std::vector<rt_hit> out(rays.size());
    int rc = rt_trace_closest(scene_, reinterpret_cast<const rt_ray*>(rays.data()), rays.size(), out.data());
    if (rc != RT_OK) raise(rc);
    return out;
}

std::vector<uint8_t> World::trace_any(const std::vector<Ray>& rays) {
    std::vector<uint8_t> out(rays.size());
    int rc = rt_trace_any(scene_, reinterpret_cast<const rt_ray*>(rays.data()), rays.size(), out.data());
    if (rc != RT_OK) raise(rc);
    return out;
}

void World::draw(std::vector<uint8_t>& frame, const FrameParams& fp) {
    rt_render_params p{};
    p.width = fp.width;
    p.height = fp.height;
    p.grid = fp.grid;
    p.reflect_depth = fp.reflect_depth;
    p.diffuse_depth = fp.diffuse_depth;
    p.diffuse_scatt = fp.diffuse_scatt;
    p.sun[0] = sun.x;
    p.sun[1] = sun.y;
    p.sun[2] = sun.z;
    p.seed = fp.seed;
    p.shard_index = 0;
    p.shard_count = 1;
    frame.resize((size_t)fp.width * fp.height * 4);
    int rc = rt_render(scene_, &p, frame.data(), nullptr);
    if (rc != RT_OK) raise(rc);
}

uint32_t World::num_tris() const { return rt_scene_num_tris(scene_); }

Vec3 sun_at(float t) {
    return Vec3{std::sin(t / 10.0f) * 100000.0f, 25000.0f, std::cos(t / 10.0f) * 100000.0f};
}

void default_scene(const std::string& teapot_obj, World* world) {
    std::vector<Triangle> t;
    std::vector<TriangleData> d;
    std::vector<Vec3> v;
    world->objects.clear();
    read_obj(teapot_obj, Vec3{-0.9f, -2.0f, 9.0f}, Color::from_u8(0x50, 0xc8, 0x78), &t, &d, &v);
    world->objects.push_back(Object::from(t, d, v, Material::make(1.0f, 0.9f, 0.1f)));
    read_obj(teapot_obj, Vec3{3.0f, -2.0f, 6.0f}, Color::from_u8(0xFF, 0, 0), &t, &d, &v);
    world->objects.push_back(Object::from(t, d, v, Material::make(1.0f, 0.025f, 0.975f)));
    const Color grey = Color::from_u8(128, 128, 128);
    world->objects.push_back(Object::from(
        {Triangle{0, 1, 2}, Triangle{0, 3, 2}},
        {TriangleData{Vec3{0.0f, 1.0f, 0.0f}, grey}, TriangleData{Vec3{0.0f, 1.0f, 0.0f}, grey}},
        {Vec3{-10000.0f, -3.0f, -10000.0f}, Vec3{-10000.0f, -3.0f, 10000.0f}, Vec3{10000.0f, -3.0f, 10000.0f},
         Vec3{10000.0f, -3.0f, -10000.0f}},
        Material::make(1.0f, 0.0f, 1.0f)));
}

}  // namespace rays

// ---- C entry points for tools --------------------------------------------------------------
extern "C" {

// read_obj into malloc'ed arrays (free with rh_free). Returns 0, or -1 with *err set to a static
// or thread-local message.
int rh_read_obj(const char* filename, const float offset[3], const float color[3], float** verts, uint32_t* n_verts,
                uint32_t** tris, uint32_t* n_tris, float** tri_data, const char** err) {
    static thread_local std::string msg;
    try {
        std::vector<rays::Triangle> t;
        std::vector<rays::TriangleData> d;
        std::vector<rays::Vec3> v;
        rays::read_obj(filename, rays::Vec3{offset[0], offset[1], offset[2]}, rays::Color{color[0], color[1], color[2]},
                       &t, &d, &v);
        *n_verts = (uint32_t)v.size();
        *n_tris = (uint32_t)t.size();
        *verts = (float*)std::malloc(std::max<size_t>(1, v.size() * 12));
        *tris = (uint32_t*)std::malloc(std::max<size_t>(1, t.size() * 12));
        *tri_data = (float*)std::malloc(std::max<size_t>(1, d.size() * 24));
        if (!*verts || !*tris || !*tri_data) {
            msg = "out of memory";
            if (err) *err = msg.c_str();
            return -1;
        }
        std::memcpy(*verts, v.data(), v.size() * 12);
        std::memcpy(*tris, t.data(), t.size() * 12);
        std::memcpy(*tri_data, d.data(), d.size() * 24);
        return 0;
    } catch (const std::exception& e) {
        msg = e.what();
        if (err) *err = msg.c_str();
        return -1;
    }
}

void rh_free(void* p) { std::free(p); }

void rh_calculate_normal(const float a[3], const float b[3], const float c[3], float out[3]) {
    rays::Vec3 n = rays::calculate_normal(rays::Vec3{a[0], a[1], a[2]}, rays::Vec3{b[0], b[1], b[2]},
                                          rays::Vec3{c[0], c[1], c[2]});
    out[0] = n.x;
    out[1] = n.y;
    out[2] = n.z;
}

float rh_material_albedo(float albedo) { return rays::Material::make(albedo, 0.0f, 0.0f).albedo; }

}  // extern "C"
