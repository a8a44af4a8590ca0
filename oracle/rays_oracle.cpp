// rays_oracle.cpp -- CPU oracle: a C++ restatement of the reference ray tracer's trace_ray path.
//
// TEST INFRASTRUCTURE ONLY (see rays_oracle.h). Build: g++ -O3 -mavx2 -mfma -ffp-contract=off.
// -ffp-contract=off matters: Rust never fuses a*b+c, so every scalar expression here is rounded
// operation by operation; the only fused operations are the explicit _mm256_fmadd/fmsub calls the
// reference itself makes.
//
// Every routine cites the reference file:line it follows (paths relative to the reference root).

#include "rays_oracle.h"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr float F32_EPS = 1.1920929e-7f;  // f32::EPSILON
constexpr float INF = std::numeric_limits<float>::infinity();
constexpr uint32_t MISS = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------------
// glam 0.27.0 scalar Vec3 (Cargo.toml:17), as recalled from the published crate.
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
// f32::min / f32::max: the non-NaN operand wins.
inline float rmin(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float rmax(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
inline V3 vmin(V3 a, V3 b) { return v3(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z)); }
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline V3 recip(V3 a) { return v3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Ray {
    V3 origin, dir, inv_dir;
    V3 at(float t) const { return origin + dir * t; }  // src/objects.rs:70-74
};
inline Ray make_ray(V3 o, V3 d) { return Ray{o, d, recip(d)}; }

struct Box {
    V3 mn, mx;
};

// src/objects.rs:322-333
inline bool box_intersection_check(const Ray& ray, const Box& b) {
    V3 t1 = (b.mn - ray.origin) * ray.inv_dir;
    V3 t2 = (b.mx - ray.origin) * ray.inv_dir;
    V3 tmin = vmin(t1, t2);
    V3 tmax = vmax(t1, t2);
    float t_near = rmax(rmax(tmin.x, tmin.y), tmin.z);
    float t_far = rmin(rmin(tmax.x, tmax.y), tmax.z);
    return rmin(t_near, 0.0f) <= t_far;
}

// src/objects.rs:273-308
inline bool triangle_ray_hits(V3 va, V3 vb, V3 vc, const Ray& ray, float min, float* t_out,
                              float* u_out, float* v_out) {
    V3 e1 = vb - va;
    V3 e2 = vc - va;
    V3 ray_cross_e2 = cross(ray.dir, e2);
    float det = dot(e1, ray_cross_e2);
    if (det > -F32_EPS && det < F32_EPS) return false;
    float inv_det = 1.0f / det;
    V3 s = ray.origin - va;
    float u = inv_det * dot(s, ray_cross_e2);
    if (!(u >= 0.0f && u <= 1.0f)) return false;  // !(0.0..=1.0).contains(&u)
    V3 s_cross_e1 = cross(s, e1);
    float v = inv_det * dot(ray.dir, s_cross_e1);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * dot(e2, s_cross_e1);
    if (t > F32_EPS && t < min) {
        *t_out = t;
        if (u_out) *u_out = u;
        if (v_out) *v_out = v;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// 8-wide types and kernels (src/objects.rs:222-233, src/simd_accel.rs).
// ---------------------------------------------------------------------------------------------
struct alignas(32) PackedTriangles {
    __m256 e1[3], e2[3], v0[3];
};
struct alignas(32) PackedBoxes {
    __m256 mn[3], mx[3];
};
struct SimdRay {
    __m256 origin[3], dir[3], inv_dir[3];
    bool inv_sign_mask[3];
};

// f32::signum() == 1.0: true for +0.0, positive and +inf; false for -0.0, negative, NaN.
inline bool signum_is_one(float x) { return !(x != x) && !std::signbit(x); }

// src/simd_accel.rs:254-279
inline SimdRay ray_to_avx(const Ray& r) {
    SimdRay s;
    s.origin[0] = _mm256_set1_ps(r.origin.x);
    s.origin[1] = _mm256_set1_ps(r.origin.y);
    s.origin[2] = _mm256_set1_ps(r.origin.z);
    s.dir[0] = _mm256_set1_ps(r.dir.x);
    s.dir[1] = _mm256_set1_ps(r.dir.y);
    s.dir[2] = _mm256_set1_ps(r.dir.z);
    s.inv_dir[0] = _mm256_set1_ps(r.inv_dir.x);
    s.inv_dir[1] = _mm256_set1_ps(r.inv_dir.y);
    s.inv_dir[2] = _mm256_set1_ps(r.inv_dir.z);
    s.inv_sign_mask[0] = signum_is_one(r.inv_dir.x);
    s.inv_sign_mask[1] = signum_is_one(r.inv_dir.y);
    s.inv_sign_mask[2] = signum_is_one(r.inv_dir.z);
    return s;
}

// src/simd_accel.rs:136-142
inline void avx_multi_cross(__m256 r[3], const __m256 a[3], const __m256 b[3]) {
    r[0] = _mm256_fmsub_ps(a[1], b[2], _mm256_mul_ps(b[1], a[2]));
    r[1] = _mm256_fmsub_ps(a[2], b[0], _mm256_mul_ps(b[2], a[0]));
    r[2] = _mm256_fmsub_ps(a[0], b[1], _mm256_mul_ps(b[0], a[1]));
}
// src/simd_accel.rs:145-153
inline __m256 avx_multi_dot(const __m256 a[3], const __m256 b[3]) {
    return _mm256_fmadd_ps(a[2], b[2], _mm256_fmadd_ps(a[1], b[1], _mm256_mul_ps(a[0], b[0])));
}

// src/simd_accel.rs:281-331. u and v are computed there and dropped; here they are handed back.
inline __m256 packed_tris_intersect(const PackedTriangles& p, const SimdRay& ray, __m256 ray_length,
                                    int* mask, __m256* u_out, __m256* v_out) {
    __m256 q[3];
    avx_multi_cross(q, ray.dir, p.e2);
    __m256 a = avx_multi_dot(p.e1, q);
    __m256 f = _mm256_div_ps(_mm256_set1_ps(1.0f), a);
    __m256 s[3];
    s[0] = _mm256_sub_ps(ray.origin[0], p.v0[0]);
    s[1] = _mm256_sub_ps(ray.origin[1], p.v0[1]);
    s[2] = _mm256_sub_ps(ray.origin[2], p.v0[2]);
    __m256 u = _mm256_mul_ps(f, avx_multi_dot(s, q));
    __m256 r[3];
    avx_multi_cross(r, s, p.e1);
    __m256 v = _mm256_mul_ps(f, avx_multi_dot(ray.dir, r));
    __m256 t = _mm256_mul_ps(f, avx_multi_dot(p.e2, r));

    const __m256 zero = _mm256_set1_ps(0.0f);
    const __m256 one = _mm256_set1_ps(1.0f);
    __m256 failed = _mm256_and_ps(_mm256_cmp_ps(a, _mm256_set1_ps(-F32_EPS), _CMP_GT_OQ),
                                  _mm256_cmp_ps(a, _mm256_set1_ps(F32_EPS), _CMP_LT_OQ));
    failed = _mm256_or_ps(failed, _mm256_cmp_ps(u, zero, _CMP_LT_OQ));
    failed = _mm256_or_ps(failed, _mm256_cmp_ps(v, zero, _CMP_LT_OQ));
    __m256 u_plus_v = _mm256_add_ps(u, v);
    failed = _mm256_or_ps(failed, _mm256_cmp_ps(u_plus_v, one, _CMP_GT_OQ));
    failed = _mm256_or_ps(failed, _mm256_cmp_ps(t, zero, _CMP_LT_OQ));
    failed = _mm256_or_ps(failed, _mm256_cmp_ps(t, ray_length, _CMP_GT_OQ));
    __m256 t_results = _mm256_blendv_ps(t, _mm256_set1_ps(-1.0f), failed);
    *mask = _mm256_movemask_ps(t_results);
    if (u_out) *u_out = u;
    if (v_out) *v_out = v;
    return t_results;
}

// src/simd_accel.rs:333-377
inline __m256 packed_boxes_intersect(const PackedBoxes& p, const SimdRay& ray, __m256 ray_length,
                                     int* mask) {
    __m256 ts[2][3];
    for (int k = 0; k < 3; ++k) {
        ts[0][k] = _mm256_mul_ps(_mm256_sub_ps(p.mn[k], ray.origin[k]), ray.inv_dir[k]);
        ts[1][k] = _mm256_mul_ps(_mm256_sub_ps(p.mx[k], ray.origin[k]), ray.inv_dir[k]);
    }
    // avx_conditional_swap_in_place: mask true keeps the order, false swaps.
    __m256 t1[3], t2[3];
    for (int k = 0; k < 3; ++k) {
        t1[k] = ray.inv_sign_mask[k] ? ts[0][k] : ts[1][k];
        t2[k] = ray.inv_sign_mask[k] ? ts[1][k] : ts[0][k];
    }
    __m256 tmin = _mm256_max_ps(_mm256_max_ps(t1[0], t1[1]), t1[2]);
    __m256 tmax = _mm256_min_ps(_mm256_min_ps(_mm256_min_ps(t2[0], t2[1]), t2[2]), ray_length);
    __m256 m = _mm256_cmp_ps(tmin, tmax, _CMP_LE_OQ);
    __m256 t_results = _mm256_blendv_ps(_mm256_set1_ps(-1.0f), _mm256_set1_ps(1.0f), m);
    *mask = _mm256_movemask_ps(t_results);
    return t_results;
}

// ---------------------------------------------------------------------------------------------
// Scene: Object, optimize_model, pack_model.
// ---------------------------------------------------------------------------------------------
struct Tri {
    uint32_t a, b, c;
};
struct TriData {
    V3 normal;
    V3 color;
};
struct Material {
    float albedo, reflectivity, roughness;
};

struct PackedObject {
    std::vector<Tri> tris;          // post-sort
    std::vector<TriData> tri_data;  // post-sort
    std::vector<V3> verts;
    Box bounding_box;
    Material material;
    std::vector<PackedTriangles> packed_tris;
    std::vector<PackedBoxes> packed_tri_bounds;
    std::vector<Tri> rest_tri;
    std::vector<Box> rest_bounds;
    std::vector<uint32_t> order;  // sorted position -> input position
    uint32_t tri_base = 0;        // global id of this object's first triangle
};

// src/simd_accel.rs:8-66
PackedTriangles pack_triangles(const Tri* t, const std::vector<V3>& verts) {
    alignas(32) float e1[3][8], e2[3][8], v0[3][8];
    for (int i = 0; i < 8; ++i) {
        V3 b = verts[t[i].b], c = verts[t[i].c], a = verts[t[i].a];
        e1[0][i] = b.x; e1[1][i] = b.y; e1[2][i] = b.z;
        e2[0][i] = c.x; e2[1][i] = c.y; e2[2][i] = c.z;
        v0[0][i] = a.x; v0[1][i] = a.y; v0[2][i] = a.z;
    }
    PackedTriangles p;
    for (int k = 0; k < 3; ++k) {
        p.v0[k] = _mm256_load_ps(v0[k]);
        p.e1[k] = _mm256_sub_ps(_mm256_load_ps(e1[k]), p.v0[k]);
        p.e2[k] = _mm256_sub_ps(_mm256_load_ps(e2[k]), p.v0[k]);
    }
    return p;
}

// src/simd_accel.rs:69-108
PackedBoxes pack_boxes(const Box* b) {
    alignas(32) float mn[3][8], mx[3][8];
    for (int i = 0; i < 8; ++i) {
        mn[0][i] = b[i].mn.x; mn[1][i] = b[i].mn.y; mn[2][i] = b[i].mn.z;
        mx[0][i] = b[i].mx.x; mx[1][i] = b[i].mx.y; mx[2][i] = b[i].mx.z;
    }
    PackedBoxes p;
    for (int k = 0; k < 3; ++k) {
        p.mn[k] = _mm256_load_ps(mn[k]);
        p.mx[k] = _mm256_load_ps(mx[k]);
    }
    return p;
}

// src/opt.rs:11-36. Rust's sort_by is stable.
void optimize_model(PackedObject& obj) {
    const size_t n = obj.tris.size();
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    const std::vector<Tri>& tr = obj.tris;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t ia, uint32_t ib) {
        const Tri& a = tr[ia];
        const Tri& b = tr[ib];
        uint32_t amin = std::min(std::min(a.a, a.b), a.c);
        uint32_t bmin = std::min(std::min(b.a, b.b), b.c);
        if (amin == bmin) {
            uint32_t amax = std::max(std::max(a.a, a.b), a.c);
            uint32_t bmax = std::max(std::max(b.a, b.b), b.c);
            return bmax < amax;  // bmax.cmp(&amax) == Less
        }
        return amin < bmin;
    });
    std::vector<Tri> tris(n);
    std::vector<TriData> data(n);
    for (size_t i = 0; i < n; ++i) {
        tris[i] = obj.tris[idx[i]];
        data[i] = obj.tri_data[idx[i]];
    }
    obj.tris.swap(tris);
    obj.tri_data.swap(data);
    obj.order.swap(idx);
}

// src/opt.rs:38-80
void pack_model(PackedObject& obj) {
    std::vector<Box> tri_bounds;
    const size_t n8 = obj.tris.size() / 8;
    for (size_t k = 0; k < n8; ++k) {
        obj.packed_tris.push_back(pack_triangles(&obj.tris[k * 8], obj.verts));
        V3 mn = v3(INF, INF, INF), mx = v3(-INF, -INF, -INF);
        for (size_t i = k * 8; i < (k + 1) * 8; ++i) {
            const Tri& t = obj.tris[i];
            V3 tmn = vmin(vmin(obj.verts[t.a], obj.verts[t.b]), obj.verts[t.c]);
            V3 tmx = vmax(vmax(obj.verts[t.a], obj.verts[t.b]), obj.verts[t.c]);
            mn = vmin(mn, tmn);
            mx = vmax(mx, tmx);
        }
        tri_bounds.push_back(Box{mn, mx});
    }
    const size_t nb8 = tri_bounds.size() / 8;
    for (size_t k = 0; k < nb8; ++k) obj.packed_tri_bounds.push_back(pack_boxes(&tri_bounds[k * 8]));
    obj.rest_tri.assign(obj.tris.begin() + n8 * 8, obj.tris.end());
    obj.rest_bounds.assign(tri_bounds.begin() + nb8 * 8, tri_bounds.end());
}

}  // namespace

struct rto_scene {
    std::vector<PackedObject> objects;
    uint32_t n_tris = 0;
};

namespace {

struct HitRec {
    uint32_t tri = MISS;  // global id
    int obj = -1;
    float t = 0.0f, u = 0.0f, v = 0.0f;
    float closest = INF;
    V3 position = v3(0, 0, 0);
    TriData data{};
};

inline void store8(float* d, __m256 v) { _mm256_storeu_ps(d, v); }  // src/simd_accel.rs:238-245

// src/main.rs:93-171, the nearest-hit search of trace_ray.
void closest_search(const rto_scene& world, const Ray& ray, HitRec& h) {
    float closest = INF;
    __m256 closest_splat = _mm256_set1_ps(closest);
    const SimdRay simd_ray = ray_to_avx(ray);
    int oi = 0;
    for (const PackedObject& obj : world.objects) {
        const int this_obj = oi++;
        if (!box_intersection_check(ray, obj.bounding_box)) continue;
        const size_t nb = obj.packed_tri_bounds.size();
        // One leaf's eight triangles (src/main.rs:111-128 and :136-153 share this body).
        auto scan_leaf = [&](size_t leaf) {
            int m2;
            __m256 uu, vv;
            __m256 tt = packed_tris_intersect(obj.packed_tris[leaf], simd_ray, closest_splat, &m2,
                                              &uu, &vv);
            if (m2 != 0xFF) {
                float t_arr[8], u_arr[8], v_arr[8];
                store8(t_arr, tt);
                store8(u_arr, uu);
                store8(v_arr, vv);
                for (int j = 0; j < 8; ++j) {
                    if (t_arr[j] > 0.0f && closest > t_arr[j]) {
                        closest = t_arr[j];
                        const size_t ti = leaf * 8 + j;
                        h.tri = obj.tri_base + (uint32_t)ti;
                        h.obj = this_obj;
                        h.t = t_arr[j];
                        h.u = u_arr[j];
                        h.v = v_arr[j];
                        h.data = obj.tri_data[ti];
                        h.position = ray.at(t_arr[j]) + (obj.tri_data[ti].normal * 0.00001f);
                    }
                }
                closest_splat = _mm256_set1_ps(closest);
            }
        };
        for (size_t idx = 0; idx < nb; ++idx) {
            int mask;
            __m256 tv = packed_boxes_intersect(obj.packed_tri_bounds[idx], simd_ray, closest_splat,
                                               &mask);
            if (mask != 0xFF) {
                float t_arr[8];
                store8(t_arr, tv);
                for (int i = 0; i < 8; ++i)
                    if (t_arr[i] > 0.0f) scan_leaf(idx * 8 + i);
            }
        }
        for (size_t idx = 0; idx < obj.rest_bounds.size(); ++idx)
            if (box_intersection_check(ray, obj.rest_bounds[idx])) scan_leaf(nb * 8 + idx);
        // src/main.rs:157-169: `closest` is deliberately NOT updated here.
        const size_t c = obj.rest_tri.size();
        for (size_t idx = 0; idx < c; ++idx) {
            const Tri& tri = obj.rest_tri[idx];
            float t, u, v;
            if (triangle_ray_hits(obj.verts[tri.a], obj.verts[tri.b], obj.verts[tri.c], ray, closest,
                                  &t, &u, &v)) {
                if (closest > t) {
                    const size_t ti = obj.tris.size() - c + idx;
                    h.tri = obj.tri_base + (uint32_t)ti;
                    h.obj = this_obj;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                    h.data = obj.tri_data[ti];
                    h.position = ray.at(t) + (obj.tri_data[ti].normal * 0.00001f);
                }
            }
        }
    }
    h.closest = closest;
}

// src/main.rs:185-233, the shadow search. Returns true when something occludes.
bool any_search(const rto_scene& world, const Ray& sun_ray) {
    const float closest = INF;
    const __m256 closest_splat = _mm256_set1_ps(closest);
    const SimdRay simd_ray = ray_to_avx(sun_ray);
    bool can_see_sun = true;
    for (const PackedObject& obj : world.objects) {
        if (!box_intersection_check(sun_ray, obj.bounding_box)) continue;
        const size_t nb = obj.packed_tri_bounds.size();
        for (size_t idx = 0; idx < nb; ++idx) {
            int mask;
            __m256 tv = packed_boxes_intersect(obj.packed_tri_bounds[idx], simd_ray, closest_splat,
                                               &mask);
            if (mask != 0xFF) {
                float t_arr[8];
                store8(t_arr, tv);
                for (int i = 0; i < 8; ++i) {
                    if (t_arr[i] > 0.0f) {
                        int m2;
                        packed_tris_intersect(obj.packed_tris[idx * 8 + i], simd_ray, closest_splat,
                                              &m2, nullptr, nullptr);
                        if (m2 != 0xFF) {
                            can_see_sun = false;
                            break;
                        }
                    }
                }
            }
            if (!can_see_sun) break;
        }
        // The reference keeps scanning rest_bounds even after an occluder was found in the
        // packets; the outcome is the same, so the early exit below is only a shortcut.
        if (can_see_sun) {
            for (size_t idx = 0; idx < obj.rest_bounds.size(); ++idx) {
                if (box_intersection_check(sun_ray, obj.rest_bounds[idx])) {
                    int m2;
                    packed_tris_intersect(obj.packed_tris[nb * 8 + idx], simd_ray, closest_splat, &m2,
                                          nullptr, nullptr);
                    if (m2 != 0xFF) {
                        can_see_sun = false;
                        break;
                    }
                }
            }
        }
        if (can_see_sun) {
            for (const Tri& tri : obj.rest_tri) {
                float t;
                if (triangle_ray_hits(obj.verts[tri.a], obj.verts[tri.b], obj.verts[tri.c], sun_ray,
                                      closest, &t, nullptr, nullptr)) {
                    can_see_sun = false;
                    break;
                }
            }
        }
        if (!can_see_sun) break;
    }
    return !can_see_sun;
}

// ---------------------------------------------------------------------------------------------
// Shading (src/main.rs:74-86, 173-311) and the frame driver (src/main.rs:313-379).
// ---------------------------------------------------------------------------------------------
struct Color {
    float r, g, b;
};
inline Color cmul(Color c, float v) { return Color{c.r * v, c.g * v, c.b * v}; }
inline Color cmul_col(Color c, Color d) { return Color{c.r * d.r, c.g * d.g, c.b * d.b}; }
inline Color cdiv(Color c, float v) { return Color{c.r / v, c.g / v, c.b / v}; }
inline Color cadd(Color c, Color d) { return Color{c.r + d.r, c.g + d.g, c.b + d.b}; }
inline Color cpow(Color c, float e) { return Color{std::pow(c.r, e), std::pow(c.g, e), std::pow(c.b, e)}; }
inline Color color_from_u8(uint8_t r, uint8_t g, uint8_t b) {
    return Color{(float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f};
}
// `as u8` saturates and maps NaN to 0; f32::min(255.0) maps NaN to 255 first.
inline uint8_t to_u8(float c) {
    float v = rmin(c * 255.0f, 255.0f);
    if (!(v > 0.0f)) return 0;
    return (uint8_t)v;
}

inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
// The injected generator standing in for rand::random::<f32>() (src/main.rs:262,264), which is
// OS-seeded and cannot be reproduced. 24 random bits -> [0,1), as rand's Standard f32 does.
inline float counter_rand(uint32_t seed, uint64_t sample_index, uint32_t scatter, uint32_t which) {
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h ^ (uint32_t)sample_index);
    h = mix32(h ^ (uint32_t)(sample_index >> 32));
    h = mix32(h ^ (scatter * 2u + which));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Sine and cosine of the scatter angle psi = rand * 2 * PI (src/main.rs:264-268 take them from libm).
// Like the random draw, they are injected: one fixed sequence of IEEE operations that the kernels
// under test repeat bit for bit, because libm differs by an ulp from platform to platform and a
// direction that differs in its last bit can meet another triangle far away. Quadrant by rounding
// psi * 2/PI, Cody-Waite reduction with a two-part PI/2, the Cephes single-precision polynomials
// on [-PI/4, PI/4] (below 2 ulp; checked against double-precision sin / cos in tests/).
inline void injected_sincos(float psi, float* sin_out, float* cos_out) {
    const int k = (int)(psi * 0.636619772367581343f + 0.5f);
    const float kf = (float)k;
    float r = std::fmaf(-kf, 1.5707962512969970703125f, psi);
    r = std::fmaf(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = std::fmaf(ps, z, -1.6666654611e-1f);
    const float s = std::fmaf(ps * z, r, r);
    float pc = std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = std::fmaf(pc, z, 4.166664568298827e-2f);
    const float c = std::fmaf(pc * z, z, std::fmaf(-0.5f, z, 1.0f));
    const int q = k & 3;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sin_out = (q & 2) ? -ss : ss;
    *cos_out = (q == 1 || q == 2) ? -cc : cc;
}

// src/main.rs:74-86
inline void get_tangent_vectors(V3 norm, V3* t1, V3* t2) {
    V3 arbitrary = (std::fabs(norm.z) < 1e-6f) ? v3(0, 0, 1) : v3(1, 0, 0);
    *t1 = normalize(cross(norm, arbitrary));
    *t2 = normalize(cross(norm, *t1));
}

struct TraceCtx {
    const rto_scene* world;
    const rto_render_params* p;
    V3 sun;
    uint64_t sample_index;
    uint64_t rays;        // nearest-hit + shadow searches performed
    uint32_t primary_hit; // hit id of the depth-1 search
};

// src/main.rs:88-311
Color trace_ray(const Ray& ray, TraceCtx& cx, uint32_t depth, float* dist) {
    const rto_scene& world = *cx.world;
    HitRec h;
    closest_search(world, ray, h);
    cx.rays += 1;
    if (depth == 1) cx.primary_hit = h.tri;
    if (h.tri == MISS) {
        *dist = INF;
        return color_from_u8(0x87, 0xce, 0xeb);
    }
    Color color{0.0f, 0.0f, 0.0f};
    const Material& mat = world.objects[h.obj].material;
    const V3 hit_pos = h.position;
    const V3 sun_dir = normalize(cx.sun - hit_pos);
    const Ray sun_ray = make_ray(hit_pos, sun_dir);
    if (dot(h.data.normal, sun_dir) > 0.0f) {
        cx.rays += 1;
        if (!any_search(world, sun_ray)) {
            float light_power = dot(h.data.normal, sun_dir);
            color = cmul(Color{h.data.color.x, h.data.color.y, h.data.color.z},
                         light_power * mat.albedo);
        }
    }
    const Color hit_color{h.data.color.x, h.data.color.y, h.data.color.z};
    Color reflect_color{0.0f, 0.0f, 0.0f};
    if (depth < cx.p->reflect_depth && mat.reflectivity > 0.0f) {
        V3 reflection_dir = ray.dir - 2.0f * h.data.normal * dot(ray.dir, h.data.normal);
        float d;
        Color col = trace_ray(make_ray(hit_pos, reflection_dir), cx, depth + 1, &d);
        reflect_color = cmul(col, mat.reflectivity);
    }
    Color diffuse_color{0.0f, 0.0f, 0.0f};
    if (depth < cx.p->diffuse_depth) {
        const uint32_t n = cx.p->diffuse_scatt;
        for (uint32_t k = 0; k < n; ++k) {
            float sin_theta = std::sqrt(counter_rand(cx.p->seed, cx.sample_index, k, 0));
            float cos_theta = std::sqrt(1.0f - sin_theta * sin_theta);
            float psi = counter_rand(cx.p->seed, cx.sample_index, k, 1) * 2.0f * 3.14159265358979323846f;
            float sin_psi, cos_psi;
            injected_sincos(psi, &sin_psi, &cos_psi);
            float a = sin_theta * cos_psi;
            float b = sin_theta * sin_psi;
            float c = cos_theta;
            V3 tang1, tang2;
            get_tangent_vectors(h.data.normal, &tang1, &tang2);
            V3 vel = normalize(v3(a * tang1.x + b * tang2.x + c * h.data.normal.x,
                                  a * tang1.y + b * tang2.y + c * h.data.normal.y,
                                  a * tang1.z + b * tang2.z + c * h.data.normal.z));
            float light_power = dot(h.data.normal, vel);
            float d;
            Color col = trace_ray(make_ray(hit_pos, vel), cx, depth + 1, &d);
            diffuse_color = cadd(diffuse_color,
                                 cmul(cmul_col(col, hit_color), light_power * mat.roughness));
        }
        diffuse_color = cdiv(diffuse_color, (float)n);
    }
    color = cadd(color, reflect_color);
    color = cadd(color, diffuse_color);
    *dist = h.closest;
    if (depth == 1) return cpow(color, 1.0f / 2.2f);
    return color;
}

// Sub-sample offsets. grid 2 is the reference's fixed order (src/main.rs:322-364):
// (.25,.25) (.75,.75) (.25,.75) (.75,.25); other grids go row-major over (i+0.5)/n.
inline void sample_offset(uint32_t grid, uint32_t s, float* ox, float* oy) {
    if (grid == 2) {
        static const float X[4] = {0.25f, 0.75f, 0.25f, 0.75f};
        static const float Y[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        *ox = X[s];
        *oy = Y[s];
    } else {
        *ox = ((float)(s % grid) + 0.5f) / (float)grid;
        *oy = ((float)(s / grid) + 0.5f) / (float)grid;
    }
}

// src/main.rs:322-330. aspect_ratio is an integer division there (src/main.rs:314).
inline Ray camera_ray(uint32_t W, uint32_t H, uint32_t x, uint32_t y, float ox, float oy) {
    const float SCALE = 1.0f;
    const uint32_t aspect_ratio = W / H;
    float xx = (2.0f * ((float)x + ox) / (float)W - 1.0f) * (float)aspect_ratio * SCALE;
    float yy = (1.0f - 2.0f * ((float)y + oy) / (float)H) * SCALE;
    V3 d = v3(xx, yy, 1.0f);
    return make_ray(v3(0, 0, 0), d);
}

template <class F>
void parallel_for(uint64_t n, int n_threads, F f) {
    if (n_threads <= 1 || n < 2) {
        f(0, n, 0);
        return;
    }
    // Dynamic chunks, in the spirit of rayon's work stealing over rows (src/main.rs:317-319).
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = std::max<uint64_t>(1, n / ((uint64_t)n_threads * 32));
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; ++w) {
        th.emplace_back([&, w]() {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk), w);
            }
        });
    }
    for (auto& t : th) t.join();
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C interface
// ---------------------------------------------------------------------------------------------
extern "C" {

rto_scene* rto_scene_create(const rto_object_desc* objs, uint32_t n_objs) {
    rto_scene* s = new rto_scene();
    uint32_t base = 0;
    for (uint32_t o = 0; o < n_objs; ++o) {
        const rto_object_desc& d = objs[o];
        PackedObject po;
        po.verts.resize(d.n_verts);
        for (uint32_t i = 0; i < d.n_verts; ++i)
            po.verts[i] = v3(d.verts[3 * i], d.verts[3 * i + 1], d.verts[3 * i + 2]);
        po.tris.resize(d.n_tris);
        po.tri_data.resize(d.n_tris);
        for (uint32_t i = 0; i < d.n_tris; ++i) {
            po.tris[i] = Tri{d.tris[3 * i], d.tris[3 * i + 1], d.tris[3 * i + 2]};
            const float* td = d.tri_data + 6 * (size_t)i;
            po.tri_data[i] = TriData{v3(td[0], td[1], td[2]), v3(td[3], td[4], td[5])};
        }
        // Object::from, src/objects.rs:42-61
        V3 mn = po.verts[0], mx = po.verts[0];
        for (const V3& t : po.verts) {
            mn = vmin(vmin(vmin(mn, t), t), t);
            mx = vmax(vmax(vmax(mx, t), t), t);
        }
        po.bounding_box = Box{mn, mx};
        po.material = Material{d.albedo_over_pi, d.reflectivity, d.roughness};
        optimize_model(po);
        pack_model(po);
        po.tri_base = base;
        base += d.n_tris;
        s->objects.push_back(std::move(po));
    }
    s->n_tris = base;
    return s;
}

void rto_scene_destroy(rto_scene* s) { delete s; }
uint32_t rto_scene_num_tris(const rto_scene* s) { return s->n_tris; }

int rto_scene_order(const rto_scene* s, uint32_t obj, uint32_t* order) {
    if (obj >= s->objects.size()) return -1;
    const auto& o = s->objects[obj].order;
    std::memcpy(order, o.data(), o.size() * sizeof(uint32_t));
    return 0;
}
uint32_t rto_scene_num_packed_tris(const rto_scene* s, uint32_t obj) {
    return (uint32_t)s->objects[obj].packed_tris.size();
}
uint32_t rto_scene_num_packed_bounds(const rto_scene* s, uint32_t obj) {
    return (uint32_t)s->objects[obj].packed_tri_bounds.size();
}
uint32_t rto_scene_num_rest_bounds(const rto_scene* s, uint32_t obj) {
    return (uint32_t)s->objects[obj].rest_bounds.size();
}
uint32_t rto_scene_num_rest_tri(const rto_scene* s, uint32_t obj) {
    return (uint32_t)s->objects[obj].rest_tri.size();
}
int rto_scene_packed_tris(const rto_scene* s, uint32_t obj, float* out) {
    const auto& v = s->objects[obj].packed_tris;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(PackedTriangles));
    return 0;
}
int rto_scene_packed_bounds(const rto_scene* s, uint32_t obj, float* out) {
    const auto& v = s->objects[obj].packed_tri_bounds;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(PackedBoxes));
    return 0;
}
int rto_scene_rest_bounds(const rto_scene* s, uint32_t obj, float* out) {
    const auto& v = s->objects[obj].rest_bounds;
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(Box));
    return 0;
}
int rto_scene_object_bounds(const rto_scene* s, uint32_t obj, float* out) {
    std::memcpy(out, &s->objects[obj].bounding_box, sizeof(Box));
    return 0;
}

int rto_trace_closest(const rto_scene* s, const rto_ray* rays, uint64_t n, rto_hit* out,
                      int n_threads) {
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e, int) {
        for (uint64_t i = b; i < e; ++i) {
            Ray r = make_ray(v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]),
                             v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]));
            HitRec h;
            closest_search(*s, r, h);
            out[i].tri = h.tri;
            out[i].t = h.t;
            out[i].u = h.u;
            out[i].v = h.v;
            out[i].closest = h.closest;
        }
    });
    return 0;
}

int rto_trace_any(const rto_scene* s, const rto_ray* rays, uint64_t n, uint8_t* occluded,
                  int n_threads) {
    parallel_for(n, n_threads, [&](uint64_t b, uint64_t e, int) {
        for (uint64_t i = b; i < e; ++i) {
            Ray r = make_ray(v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]),
                             v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]));
            occluded[i] = any_search(*s, r) ? 1 : 0;
        }
    });
    return 0;
}

int rto_render(const rto_scene* s, const rto_render_params* p, uint8_t* rgba8, uint32_t* hit_ids,
               uint64_t* ray_count, int n_threads) {
    const uint32_t W = p->width, H = p->height, G = p->grid;
    if (W == 0 || H == 0 || G == 0) return -1;
    const uint32_t spp = G * G;
    const uint32_t stride = p->pixel_stride ? p->pixel_stride : 1;
    std::vector<uint64_t> counts((size_t)std::max(1, n_threads), 0);
    parallel_for(H, n_threads, [&](uint64_t yb, uint64_t ye, int w) {
        uint64_t local = 0;
        for (uint32_t y = (uint32_t)yb; y < (uint32_t)ye; ++y) {
            for (uint32_t x = 0; x < W; ++x) {
                const uint64_t pix = (uint64_t)y * W + x;
                if (pix % stride != 0) continue;
                Color acc{0, 0, 0};
                for (uint32_t sidx = 0; sidx < spp; ++sidx) {
                    float ox, oy;
                    sample_offset(G, sidx, &ox, &oy);
                    Ray ray = camera_ray(W, H, x, y, ox, oy);
                    TraceCtx cx{s, p, v3(p->sun[0], p->sun[1], p->sun[2]), pix * spp + sidx, 0, MISS};
                    float d;
                    Color c = trace_ray(ray, cx, 1, &d);
                    acc = (sidx == 0) ? c : cadd(acc, c);
                    local += cx.rays;
                    if (hit_ids) hit_ids[pix * spp + sidx] = cx.primary_hit;
                }
                acc = cdiv(acc, (float)spp);
                rgba8[pix * 4 + 0] = to_u8(acc.r);
                rgba8[pix * 4 + 1] = to_u8(acc.g);
                rgba8[pix * 4 + 2] = to_u8(acc.b);
                rgba8[pix * 4 + 3] = 0xff;
            }
        }
        counts[(size_t)w] += local;
    });
    if (ray_count) {
        uint64_t tot = 0;
        for (uint64_t c : counts) tot += c;
        *ray_count = tot;
    }
    return 0;
}

int rto_packed_boxes_intersect(const float* boxes48, const rto_ray* ray, float ray_length,
                               float* t_out8) {
    PackedBoxes p;
    for (int k = 0; k < 3; ++k) {
        p.mn[k] = _mm256_loadu_ps(boxes48 + 8 * k);
        p.mx[k] = _mm256_loadu_ps(boxes48 + 24 + 8 * k);
    }
    Ray r = make_ray(v3(ray->o[0], ray->o[1], ray->o[2]), v3(ray->d[0], ray->d[1], ray->d[2]));
    int mask;
    __m256 t = packed_boxes_intersect(p, ray_to_avx(r), _mm256_set1_ps(ray_length), &mask);
    _mm256_storeu_ps(t_out8, t);
    return mask;
}

int rto_packed_tris_intersect(const float* tris72, const rto_ray* ray, float ray_length,
                              float* t_out8, float* u_out8, float* v_out8) {
    PackedTriangles p;
    for (int k = 0; k < 3; ++k) {
        p.e1[k] = _mm256_loadu_ps(tris72 + 8 * k);
        p.e2[k] = _mm256_loadu_ps(tris72 + 24 + 8 * k);
        p.v0[k] = _mm256_loadu_ps(tris72 + 48 + 8 * k);
    }
    Ray r = make_ray(v3(ray->o[0], ray->o[1], ray->o[2]), v3(ray->d[0], ray->d[1], ray->d[2]));
    int mask;
    __m256 u, v;
    __m256 t = packed_tris_intersect(p, ray_to_avx(r), _mm256_set1_ps(ray_length), &mask, &u, &v);
    _mm256_storeu_ps(t_out8, t);
    if (u_out8) _mm256_storeu_ps(u_out8, u);
    if (v_out8) _mm256_storeu_ps(v_out8, v);
    return mask;
}

int rto_box_intersection_check(const float o[3], const float inv_dir[3], const float bmin[3],
                               const float bmax[3]) {
    Ray r{v3(o[0], o[1], o[2]), v3(0, 0, 0), v3(inv_dir[0], inv_dir[1], inv_dir[2])};
    Box b{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    return box_intersection_check(r, b) ? 1 : 0;
}

int rto_triangle_ray_hits(const float a[3], const float b[3], const float c[3], const rto_ray* ray,
                          float min, float* t, float* u, float* v) {
    Ray r = make_ray(v3(ray->o[0], ray->o[1], ray->o[2]), v3(ray->d[0], ray->d[1], ray->d[2]));
    return triangle_ray_hits(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]), v3(c[0], c[1], c[2]), r,
                             min, t, u, v)
               ? 1
               : 0;
}

void rto_calculate_normal(const float a[3], const float b[3], const float c[3], float out[3]) {
    V3 n = normalize(cross(v3(b[0], b[1], b[2]) - v3(a[0], a[1], a[2]),
                           v3(c[0], c[1], c[2]) - v3(a[0], a[1], a[2])));
    out[0] = n.x;
    out[1] = n.y;
    out[2] = n.z;
}

void rto_sincos(float psi, float* sin_out, float* cos_out) { injected_sincos(psi, sin_out, cos_out); }

float rto_rand(uint32_t seed, uint64_t sample_index, uint32_t scatter, uint32_t which) {
    return counter_rand(seed, sample_index, scatter, which);
}

}  // extern "C"
