// rt_exact.h -- the reference's arithmetic, operation for operation, for one ray per lane.
//
// Everything here must round exactly as the reference's x86 code does, so this file is compiled
// with -ffp-contract=off and spells every fused multiply-add as an explicit __builtin_fmaf.
// IEEE division and square root are hipcc's default for f32 (correctly rounded), and gfx950
// kernels keep f32 subnormals, as x86 does.
//
// Paths cited are relative to the reference root.
#ifndef RT_EXACT_H
#define RT_EXACT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_F32_EPS 1.1920929e-7f  // f32::EPSILON
#define RT_INF __builtin_huge_valf()

struct rt_f3 {
    float x, y, z;
};

__device__ __forceinline__ rt_f3 f3(float x, float y, float z) { return rt_f3{x, y, z}; }
__device__ __forceinline__ rt_f3 operator+(rt_f3 a, rt_f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ rt_f3 operator-(rt_f3 a, rt_f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ rt_f3 operator*(rt_f3 a, rt_f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ rt_f3 operator*(rt_f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ rt_f3 operator*(float s, rt_f3 a) { return f3(s * a.x, s * a.y, s * a.z); }

// ---- glam 0.27 scalar Vec3 (separately rounded operations) ---------------------------------
// f32::min / f32::max: the non-NaN operand wins.
__device__ __forceinline__ float g_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
__device__ __forceinline__ float g_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
__device__ __forceinline__ float g_dot(rt_f3 a, rt_f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ __forceinline__ rt_f3 g_cross(rt_f3 a, rt_f3 b) {
    return f3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ rt_f3 g_recip(rt_f3 a) { return f3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
__device__ __forceinline__ rt_f3 g_normalize(rt_f3 a) { return a * (1.0f / __builtin_sqrtf(g_dot(a, a))); }

// box_intersection_check, src/objects.rs:322-333.
__device__ __forceinline__ bool box_intersection_check(rt_f3 o, rt_f3 inv, rt_f3 bmin, rt_f3 bmax) {
    rt_f3 t1 = (bmin - o) * inv;
    rt_f3 t2 = (bmax - o) * inv;
    float nx = g_min(t1.x, t2.x), ny = g_min(t1.y, t2.y), nz = g_min(t1.z, t2.z);
    float fx = g_max(t1.x, t2.x), fy = g_max(t1.y, t2.y), fz = g_max(t1.z, t2.z);
    float t_near = g_max(g_max(nx, ny), nz);
    float t_far = g_min(g_min(fx, fy), fz);
    return g_min(t_near, 0.0f) <= t_far;
}

// Triangle::ray_hits, src/objects.rs:273-308 (the path for an object's trailing n % 8 triangles).
__device__ __forceinline__ bool triangle_ray_hits(rt_f3 va, rt_f3 vb, rt_f3 vc, rt_f3 o, rt_f3 d,
                                                  float min, float* t_out, float* u_out, float* v_out) {
    rt_f3 e1 = vb - va;
    rt_f3 e2 = vc - va;
    rt_f3 ray_cross_e2 = g_cross(d, e2);
    float det = g_dot(e1, ray_cross_e2);
    if (det > -RT_F32_EPS && det < RT_F32_EPS) return false;
    float inv_det = 1.0f / det;
    rt_f3 s = o - va;
    float u = inv_det * g_dot(s, ray_cross_e2);
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    rt_f3 s_cross_e1 = g_cross(s, e1);
    float v = inv_det * g_dot(d, s_cross_e1);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * g_dot(e2, s_cross_e1);
    if (t > RT_F32_EPS && t < min) {
        *t_out = t;
        *u_out = u;
        *v_out = v;
        return true;
    }
    return false;
}

// ---- sine and cosine of the scatter angle -----------------------------------------------------
// The reference takes psi.cos() and psi.sin() of psi = rand * 2 * PI from libm (src/main.rs:264-268).
// libm results differ by an ulp between hosts and devices, and a scatter direction that differs in
// its last bit can meet another triangle far away. The random draw itself is injected (the
// reference's generator is OS-seeded), so the angle's sine and cosine are defined here instead, as
// one fixed sequence of IEEE operations that the CPU checker repeats bit for bit: quadrant by
// rounding psi * 2/PI, Cody-Waite reduction with a two-part PI/2, the Cephes single-precision
// polynomials on [-PI/4, PI/4] (error below 2 ulp of the result), every multiply-add explicit.
// Valid for 0 <= psi <= 2 PI + a little.
__device__ __forceinline__ void rt_sincos(float psi, float* sin_out, float* cos_out) {
    const int k = (int)(psi * 0.636619772367581343f + 0.5f);
    const float kf = (float)k;
    float r = __builtin_fmaf(-kf, 1.5707962512969970703125f, psi);   // PI/2, upper 24 bits
    r = __builtin_fmaf(-kf, 7.54978995489188216e-8f, r);             // PI/2, the rest
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float s = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = k & 3;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sin_out = (q & 2) ? -ss : ss;
    *cos_out = (q == 1 || q == 2) ? -cc : cc;
}

// ---- the 8-wide kernels, one lane of them ----------------------------------------------------
// avx_multi_cross, src/simd_accel.rs:136-142: fmsub(a1,b2, b1*a2) etc.
__device__ __forceinline__ rt_f3 s_cross(rt_f3 a, rt_f3 b) {
    return f3(__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)),
              __builtin_fmaf(a.x, b.y, -(b.x * a.y)));
}
// avx_multi_dot, src/simd_accel.rs:145-153: fmadd(a2,b2, fmadd(a1,b1, a0*b0)).
__device__ __forceinline__ float s_dot(rt_f3 a, rt_f3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}

struct rt_tri_eval {
    float t, u, v;
    bool failed;  // the `failed` mask of the reference, without the t > ray_length term
};

// PackedTriangles::intersect, src/simd_accel.rs:281-331, one lane. The t > ray_length term of the
// failure mask is left to the caller: the nearest-hit consumer applies the stricter
// `closest > t` anyway (src/main.rs:117) and the shadow consumer passes +inf (src/main.rs:185).
__device__ __forceinline__ rt_tri_eval tri_intersect(rt_f3 e1, rt_f3 e2, rt_f3 v0, rt_f3 o, rt_f3 d) {
    rt_f3 q = s_cross(d, e2);
    float a = s_dot(e1, q);
    float f = 1.0f / a;
    rt_f3 s = o - v0;
    float u = f * s_dot(s, q);
    rt_f3 r = s_cross(s, e1);
    float v = f * s_dot(d, r);
    float t = f * s_dot(e2, r);
    bool failed = (a > -RT_F32_EPS) && (a < RT_F32_EPS);
    failed = failed || (u < 0.0f);
    failed = failed || (v < 0.0f);
    failed = failed || ((u + v) > 1.0f);
    failed = failed || (t < 0.0f);
    return rt_tri_eval{t, u, v, failed};
}

// The same test in two steps, for callers that can stop after the first: tri_begin computes what
// the u coordinate needs, tri_finish the rest; together they perform exactly tri_intersect's
// operations in tri_intersect's order.
struct rt_tri_head {
    rt_f3 q, s;
    float a, un;   // a = e1 . (d x e2), un = s . (d x e2)  (u = (1 / a) * un)
};
__device__ __forceinline__ rt_tri_head tri_begin(rt_f3 e1, rt_f3 e2, rt_f3 v0, rt_f3 o, rt_f3 d) {
    rt_tri_head h;
    h.q = s_cross(d, e2);
    h.a = s_dot(e1, h.q);
    h.s = o - v0;
    h.un = s_dot(h.s, h.q);
    return h;
}
// True only when tri_finish would report `failed` (or a NaN t) for this lane, decided from the
// head alone: |a| below the reference's epsilon (its own first term); un and a of opposite sign,
// which makes u = fl(fl(1 / a) * un) negative unless the product underflows (excluded by the
// magnitude guard); |un| above |a| by more than the two roundings of u can undo, which makes
// u > 1 and, v being >= 0 or the test failed anyway, u + v > 1. NaN operands reject nothing.
// Valid when no intermediate can overflow (coordinates far below 1e18): then v and t are not NaN
// unless a or un is.
__device__ __forceinline__ bool tri_head_fails(const rt_tri_head& h) {
    const float aa = __builtin_fabsf(h.a), au = __builtin_fabsf(h.un);
    const bool tiny = (h.a > -RT_F32_EPS) && (h.a < RT_F32_EPS);
    const bool neg = (h.un * h.a < 0.0f) && (au > 1e-30f * aa);
    const bool big = au > 1.000001f * aa;
    return tiny || neg || big;
}
__device__ __forceinline__ rt_tri_eval tri_finish(const rt_tri_head& h, rt_f3 e1, rt_f3 e2, rt_f3 d) {
    float f = 1.0f / h.a;
    float u = f * h.un;
    rt_f3 r = s_cross(h.s, e1);
    float v = f * s_dot(d, r);
    float t = f * s_dot(e2, r);
    bool failed = (h.a > -RT_F32_EPS) && (h.a < RT_F32_EPS);
    failed = failed || (u < 0.0f);
    failed = failed || (v < 0.0f);
    failed = failed || ((u + v) > 1.0f);
    failed = failed || (t < 0.0f);
    return rt_tri_eval{t, u, v, failed};
}

// _mm256_max_ps / _mm256_min_ps: the SECOND operand is returned when either is NaN.
__device__ __forceinline__ float x86_max(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float x86_min(float a, float b) { return (a < b) ? a : b; }
// f32::signum() == 1.0 (src/simd_accel.rs:268-270): false for -0.0, negatives and NaN.
__device__ __forceinline__ bool signum_is_one(float x) { return !(x != x) && !__builtin_signbit(x); }

// PackedBoxes::intersect, src/simd_accel.rs:342-377, one lane: true when the lane's result is
// +1.0 (hit), false when it is -1.0.
__device__ __forceinline__ bool packed_box_hit(rt_f3 bmin, rt_f3 bmax, rt_f3 o, rt_f3 inv, float ray_length) {
    rt_f3 a = (bmin - o) * inv;
    rt_f3 b = (bmax - o) * inv;
    bool kx = signum_is_one(inv.x), ky = signum_is_one(inv.y), kz = signum_is_one(inv.z);
    float t1x = kx ? a.x : b.x, t2x = kx ? b.x : a.x;
    float t1y = ky ? a.y : b.y, t2y = ky ? b.y : a.y;
    float t1z = kz ? a.z : b.z, t2z = kz ? b.z : a.z;
    float tmin = x86_max(x86_max(t1x, t1y), t1z);
    float tmax = x86_min(x86_min(x86_min(t2x, t2y), t2z), ray_length);
    return tmin <= tmax;
}

#endif
