/*
 * rays_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY. This library restates, on the CPU, the arithmetic and the
 * traversal order of the reference ray tracer's trace_ray path (linear packet scan, AVX2/FMA
 * intrinsics). It exists so that tests and the benchmark's cpu_baseline leg can check and time
 * the HIP path against it. Nothing in the shipped library links, loads or calls it.
 *
 * It is a restatement in C++, not the reference's Rust binary (no Rust toolchain in the build
 * image). The 8-wide paths use the same x86 instructions through <immintrin.h>, so they are
 * bit-exact with the reference by construction. The scalar paths follow glam 0.27.0's scalar
 * Vec3 definitions as recalled from the published crate; glam's source is not available offline,
 * so parity at that boundary is pinned only by the reference's single box-test known answer
 * (src/objects.rs:339-353): "parity unpinned at the glam boundary".
 */
#ifndef RAYS_ORACLE_H
#define RAYS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rto_scene rto_scene;

/* One object as the reference's Object::from receives it (src/objects.rs:42-61). */
typedef struct {
    const float*    verts;      /* n_verts * 3, as Object.verts                       */
    uint32_t        n_verts;
    const uint32_t* tris;       /* n_tris * 3 vertex indices, BEFORE optimize_model   */
    uint32_t        n_tris;
    const float*    tri_data;   /* n_tris * 6: normal xyz, color rgb                  */
    float           albedo_over_pi, reflectivity, roughness; /* Material after ::new  */
} rto_object_desc;

typedef struct { float o[3], d[3]; } rto_ray;          /* inv_dir = 1/d componentwise  */

typedef struct {
    uint32_t tri;      /* global triangle id in post-optimize_model order; 0xFFFFFFFF = miss */
    float    t;        /* t of the accepted triangle                                       */
    float    u, v;     /* barycentrics (computed and discarded by the reference)           */
    float    closest;  /* the distance trace_ray returns (stale for rest_tri hits, Q7)      */
} rto_hit;

typedef struct {
    uint32_t width, height;
    uint32_t grid;            /* n: n x n sub-samples per pixel; 2 = the reference pattern */
    uint32_t reflect_depth;   /* REFLECT_DEPTH, reference 4 (src/main.rs:34)                */
    uint32_t diffuse_depth;   /* DIFFUSE_DEPTH, reference 2 (src/main.rs:35)                */
    uint32_t diffuse_scatt;   /* DIFFUSE_SCATT, reference 10 (src/main.rs:36)               */
    float    sun[3];
    uint32_t seed;            /* key of the injected counter RNG                            */
    uint32_t pixel_stride;    /* 1 = every pixel; k = only pixels whose index % k == 0      */
} rto_render_params;

/* Scene: applies optimize_model then pack_model to every object (src/opt.rs:11-80). */
rto_scene* rto_scene_create(const rto_object_desc* objs, uint32_t n_objs);
void       rto_scene_destroy(rto_scene*);
uint32_t   rto_scene_num_tris(const rto_scene*);
/* order[i] = index, within the object's input list, of the triangle at sorted position i. */
int        rto_scene_order(const rto_scene*, uint32_t obj, uint32_t* order);
/* Copies of the packed arrays, for tests of pack_triangles / pack_boxes / pack_model. */
uint32_t   rto_scene_num_packed_tris(const rto_scene*, uint32_t obj);
uint32_t   rto_scene_num_packed_bounds(const rto_scene*, uint32_t obj);
uint32_t   rto_scene_num_rest_bounds(const rto_scene*, uint32_t obj);
uint32_t   rto_scene_num_rest_tri(const rto_scene*, uint32_t obj);
int        rto_scene_packed_tris(const rto_scene*, uint32_t obj, float* out /* n*72 */);
int        rto_scene_packed_bounds(const rto_scene*, uint32_t obj, float* out /* n*48 */);
int        rto_scene_rest_bounds(const rto_scene*, uint32_t obj, float* out /* n*6 */);
int        rto_scene_object_bounds(const rto_scene*, uint32_t obj, float* out /* 6 */);

/* trace_ray's nearest-hit search (src/main.rs:88-171) over a batch, n_threads workers. */
int rto_trace_closest(const rto_scene*, const rto_ray* rays, uint64_t n, rto_hit* out,
                      int n_threads);
/* trace_ray's shadow search (src/main.rs:185-233): occluded[i] = !can_see_sun. */
int rto_trace_any(const rto_scene*, const rto_ray* rays, uint64_t n, uint8_t* occluded,
                  int n_threads);
/* draw (src/main.rs:313-379): RGBA8 frame; hit_ids (nullable) = primary hit per sample,
 * laid out [pixel][sample]; ray_count (nullable) = rays cast (nearest + shadow searches). */
int rto_render(const rto_scene*, const rto_render_params*, uint8_t* rgba8, uint32_t* hit_ids,
               uint64_t* ray_count, int n_threads);

/* Unit-level entry points, one per reference routine. */
/* PackedBoxes::intersect (src/simd_accel.rs:342-377): boxes = min.x[8] .. max.z[8]. */
int  rto_packed_boxes_intersect(const float* boxes48, const rto_ray* ray, float ray_length,
                                float* t_out8);
/* PackedTriangles::intersect (src/simd_accel.rs:281-331): tris = e1.x[8] .. v0.z[8];
 * u_out8 / v_out8 nullable. */
int  rto_packed_tris_intersect(const float* tris72, const rto_ray* ray, float ray_length,
                               float* t_out8, float* u_out8, float* v_out8);
/* box_intersection_check (src/objects.rs:322-333) with an explicit inv_dir. */
int  rto_box_intersection_check(const float o[3], const float inv_dir[3], const float bmin[3],
                                const float bmax[3]);
/* Triangle::ray_hits (src/objects.rs:273-308): returns 1 and *t on Some(t). */
int  rto_triangle_ray_hits(const float a[3], const float b[3], const float c[3],
                           const rto_ray* ray, float min, float* t, float* u, float* v);
/* calculate_normal (src/objects.rs:5-7). */
void rto_calculate_normal(const float a[3], const float b[3], const float c[3], float out[3]);
/* The injected generator: uniform f32 in [0,1) for (seed, sample index, scatter, which). */
/* The injected sine / cosine of the scatter angle (0 <= psi <= 2 PI), see rays_oracle.cpp. */
void rto_sincos(float psi, float* sin_out, float* cos_out);
float rto_rand(uint32_t seed, uint64_t sample_index, uint32_t scatter, uint32_t which);

#ifdef __cplusplus
}
#endif
#endif
