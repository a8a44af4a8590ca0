/*
 * rays_hip.h -- C ABI of the MI355X (gfx950) implementation of the reference ray tracer's
 * trace_ray hot path: nearest-hit and any-hit search over packed triangles, plus the frame
 * driver around it.
 *
 * The reference (a Rust crate) has no FFI; the boundary below replaces its Rust function API.
 * Each entry point names the reference interface it stands in for (paths relative to the
 * reference root). Per-ray calls become batch calls; the host keeps the Object / World fields
 * and hands over flat arrays.
 *
 * Conventions
 *  - Every function returns 0 on success and a negative rt_status otherwise; nothing throws or
 *    aborts across the boundary. rt_last_error() returns a thread-local message for the last
 *    failure on the calling thread.
 *  - The caller owns every input and output buffer. rt_scene_build copies what it needs; the
 *    library owns device memory until rt_scene_destroy / rt_ctx_destroy. A context destroyed while
 *    scenes built on it are alive stays valid for them and is freed with the last of them.
 *  - A built scene is immutable. Calls on one rt_ctx are serialised internally.
 *  - Hit id = (sum of the triangle counts of the preceding objects) + index of the triangle in
 *    the object's post-optimize_model order. This is the index the reference uses into
 *    Object.tri_data (src/main.rs:119,144,161). 0xFFFFFFFF means miss.
 *  - Functions with the suffix _device take device pointers (memory of the context's GPU) and a
 *    hipStream_t passed as void*; they enqueue work and return without synchronising.
 */
#ifndef RAYS_HIP_H
#define RAYS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MISS 0xFFFFFFFFu

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, index out of range ...) */
    RT_ERR_DEVICE = -2,    /* a HIP call failed                                               */
    RT_ERR_NOMEM = -3,     /* host or device allocation failed                                */
    RT_ERR_UNSUPPORTED = -4
} rt_status;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

/* One object, as Object::from receives it (src/objects.rs:25-31,42-61). */
typedef struct {
    const float*    verts;     /* n_verts * 3 floats, Object.verts                           */
    uint32_t        n_verts;
    const uint32_t* tris;      /* n_tris * 3 vertex indices, Object.tris BEFORE optimize_model */
    uint32_t        n_tris;
    const float*    tri_data;  /* n_tris * 6 floats: TriangleData.normal xyz, .color rgb     */
    float           albedo_over_pi;  /* Material.albedo as Material::new stores it
                                        (src/objects.rs:183-190: albedo / PI)                */
    float           reflectivity;
    float           roughness;
} rt_object_desc;

/* Ray (src/objects.rs:64-68). inv_dir is derived as 1.0f / d, as every caller in the reference
 * does (dir.recip()). */
typedef struct { float o[3], d[3]; } rt_ray;

/* Result of one nearest-hit search. u and v are the barycentrics the reference computes and
 * drops (src/simd_accel.rs:298,303). */
typedef struct { uint32_t tri; float t, u, v; } rt_hit;

/* Frame description: the constants of src/main.rs:29-36 made parameters. */
typedef struct {
    uint32_t width, height;   /* WIDTH, HEIGHT                                               */
    uint32_t grid;            /* n: n x n sub-samples per pixel. 2 reproduces the reference's
                                 four fixed samples in its order (src/main.rs:322-364)       */
    uint32_t reflect_depth;   /* REFLECT_DEPTH (reference: 4)                                */
    uint32_t diffuse_depth;   /* DIFFUSE_DEPTH (reference: 2)                                */
    uint32_t diffuse_scatt;   /* DIFFUSE_SCATT (reference: 10)                               */
    float    sun[3];          /* World.sun                                                   */
    uint32_t seed;            /* key of the counter-based generator that replaces
                                 rand::random (src/main.rs:262,264)                          */
    uint32_t shard_index;     /* this GPU's rank among shard_count tile owners               */
    uint32_t shard_count;     /* 1 = render the whole frame                                  */
} rt_render_params;

/* Counters of one rt_render_device_stats call: whole frame, then the camera-ray kernel alone
 * (equal to the whole frame when the frame runs as one kernel). */
typedef struct {
    uint64_t rays;        /* nearest-hit + any-hit searches                                  */
    uint64_t box_tests;   /* ray / AABB tests (per ray, or per wave where a wave tests a box
                             once for all its rays)                                          */
    uint64_t tri_tests;   /* ray / triangle tests                                            */
    uint64_t node_visits; /* inner-node fetches                                              */
    uint64_t primary_rays, primary_box_tests, primary_tri_tests, primary_node_visits;
    uint64_t fetch_bytes; /* bytes of tree nodes and triangle records the loads request: a load
                             shared by a wave counts once, a per-ray load once per ray        */
    uint64_t primary_fetch_bytes;
    uint64_t camera_by_projection; /* 1: the camera rays were found by projecting the triangles onto the
                                      sample grid; the primary_* counters then count samples put
                                      through the triangle test (tri), object box tests (box) and
                                      blocks of 8 x 8 samples stepped over (node)                  */
    /* The launches that trace the secondary rays of a frame with diffuse scattering (scatter and
     * mirror rays, every level; their shadow rays are not among them): zero for other frames.   */
    uint64_t bounce_rays, bounce_box_tests, bounce_tri_tests, bounce_node_visits, bounce_fetch_bytes;
    /* Shadow rays of a camera + shadow frame by light-space cells (1) or by tree walk (0). By cells,
     * box_tests counts the light-space box tests of rays still unoccluded, tri_tests the pairs put
     * through the triangle test, and the shadow rays add nothing to node_visits and fetch_bytes.
     * shadow_list_*: the per-cell triangle lists of the frame (all entries, the longest list, the
     * cells that have one); zero when the lists were not built. shadow_redone: rays whose occluder
     * failed the box test of its leaf of eight (src/main.rs:192-200) and that the reference's own
     * search then decided; normally zero.                                                         */
    uint64_t shadow_by_cells, shadow_list_entries, shadow_list_longest, shadow_list_cells, shadow_redone;
    uint64_t tree_built_on_device; /* 1: the scene's trees were built on the GPU, 0: on the host  */
} rt_stats;

/* Number of events rt_render_device_events records. */
#define RT_RENDER_EVENTS 8

#define RT_TILE 32u   /* screen tiles are RT_TILE x RT_TILE pixels                           */

/* ---- context --------------------------------------------------------------------------- */
/* Binds the library to GPU devices[0] (n must be 1: one process per GPU). Fails with
 * RT_ERR_DEVICE when no gfx950 device is available. No reference counterpart: the reference
 * runs on the host only. */
int  rt_ctx_create(const int* devices, int n, rt_ctx** out);
void rt_ctx_destroy(rt_ctx*);
/* The library's switches (environment variables RAYS_HIP_*: which search
 * a frame's rays take, limits, verbosity) are read when the context is created; this reads them
 * again, for hosts that change them between calls (the tests and the benchmark do). No reference
 * counterpart: the reference's switches are compile-time constants (src/main.rs:29-36). */
int  rt_ctx_reload_env(rt_ctx*);

/* ---- scene ----------------------------------------------------------------------------- */
/* Replaces optimize_model + pack_model (src/opt.rs:11-80, src/main.rs:450-454): sorts every
 * object's triangles with the reference's key, precomputes e1 = b - a, e2 = c - a, v0 = a per
 * triangle as pack_triangles does (src/simd_accel.rs:8-66), builds the traversal structure and
 * uploads it. */
int  rt_scene_build(rt_ctx*, const rt_object_desc* objs, uint32_t n_objs, rt_scene** out);
/* As rt_scene_build, with flags that say where the traversal structure is built. RT_BUILD_DEVICE: on
 * the GPU (triangles in Morton order, a binary radix tree collapsed to the walks' node widths), in
 * milliseconds to a second for millions of triangles. RT_BUILD_HOST: on the host's cores by binned
 * surface-area splits, seconds for millions of triangles; its trees are walked 4 to 12 % faster on scenes
 * of one to three million triangles and no faster from about ten million on. Neither flag (and
 * rt_scene_build): the host below 4,000,000 triangles, the device from there on. Results are the same
 * either way: the nearest hit does not depend on the structure. No reference counterpart (the reference
 * has no tree, src/opt.rs:38-80 only packs leaves). */
#define RT_BUILD_DEVICE 1u
#define RT_BUILD_HOST 2u
int  rt_scene_build_ex(rt_ctx*, const rt_object_desc* objs, uint32_t n_objs, uint32_t flags, rt_scene** out);
void rt_scene_destroy(rt_scene*);
/* Total triangle count (the hit-id space). */
uint32_t rt_scene_num_tris(const rt_scene*);
/* Device bytes of the scene's arrays: sizes[0] = wide tree (32 B per child), [1] = 4-wide tree
 * (128 B per node; the persistent per-lane walk reads a 64-byte form of the same nodes, kept beside
 * it: another half of this figure), [2] = triangle records in tree order (48 B per triangle),
 * [3] = normals and colours (24 B per triangle). For measurement; no reference counterpart. */
int  rt_scene_sizes(const rt_scene*, uint64_t sizes[4]);
/* optimize_model's permutation for one object: order[i] = input position of the triangle at
 * sorted position i (src/opt.rs:12-35). */
int  rt_scene_order(const rt_scene*, uint32_t obj, uint32_t* order);
/* The packed arrays pack_model produces (src/opt.rs:38-80), read back from device memory:
 * tris9 = n_tris * 9 floats (e1 xyz, e2 xyz, v0 xyz) in sorted order, leaf_bounds = n_tris/8
 * boxes of 6 floats (min xyz, max xyz). Either pointer may be null. */
int  rt_scene_packed(const rt_scene*, uint32_t obj, float* tris9, float* leaf_bounds6);

/* ---- search ---------------------------------------------------------------------------- */
/* Replaces the nearest-hit search of trace_ray (src/main.rs:93-171) for n rays. Host buffers. */
int  rt_trace_closest(rt_scene*, const rt_ray* rays, uint64_t n, rt_hit* out);
/* Replaces the shadow search of trace_ray (src/main.rs:185-233): occluded[i] = 1 when the ray
 * hits anything. Host buffers. */
int  rt_trace_any(rt_scene*, const rt_ray* rays, uint64_t n, uint8_t* occluded);
/* Same two searches on device buffers. */
int  rt_trace_closest_device(rt_scene*, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                             void* stream);
int  rt_trace_any_device(rt_scene*, const rt_ray* d_rays, uint64_t n, uint8_t* d_occluded,
                         void* stream);
/* rt_trace_closest_device for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc. No counterpart in the reference. Same arguments
 * and results as rt_trace_closest_device; it calls no hipMalloc / hipFree or any other allocating
 * routine. The work counter of its kernel lives in a 512-byte file-scope device array that every GPU
 * the library runs on holds until the process ends (rt_ctx_create looks its address up once per
 * device). At most 2^31 rays per call (RT_ERR_UNSUPPORTED beyond). Every call on a GPU uses that
 * GPU's array: calls that overlap on one GPU (different streams, or contexts on the same device)
 * race; calls on one stream, or ordered by the caller, do not. */
int  rt_trace_closest_device_noalloc(rt_scene*, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                                     void* stream);
/* The reference's own search order run on the GPU: every 8-triangle leaf box of every object
 * in turn, no tree (src/main.rs:100-171). Slow; exists so that arithmetic parity can be checked
 * apart from tree traversal. Host buffers. */
int  rt_trace_closest_linear(rt_scene*, const rt_ray* rays, uint64_t n, rt_hit* out);

/* ---- frame ----------------------------------------------------------------------------- */
/* Replaces draw (src/main.rs:313-379) including the trace_ray recursion under it
 * (src/main.rs:88-311): camera rays, nearest-hit search, sun shadow search, shading, mirror and
 * diffuse bounces, gamma, sub-sample average, RGBA8. rgba8 = width * height * 4 bytes (host).
 * hit_ids (nullable) = width * height * grid * grid primary hit ids, [pixel][sample]. With
 * shard_count > 1 only this shard's tiles are written; the rest of the buffers is untouched. */
int  rt_render(rt_scene*, const rt_render_params*, uint8_t* rgba8, uint32_t* hit_ids);
/* Device-buffer form. With shard_count > 1, d_rgba8 is this shard's compact tile buffer of
 * rt_shard_tiles() tiles, each RT_TILE * RT_TILE * 4 bytes, in ascending tile id. The frames of one
 * scene share the scene's per-sample workspace: enqueue them on one stream, or order the streams
 * yourself; frames of different scenes may overlap. The caller's buffers are written on `stream`
 * only (by the frame's last kernel). Camera + shadow frames (no bounces) keep two frames in flight:
 * the phases before that last kernel run on streams of the scene's own, on alternating sets of
 * workspace, so that consecutive frames overlap on the GPU; they are ordered behind nothing on
 * `stream` (they read the immutable scene and the parameters, which are copied at the call) and the
 * last kernel waits for them. RAYS_HIP_FRAMES_IN_FLIGHT=1 in the environment keeps every phase on
 * `stream` (needed under stream capture), except that the triangle lists of the light-space shadow
 * search (scenes of 32,768 triangles or more) are built on a stream of the scene's own that is forked
 * from `stream` and joined to it again by events inside the call. */
int  rt_render_device(rt_scene*, const rt_render_params*, uint8_t* d_rgba8, uint32_t* d_hit_ids,
                      void* stream);
/* As rt_render_device, and records the caller's hipEvent_t handles on the stream around the
 * frame's phases: events[0] before the camera-ray kernel, [1] after it, [2] after the shadow-ray
 * sort, [3] after the shadow-ray kernel, [4] after the resolve kernel. A frame that runs as one
 * kernel records [0] before and [1]..[4] after it. Frames with diffuse scattering also record [5]
 * before and [6] after the launches that trace the secondary rays (all levels of the frame's last
 * run of tiles; [3] then follows the shadow rays of those levels); other frames record [5] and
 * [6] together with [4]. [7] lies between [0] and [1]: the camera rays' set-up (camera table,
 * binning of the triangles by screen tile) is done and their main kernel is next; a one-kernel
 * frame records it with the rest. With two frames in flight [0]..[3] and [7] are recorded on the
 * scene's own stream and the phases of consecutive frames overlap. For measurement; no reference
 * counterpart. */
int  rt_render_device_events(rt_scene*, const rt_render_params*, uint8_t* d_rgba8, uint32_t* d_hit_ids,
                             void* stream, void* const* events, int n_events);
/* As rt_render_device, from an instrumented build of the same kernels; synchronises and fills
 * *stats. Used for the roofline figures, never for timing. */
int  rt_render_device_stats(rt_scene*, const rt_render_params*, uint8_t* d_rgba8,
                            uint32_t* d_hit_ids, rt_stats* stats);
/* Number of tiles shard_index owns out of shard_count for a width x height frame. */
uint32_t rt_shard_tiles(uint32_t width, uint32_t height, uint32_t shard_index,
                        uint32_t shard_count);
/* Scatters gathered compact tile buffers back into a width * height * 4 frame. d_gathered holds
 * shard_count slabs of slab_tiles tiles each (slab s = the output of shard s). No reference
 * counterpart: the reference renders on one host. */
int  rt_deinterleave_tiles_device(rt_ctx*, const uint8_t* d_gathered, uint32_t slab_tiles,
                                  uint32_t width, uint32_t height, uint32_t shard_count,
                                  uint8_t* d_rgba8, void* stream);

/* ---- several GPUs ----------------------------------------------------------------------- */
/* One process per GPU; every process builds the same scene and renders its shard of the frame
 * (rt_render_params.shard_index / shard_count = its rank and the number of ranks) into a compact
 * slab of its tiles; one exchange step per frame brings the slabs to rank 0. No reference
 * counterpart: the reference renders on one host (src/main.rs:313-379).
 *
 * rt_gather_tiles_device is that step. nccl_comm is an RCCL communicator (ncclComm_t) over the
 * ranks, made by the host's own RCCL binding or by rt_comm_create; rank and rank count are read
 * from it. Every rank passes its slab (the d_rgba8 of its rt_render_device call: rt_shard_tiles()
 * tiles of RT_TILE * RT_TILE * 4 bytes); ranks other than 0 enqueue one ncclSend of it and ignore
 * d_rgba8; rank 0 enqueues its own copy, one ncclRecv per peer (grouped) and the scatter of all
 * tiles into d_rgba8 = width * height * 4 bytes. Everything is enqueued on `stream`; nothing
 * synchronises (except once when rank 0's receive slab grows). */
int  rt_gather_tiles_device(rt_ctx*, void* nccl_comm, const uint8_t* d_slab, uint32_t width,
                            uint32_t height, uint8_t* d_rgba8, void* stream);
/* Communicator helpers for hosts without an RCCL binding of their own: rank 0 calls
 * rt_comm_id_create and hands the 128 bytes to the other ranks over any host channel; then every
 * rank calls rt_comm_create (collective: ncclCommInitRank on the context's GPU). */
#define RT_COMM_ID_BYTES 128
typedef struct { char bytes[RT_COMM_ID_BYTES]; } rt_comm_id;
int  rt_comm_id_create(rt_comm_id* out);
int  rt_comm_create(rt_ctx*, const rt_comm_id*, int rank, int n_ranks, void** nccl_comm);
int  rt_comm_ranks(void* nccl_comm, int* rank, int* n_ranks);
int  rt_comm_destroy(void* nccl_comm);

/* ---- misc ------------------------------------------------------------------------------ */
const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
