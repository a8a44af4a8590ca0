// rt_api_internal.h -- the objects behind the opaque handles of include/rays_hip.h, shared by the
// translation units that implement its entry points (rt_api.cpp, rt_noalloc.hip, rt_gather.cpp).
#ifndef RT_API_INTERNAL_H
#define RT_API_INTERNAL_H

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/rays_hip.h"
#include "rt_device.h"

// Records the message rt_last_error() returns on the calling thread; returns `code`.
int rt_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// RT_OK while the process holds one HIP runtime; RT_ERR_UNSUPPORTED (with both paths in the message) once a
// second libamdhip64 is mapped. Called first by every entry point that takes a device, a stream, an event or
// a device pointer from the caller.
int rt_runtime_guard();

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rt_fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, "%s: %s", #expr, \
                           hipGetErrorString(e_));                                                \
    } while (0)

struct TileMap {
    uint32_t tiles_x = 0, tiles_y = 0;
    std::vector<uint32_t> list;  // tiles of one shard, ascending
    uint32_t* d_list = nullptr;
    std::vector<uint32_t> slot;  // every tile's rank within its shard
    uint32_t* d_slot = nullptr;
};

// The RAYS_HIP_* switches a frame or a batch looks at, read from the environment when the context is
// created and again on rt_ctx_reload_env (tests and the benchmark's legs change them between calls): the
// per-call path reads none of them from the environment.
struct RtOptions {
    bool megakernel = false;      // RAYS_HIP_MEGAKERNEL: one-kernel frames also where the pipeline would run
    bool one_in_flight = false;   // RAYS_HIP_FRAMES_IN_FLIGHT=1: every frame on the caller's stream
    bool verbose = false;         // RAYS_HIP_VERBOSE
    bool sort_rocprim = false;    // RAYS_HIP_SORT=rocprim: the library sort for the shadow rays' keys
    bool cells_redo_all = false;  // RAYS_HIP_CELLS_REDO=all (tests)
    char raster = 0;              // RAYS_HIP_RASTER: '0' no projection, 'a' its older form, 0 default
    char shadow = 0;              // RAYS_HIP_SHADOW: 'c' cells whenever they fit, 'w' tree walk, 0 default
    char trace = 0;               // RAYS_HIP_TRACE: 'l' per-lane walk to the end, 'w' wave walks, 0 persistent walk
    long long bounce_rays = 0;    // RAYS_HIP_BOUNCE_RAYS: secondary-ray records per run of tiles (0: default)
    double raster_ratio = -1.0;   // RAYS_HIP_RASTER_RATIO: work limit of projection (< 0: default)
    long cell_mean = -1;          // RAYS_HIP_SHADOW_CELL_MEAN: longest mean list the cells are taken for (< 0: default)
    long sort_bits = 0;           // RAYS_HIP_SORT_BITS: key bits sorted on (0: by frame size)
    void read_env();
};

struct rt_ctx {
    int device = 0;
    std::mutex mu;
    RtOptions opt;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, TileMap> tile_maps;
    // Work counters of the persistent kernels: one per launch, taken in turn, so that launches in
    // flight on different streams do not share one.
    static constexpr uint32_t N_COUNTERS = 256;
    uint32_t* d_counters = nullptr;
    uint32_t next_counter = 0;
    // Scenes built on this context and not yet destroyed. rt_ctx_destroy on a context that still
    // has scenes only marks it; the last rt_scene_destroy then deletes it.
    uint32_t live_scenes = 0;
    bool destroy_requested = false;
    // This device's copy of the file-scope scratch array of rt_noalloc.hip (its device address).
    void* noalloc_scratch = nullptr;
    // Slab that rt_gather_tiles_device receives the shards' tiles into (rank 0 only), grown on demand.
    uint8_t* d_gather = nullptr;
    size_t gather_bytes = 0;
    ~rt_ctx() {
        if (d_counters) (void)hipFree(d_counters);
        if (d_gather) (void)hipFree(d_gather);
        for (auto& kv : tile_maps) {
            if (kv.second.d_list) (void)hipFree(kv.second.d_list);
            if (kv.second.d_slot) (void)hipFree(kv.second.d_slot);
        }
    }
};

// Per-sample buffers of the camera + shadow pipeline, grown on demand and kept with the scene.
struct FrameWork {
    size_t capacity = 0;  // sample slots
    float4* samp = nullptr;
    uint32_t* state = nullptr;
    uint32_t* keys[2] = {nullptr, nullptr};
    uint32_t* order = nullptr;  // slot numbers in sorted key order (before the sort: ray queues of the levels)
    uint32_t* counts = nullptr; // rays queued per level of the secondary-ray tree
    unsigned long long* zbuf = nullptr;  // nearest (t, id) per sample of a frame rendered by projection
    size_t zbuf_words = 0;
    bool zbuf_clean = false;             // every word says "none" (the record pass of a whole frame leaves it so)
    float* cam_table = nullptr;  // image-plane coordinates of the sample columns and rows
    size_t cam_table_floats = 0;
    // Projection by screen tile: triangles per raster tile (counts, then fill cursors), list offsets,
    // the lists. The lists' total length depends on the scene (immutable) and on the frame described by
    // bin_key only, so it is read back once per such frame and kept.
    uint32_t* bin_counts = nullptr;
    uint32_t* bin_offsets = nullptr;
    size_t bin_tiles = 0;
    uint32_t* bin_list = nullptr;
    size_t bin_capacity = 0;
    float4* bin_recs = nullptr;      // per triangle: what the counting pass prepared (96 B)
    size_t bin_recs_tris = 0;
    uint32_t bin_key[5] = {0, 0, 0, 0, 0};   // width, height, grid, shard_index, shard_count
    void* temp = nullptr;
    size_t temp_bytes = 0;
    void* radix_work = nullptr;      // workspace of the library's own sort (rtk::radix_sort_valid)
    uint32_t* n_valid = nullptr;     // device word: shadow rays among the sorted keys
    // Shadow rays by light-space cells: triangles per cell, the lists (24 bytes an entry: patch, id and
    // depth; the same room for every cell), and the device words that say whether the lists are usable.
    uint32_t* lc_counts = nullptr;
    size_t lc_cells = 0;
    float* lc_lists = nullptr;
    size_t lc_capacity = 0;
    uint32_t* lc_flags = nullptr;
    void release() {
        for (void* q : {(void*)samp, (void*)state, (void*)keys[0], (void*)keys[1], (void*)order, (void*)counts, (void*)zbuf, (void*)cam_table, temp,
                        (void*)bin_counts, (void*)bin_offsets, (void*)bin_list, (void*)bin_recs, radix_work, (void*)n_valid,
                        (void*)lc_counts, (void*)lc_lists, (void*)lc_flags})
            if (q) (void)hipFree(q);
        *this = FrameWork();
    }
};

struct rt_scene {
    rt_ctx* ctx = nullptr;
    FrameWork work;         // per sample: camera rays and their shadow rays
    FrameWork bounce_work;  // per scatter ray of a run of tiles
    // Two frames in flight (camera + shadow frames that run as one piece): consecutive frames alternate
    // between `work` and `work_b` and between two streams of the scene's own, on which everything up to
    // the shadow kernel runs; only the resolve kernel, which writes the caller's buffer, runs on the
    // caller's stream. A frame's memory-bound phases then overlap the next frame's compute-bound ones.
    // pipe_ready[b]: the shadow kernel of the frame in set b is done. pipe_free[b]: the resolve kernel
    // that read set b is done (recorded on the caller's stream).
    FrameWork work_b;
    hipStream_t pipe_stream[2] = {nullptr, nullptr};
    hipEvent_t pipe_ready[2] = {nullptr, nullptr}, pipe_free[2] = {nullptr, nullptr};
    bool pipe_free_set[2] = {false, false};
    // Streams on which a frame's light-space triangle lists are built beside its camera rays, one per
    // set of buffers, with the events that order them against the frame's own stream.
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t side_fork[2] = {nullptr, nullptr}, side_done[2] = {nullptr, nullptr};
    uint64_t pipe_frames = 0;
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};  // over the objects that have a tree
    // Camera rays by projection (the camera of src/main.rs:314-330 stands at the origin and looks along
    // +z): possible when no object has trailing triangles and every triangle lies in front of the
    // camera; raster_dx / raster_dy are the widest extents of a triangle on the image plane z = 1.
    bool raster_ok = false;
    float raster_dx = 0.0f, raster_dy = 0.0f;
    double raster_sum_dx = 0.0, raster_sum_dy = 0.0, raster_sum_area = 0.0;  // over all triangles
    // Mean side of a triangle's bounding box along x, y, z: how many cells of the light-space grid a
    // triangle's patch is likely to touch (the estimate that decides whether the lists are tried).
    double mean_tri_side[3] = {0.0, 0.0, 0.0};
    uint32_t n_tris_total = 0;
    uint32_t n_rest_total = 0;   // trailing n % 8 triangles of all objects together
    bool built_on_device = false;
    SceneView view{};
    std::vector<void*> device_ptrs;
    uint32_t n_tris = 0;
    uint32_t n_leaves = 0;
    uint32_t n_nodes = 0;
    uint64_t sizes[4] = {0, 0, 0, 0};  // device bytes: wide tree, 4-wide tree, triangle records, tri_data
    uint32_t max_stack = 0;
    uint32_t wide_depth = 1;
    std::vector<DevObject> objects;
    std::vector<std::vector<uint32_t>> order;  // per object: sorted position -> input position
    ~rt_scene() {
        for (int b = 0; b < 2; ++b) {
            if (pipe_stream[b]) {
                (void)hipStreamSynchronize(pipe_stream[b]);
                (void)hipStreamDestroy(pipe_stream[b]);
            }
            if (side_stream[b]) {
                (void)hipStreamSynchronize(side_stream[b]);
                (void)hipStreamDestroy(side_stream[b]);
            }
            if (side_fork[b]) (void)hipEventDestroy(side_fork[b]);
            if (side_done[b]) (void)hipEventDestroy(side_done[b]);
            if (pipe_ready[b]) (void)hipEventDestroy(pipe_ready[b]);
            if (pipe_free[b]) (void)hipEventDestroy(pipe_free[b]);
        }
        work.release();
        work_b.release();
        bounce_work.release();
        for (void* p : device_ptrs) (void)hipFree(p);
    }
};

// rt_noalloc.hip: looks up the device address of the file-scope scratch array on the current
// device (the caller has made ctx->device current) and keeps it in ctx->noalloc_scratch.
int rt_noalloc_setup(rt_ctx* ctx);

#endif
