// rt_api.cpp -- the C ABI of include/rays_hip.h: context, scene build, search and frame calls.
// Host code; every piece of arithmetic that reaches a result runs in the kernels.

#include <hip/hip_runtime.h>
#include <link.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/rays_hip.h"
#include "bvh_build.h"
#include "rt_api_internal.h"
#include "rt_device.h"
#include "rt_kernels.h"

namespace {
thread_local std::string g_last_error;
}

int rt_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// A process can end up with two HIP runtimes: this library binds to the system's libamdhip64.so.7, and a
// Python wheel of PyTorch brings a copy of its own under another path, which the loader maps as well when
// it comes second (loaded first, its soname satisfies this library too and there is one runtime). Streams,
// events and device pointers of one runtime mean nothing to the other, and handing them across ends in an
// abort inside the runtime. So every entry point that takes such a handle from the caller first makes sure
// the process holds one runtime, and refuses otherwise. The walk over the loaded objects is repeated only
// when the loader's count of objects added (dlpi_adds) has changed since the last one.
namespace {
struct RuntimeScan {
    unsigned long long adds = 0;
    int copies = 0;
    char first[256] = "", second[256] = "";
};
int scan_adds(struct dl_phdr_info* info, size_t, void* out) {
    *static_cast<unsigned long long*>(out) = info->dlpi_adds;
    return 1;   // the counter is the same in every object's record: stop at the first
}
int scan_objects(struct dl_phdr_info* info, size_t, void* out) {
    RuntimeScan* sc = static_cast<RuntimeScan*>(out);
    sc->adds = info->dlpi_adds;
    const char* name = info->dlpi_name ? info->dlpi_name : "";
    const char* base = std::strrchr(name, '/');
    base = base ? base + 1 : name;
    if (std::strncmp(base, "libamdhip64.so", 14) != 0) return 0;
    if (sc->copies == 0) std::snprintf(sc->first, sizeof(sc->first), "%s", name);
    else if (std::strcmp(sc->first, name) == 0) return 0;
    else if (sc->copies == 1) std::snprintf(sc->second, sizeof(sc->second), "%s", name);
    sc->copies += 1;
    return 0;
}
std::mutex g_scan_mu;
RuntimeScan g_scan;
std::atomic<unsigned long long> g_scan_adds{~0ull};
std::atomic<int> g_scan_copies{0};
}  // namespace

int rt_runtime_guard() {
    unsigned long long adds = 0;
    dl_iterate_phdr(scan_adds, &adds);
    if (adds != g_scan_adds.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(g_scan_mu);
        RuntimeScan sc;
        dl_iterate_phdr(scan_objects, &sc);
        g_scan = sc;
        g_scan_copies.store(sc.copies, std::memory_order_release);
        g_scan_adds.store(sc.adds, std::memory_order_release);
    }
    if (g_scan_copies.load(std::memory_order_acquire) <= 1) return RT_OK;
    std::lock_guard<std::mutex> lock(g_scan_mu);
    return rt_fail(RT_ERR_UNSUPPORTED,
                   "two HIP runtimes in one process (%s and %s): their streams, events and pointers do not mix; load the "
                   "other one first (import torch before this library) so that both bind to the same copy",
                   g_scan.first, g_scan.second);
}

namespace {

// Owns a set of device allocations; frees them unless released.
struct DevPool {
    std::vector<void*> ptrs;
    ~DevPool() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    hipError_t alloc(void** out, size_t bytes) {
        *out = nullptr;
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
    void release() { ptrs.clear(); }
};


inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    return (int)std::min(16u, n);
}

// f32::min / f32::max as glam's Vec3::min / max use them (the non-NaN operand wins).
inline float g_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float g_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

// Relative growth of a projected triangle's bounding box that holds every camera sample the exact
// triangle test can accept: the bound of raster_setup (rt_kernels.hip) in double precision, for
// |dx| <= 1 (the reference's aspect of 1; wider frames grow it a little: an estimate for the work
// limit, the kernel computes its own), with the edges rounded to float as pack_triangles stores them.
// Infinity when the triangle's plane passes too close to the camera for a bound.
double raster_grow(const float* a, const float* b, const float* c) {
    double e1[3], e2[3], sv[3];
    for (int k = 0; k < 3; ++k) {
        e1[k] = (double)(float)(b[k] - a[k]);
        e2[k] = (double)(float)(c[k] - a[k]);
        sv[k] = -(double)a[k];
    }
    const double Dx = 1.0, Dy = 1.0;
    auto ab = [](double v) { return std::fabs(v); };
    const double r[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
    const double T = e2[0] * r[0] + e2[1] * r[1] + e2[2] * r[2];
    const double qh[3] = {Dy * ab(e2[2]) + ab(e2[1]), ab(e2[0]) + Dx * ab(e2[2]), Dx * ab(e2[1]) + Dy * ab(e2[0])};
    const double Mu = ab(sv[0]) * qh[0] + ab(sv[1]) * qh[1] + ab(sv[2]) * qh[2];
    const double Ma = ab(e1[0]) * qh[0] + ab(e1[1]) * qh[1] + ab(e1[2]) * qh[2];
    const double rh[3] = {ab(sv[1]) * ab(e1[2]) + ab(sv[2]) * ab(e1[1]), ab(sv[2]) * ab(e1[0]) + ab(sv[0]) * ab(e1[2]),
                          ab(sv[0]) * ab(e1[1]) + ab(sv[1]) * ab(e1[0])};
    const double Mv = Dx * rh[0] + Dy * rh[1] + rh[2];
    const double Mt = ab(e2[0]) * rh[0] + ab(e2[1]) * rh[1] + ab(e2[2]) * rh[2];
    const double U16 = 9.5367431640625e-7;
    const double thr = U16 * (Mu + Mv + 2.0 * (Ma + Mu + Mv)) + 3e-37;
    // Stricter than the kernel's own float test (|T| - 16 u Mt > 0), so that the two never disagree.
    const double Tlow = ab(T) - 5.0 * U16 * Mt;
    if (!(Tlow > 0.0)) return INFINITY;
    const double zmax = std::max(std::max((double)a[2], (double)b[2]), (double)c[2]);
    return thr * zmax / Tlow * 1.01;
}

}  // namespace


void RtOptions::read_env() {
    auto first = [](const char* name) -> char {
        const char* e = std::getenv(name);
        return e ? e[0] : 0;
    };
    auto num = [](const char* name, double none) -> double {
        const char* e = std::getenv(name);
        return e ? std::atof(e) : none;
    };
    *this = RtOptions();
    megakernel = std::getenv("RAYS_HIP_MEGAKERNEL") != nullptr;
    one_in_flight = (int)num("RAYS_HIP_FRAMES_IN_FLIGHT", 0.0) == 1;
    verbose = std::getenv("RAYS_HIP_VERBOSE") != nullptr;
    sort_rocprim = first("RAYS_HIP_SORT") == 'r';
    cells_redo_all = first("RAYS_HIP_CELLS_REDO") == 'a';
    raster = first("RAYS_HIP_RASTER");
    shadow = first("RAYS_HIP_SHADOW");
    trace = first("RAYS_HIP_TRACE");
    bounce_rays = (long long)num("RAYS_HIP_BOUNCE_RAYS", 0.0);
    raster_ratio = num("RAYS_HIP_RASTER_RATIO", -1.0);
    cell_mean = std::getenv("RAYS_HIP_SHADOW_CELL_MEAN") ? std::max(0L, (long)num("RAYS_HIP_SHADOW_CELL_MEAN", 0.0)) : -1L;
    sort_bits = (long)num("RAYS_HIP_SORT_BITS", 0.0);
}

// Scenes below this many triangles use all the walk's counters (RAYS_HIP_WALK_SHARDS_BELOW; tuning).
static uint32_t walk_shards_below() {
    if (const char* e = std::getenv("RAYS_HIP_WALK_SHARDS_BELOW")) return (uint32_t)std::strtoul(e, nullptr, 10);
    return 2000000u;
}

// A zeroed block of work counters for one launch on `stream` (the caller holds ctx->mu).
int take_counter(rt_ctx* ctx, hipStream_t stream, uint32_t** out) {
    if (!ctx->d_counters) HIP_TRY(hipMalloc((void**)&ctx->d_counters, (size_t)rt_ctx::N_COUNTERS * RT_COUNTER_BYTES));
    uint32_t* c = ctx->d_counters + (RT_COUNTER_BYTES / 4u) * (ctx->next_counter++ % rt_ctx::N_COUNTERS);
    HIP_TRY(hipMemsetAsync(c, 0, RT_COUNTER_BYTES, stream));
    *out = c;
    return RT_OK;
}



namespace {

// Tiles of a width x height frame owned by shard `index` of `count`: tile (tx, ty) belongs to
// shard (tx + ty) % count, so neighbouring tiles go to different GPUs in both directions.
int get_tile_map(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t index, uint32_t count, TileMap** out) {
    auto key = std::make_tuple(width, height, index, count);
    auto it = ctx->tile_maps.find(key);
    if (it != ctx->tile_maps.end()) {
        *out = &it->second;
        return RT_OK;
    }
    TileMap tm;
    tm.tiles_x = ceil_div(width, RT_TILE);
    tm.tiles_y = ceil_div(height, RT_TILE);
    const uint32_t n = tm.tiles_x * tm.tiles_y;
    tm.slot.resize(n);
    std::vector<uint32_t> next(count, 0);
    for (uint32_t t = 0; t < n; ++t) {
        const uint32_t shard = (t % tm.tiles_x + t / tm.tiles_x) % count;
        tm.slot[t] = next[shard]++;
        if (shard == index) tm.list.push_back(t);
    }
    HIP_TRY(hipMalloc((void**)&tm.d_list, std::max<size_t>(16, tm.list.size() * 4)));
    HIP_TRY(hipMalloc((void**)&tm.d_slot, std::max<size_t>(16, tm.slot.size() * 4)));
    if (!tm.list.empty()) HIP_TRY(hipMemcpy(tm.d_list, tm.list.data(), tm.list.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(tm.d_slot, tm.slot.data(), tm.slot.size() * 4, hipMemcpyHostToDevice));
    auto ins = ctx->tile_maps.emplace(key, std::move(tm));
    *out = &ins.first->second;
    return RT_OK;
}

int check_render_params(const rt_render_params* p, uint32_t* spp) {
    if (!p) return rt_fail(RT_ERR_INVALID, "render params are null");
    if (p->width == 0 || p->height == 0) return rt_fail(RT_ERR_INVALID, "frame size is zero");
    if ((uint64_t)p->width * p->height > (1ull << 30)) return rt_fail(RT_ERR_INVALID, "frame too large");
    if (p->grid != 1 && p->grid != 2 && p->grid != 4 && p->grid != 8)
        return rt_fail(RT_ERR_UNSUPPORTED, "grid must be 1, 2, 4 or 8 (got %u)", p->grid);
    if (p->reflect_depth > RT_MAX_REFLECT + 1)
        return rt_fail(RT_ERR_UNSUPPORTED, "reflect_depth above %d is not supported", RT_MAX_REFLECT + 1);
    if (p->diffuse_depth > 2) return rt_fail(RT_ERR_UNSUPPORTED, "diffuse_depth above 2 is not supported");
    if (p->diffuse_depth > 1 && p->diffuse_scatt == 0) return rt_fail(RT_ERR_INVALID, "diffuse_scatt is zero");
    if (p->shard_count > 4096) return rt_fail(RT_ERR_UNSUPPORTED, "more than 4096 shards");
    if (p->shard_count == 0 || p->shard_index >= p->shard_count)
        return rt_fail(RT_ERR_INVALID, "shard_index %u out of range for shard_count %u", p->shard_index, p->shard_count);
    *spp = p->grid * p->grid;
    return RT_OK;
}

// Per-sample (or per-ray) arrays of the kernel pipeline, grown on demand and kept with the scene.
int ensure_work(FrameWork& w, size_t n, hipStream_t stream) {
    if (w.capacity >= n) return RT_OK;
    if (n >= (1ull << 32)) return rt_fail(RT_ERR_UNSUPPORTED, "too many rays in one launch");
    w.release();
    size_t tb = 0;
    // The temporary size is asked for the widest key, so one workspace serves every key width.
    HIP_TRY(rtk::sort_pairs_u32(nullptr, &tb, nullptr, nullptr, nullptr, nullptr, n, 32, stream));
    HIP_TRY(hipMalloc((void**)&w.samp, n * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&w.state, n * 4));
    for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc((void**)&w.keys[k], n * 4));
    // The unsorted values are the slot numbers themselves and are never stored.
    HIP_TRY(hipMalloc((void**)&w.order, n * 4));
    HIP_TRY(hipMalloc((void**)&w.counts, (RT_MAX_REFLECT + 1) * 4));
    HIP_TRY(hipMalloc(&w.temp, std::max<size_t>(tb, 16)));
    w.temp_bytes = tb;
    HIP_TRY(hipMalloc(&w.radix_work, std::max<size_t>(rtk::radix_workspace_bytes(n), 16)));
    HIP_TRY(hipMalloc((void**)&w.n_valid, 256));
    w.capacity = n;
    return RT_OK;
}

// Projection from the sun across its dominant axis, scaled to the scene's bounds, and the number of
// key bits to sort n shadow-ray candidates on.
// Mean length the light-space lists are expected to have with 2^key_bits cells: a triangle whose box
// has mean sides (a, b) across the light goes to about (1 + a / cell) (1 + b / cell) cells, cell = the
// scene's side / cells a side. Runs about 1.4 x high on random soups.
static double cells_mean_list(const rt_scene* s, const LightProj& lp, uint32_t key_bits) {
    const double per_axis = (double)(1u << (key_bits / 2u));
    double lists_per_tri = 1.0;
    for (int axis : {lp.a1, lp.a2}) {
        const double cell = ((double)s->bounds_hi[axis] - s->bounds_lo[axis]) / per_axis;
        lists_per_tri *= cell > 0.0 ? 1.0 + s->mean_tri_side[axis] / cell : 1.0;
    }
    return lists_per_tri * s->n_tris_total / (double)((size_t)1 << key_bits);
}
// Longest mean list the cells are taken for (RAYS_HIP_SHADOW_CELL_MEAN overrides): the work per ray
// grows with it, the tree walk's does not.
static uint32_t cells_mean_limit(const rt_scene* s) {
    return s->ctx->opt.cell_mean >= 0 ? (uint32_t)s->ctx->opt.cell_mean : 192u;
}

// coarse: the frame's shadow rays may go by light-space cells; then 16 key bits are kept whenever the
// lists come out short enough with them, however many rays there are (more rays per cell only mean
// more waves per list, and every 8 bits more cost the sort a pass: 8K frame of a million triangles,
// 16 against 18 bits: sort 1.0 / 1.5 ms, shadow rays 2.2 / 2.0 ms).
LightProj light_projection(const rt_scene* s, const rt_render_params* p, size_t n, bool coarse = false) {
    LightProj lp{};
    // Only the leading bits of the interleaved key are kept and sorted: the sort is stable, so the
    // rays of one cell keep their screen order, which is coherent enough within a cell of a few
    // hundred rays, and each 8 bits less saves one pass over the pairs. Key bits = log2(n / 512)
    // rounded up to an even number, at least 16: a cell must stay narrower than the widest beam the
    // walk accepts (1/64 of the scene) or the waves stop forming beams.
    {
        const long env_bits = s->ctx->opt.sort_bits;
        uint32_t kb = 16;
        while (kb < 32 && ((size_t)512 << kb) < n) kb += 2;
        lp.key_bits = kb;
        if (env_bits >= 2 && env_bits <= 32) lp.key_bits = (uint32_t)env_bits & ~1u, coarse = false;
    }
    float c[3], dsun[3];
    for (int k = 0; k < 3; ++k) {
        c[k] = 0.5f * (s->bounds_lo[k] + s->bounds_hi[k]);
        dsun[k] = std::fabs(c[k] - p->sun[k]);
    }
    lp.w = (dsun[0] >= dsun[1] && dsun[0] >= dsun[2]) ? 0 : (dsun[1] >= dsun[2] ? 1 : 2);
    lp.a1 = lp.w == 0 ? 1 : 0;
    lp.a2 = lp.w == 2 ? 1 : 2;
    lp.sun_w = p->sun[lp.w];
    lp.sun_a1 = p->sun[lp.a1];
    lp.sun_a2 = p->sun[lp.a2];
    lp.c_w = c[lp.w];
    lp.c_a1 = c[lp.a1];
    lp.c_a2 = c[lp.a2];
    const float cw = lp.c_w - lp.sun_w;
    bool ok = std::fabs(cw) > 0.0f;
    lp.uc = ok ? (lp.c_a1 - lp.sun_a1) / cw : 0.0f;
    lp.vc = ok ? (lp.c_a2 - lp.sun_a2) / cw : 0.0f;
    float u0 = INFINITY, u1 = -INFINITY, v0 = INFINITY, v1 = -INFINITY;
    for (int corner = 0; corner < 8; ++corner) {
        float q[3];
        for (int k = 0; k < 3; ++k) q[k] = (corner >> k) & 1 ? s->bounds_hi[k] : s->bounds_lo[k];
        const float pw = q[lp.w] - lp.sun_w;
        if (!(std::fabs(pw) > 0.0f)) ok = false;
        const float dw = q[lp.w] - lp.c_w;
        const float u = ((q[lp.a1] - lp.c_a1) - lp.uc * dw) / pw, v = ((q[lp.a2] - lp.c_a2) - lp.vc * dw) / pw;
        u0 = std::min(u0, u); u1 = std::max(u1, u);
        v0 = std::min(v0, v); v1 = std::max(v1, v);
    }
    // The sun inside the scene's extent along w would make the projection change sign.
    ok = ok && ((s->bounds_lo[lp.w] - lp.sun_w) * (s->bounds_hi[lp.w] - lp.sun_w) > 0.0f);
    ok = ok && std::isfinite(u0) && std::isfinite(u1) && std::isfinite(v0) && std::isfinite(v1);
    lp.u0 = ok ? u0 : 0.0f;
    lp.v0 = ok ? v0 : 0.0f;
    lp.uscale = (ok && u1 > u0) ? 65535.0f / (u1 - u0) : 0.0f;  // 0: every ray gets the same key
    lp.vscale = (ok && v1 > v0) ? 65535.0f / (v1 - v0) : 0.0f;
    lp.uabs = ok ? std::max(std::fabs(u0), std::fabs(u1)) : 0.0f;
    lp.vabs = ok ? std::max(std::fabs(v0), std::fabs(v1)) : 0.0f;
    if (coarse && lp.key_bits > 16 && cells_mean_list(s, lp, 16) <= 1.4 * (double)cells_mean_limit(s)) lp.key_bits = 16;
    return lp;
}

// events: null, or RT_RENDER_EVENTS hipEvent_t handles recorded around the phases.
// primary_stats: null, or receives the counters as they stand after the camera-ray kernel.
int render_common(rt_scene* s, const rt_render_params* p, uint8_t* d_rgba8, uint32_t* d_hit_ids, DevStats* d_stats,
                  hipStream_t stream, void* const* events = nullptr, DevStats* primary_stats = nullptr,
                  bool* projected = nullptr, DevStats* bounce_stats = nullptr, bool* lists_built = nullptr) {
    hipStream_t ws = stream;   // where the phases before the resolve kernel run (see in_flight below)
    auto mark = [&](int k) -> hipError_t {
        return events ? hipEventRecord((hipEvent_t)events[k], (k <= 3 || k == 7) ? ws : stream) : hipSuccess;
    };
    uint32_t spp = 0;
    int rc = check_render_params(p, &spp);
    if (rc != RT_OK) return rc;
    if (!d_rgba8) return rt_fail(RT_ERR_INVALID, "rgba8 buffer is null");
    DevRenderParams dp{};
    dp.width = p->width;
    dp.height = p->height;
    dp.grid = p->grid;
    dp.spp = spp;
    dp.reflect_depth = p->reflect_depth;
    dp.diffuse_depth = p->diffuse_depth;
    dp.diffuse_scatt = p->diffuse_scatt;
    dp.sun[0] = p->sun[0];
    dp.sun[1] = p->sun[1];
    dp.sun[2] = p->sun[2];
    dp.seed = p->seed;
    dp.tiles_x = ceil_div(p->width, RT_TILE);
    dp.beam_levels = s->wide_depth;
    dp.shard_index = p->shard_index;
    dp.shard_count = p->shard_count;
    dp.shard_magic = (uint32_t)(((1ull << 32) + p->shard_count - 1) / p->shard_count);
    const uint32_t* d_list = nullptr;
    if (p->shard_count > 1) {
        TileMap* tm = nullptr;
        rc = get_tile_map(s->ctx, p->width, p->height, p->shard_index, p->shard_count, &tm);
        if (rc != RT_OK) return rc;
        dp.n_tiles = (uint32_t)tm->list.size();
        dp.compact = 1;
        d_list = tm->d_list;
    } else {
        dp.n_tiles = dp.tiles_x * ceil_div(p->height, RT_TILE);
        dp.compact = 0;
    }
    bool reflective = false;
    for (const DevObject& ob : s->objects) reflective = reflective || ob.reflectivity > 0.0f;
    const int mode = p->diffuse_depth > 1 ? 2 : ((reflective && p->reflect_depth > 1) ? 1 : 0);
    // One-kernel frame: mirror bounces without diffuse scattering (few rays, all from the sample's
    // own lane), or on request.
    const RtOptions& opt = s->ctx->opt;
    const bool pipeline = mode != 1 && !opt.megakernel;
    if (!pipeline) {
        HIP_TRY(mark(0));
        HIP_TRY(rtk::render(s->view, dp, d_list, mode, d_rgba8, d_hit_ids, d_stats, stream));
        for (int k = 1; k < RT_RENDER_EVENTS; ++k) HIP_TRY(mark(k));  // [5], [6], [7] with the rest
        if (primary_stats && d_stats) {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(primary_stats, d_stats, sizeof(DevStats), hipMemcpyDeviceToHost));
        }
        return RT_OK;
    }
    // Pipeline of small kernels. Camera + shadow rays only (mode 0): camera-ray kernel, light-space
    // sort of the shadow rays, shadow kernel, resolve kernel, over the whole frame at once. With
    // diffuse scattering (mode 2): the same, then the tree of secondary rays level by level, the
    // sorted shadow rays of all levels, and the resolve of everything, over runs of tiles sized so
    // that a run's secondary rays fit the workspace.
    const uint32_t tile_slots = RT_TILE * RT_TILE * spp;
    if ((size_t)dp.n_tiles * tile_slots >= (1ull << 32)) return rt_fail(RT_ERR_UNSUPPORTED, "frame has too many samples");
    if (dp.n_tiles == 0) return RT_OK;
    uint32_t run_tiles = dp.n_tiles;
    RayTree tr{};
    if (mode == 2) {
        // Scatter rays off every hit; off a mirror one mirror ray more, and every branch is followed
        // from mirror to mirror down to reflect_depth (the camera ray is depth 1, level 0 is depth 2).
        tr.scatt = p->diffuse_scatt;
        tr.mirror = (reflective && p->reflect_depth > 1) ? 1u : 0u;
        tr.branches = tr.scatt + tr.mirror;
        tr.levels = tr.mirror ? p->reflect_depth - 1 : 1u;
        // Secondary-ray records per run of tiles: at most 512 Mi (18 GB of the GPU's 288 GB as workspace,
        // allocated as far as a frame needs it); RAYS_HIP_BOUNCE_RAYS overrides. Fewer, larger runs are faster: the short
        // last run of a frame fills the device badly.
        const long long v = opt.bounce_rays;
        const size_t cap = v > 0 ? (size_t)v : ((size_t)512 << 20);
        const size_t per_tile = (size_t)tile_slots * tr.branches * tr.levels;
        if (per_tile >= (1ull << 31)) return rt_fail(RT_ERR_UNSUPPORTED, "too many secondary rays per tile");
        run_tiles = (uint32_t)std::min<size_t>(dp.n_tiles, std::max<size_t>(1, cap / per_tile));
    }
    const size_t run_slots = (size_t)run_tiles * tile_slots;
    // Two frames in flight: see rt_scene. Not for instrumented frames (their counters are read back
    // between the phases); RAYS_HIP_FRAMES_IN_FLIGHT=1 keeps every frame on the caller's stream.
    const bool one_in_flight = opt.one_in_flight;
    const bool in_flight = mode == 0 && run_tiles == dp.n_tiles && !d_stats && !primary_stats && !one_in_flight;
    const int set = in_flight ? (int)(s->pipe_frames++ & 1u) : 0;
    FrameWork& w = set ? s->work_b : s->work;
    rc = ensure_work(w, run_slots, stream);
    if (rc != RT_OK) return rc;
    if (mode == 2) {
        rc = ensure_work(s->bounce_work, run_slots * tr.branches * tr.levels, stream);
        if (rc != RT_OK) return rc;
    }
    FrameWork& bw = s->bounce_work;
    if (in_flight) {
        if (!s->pipe_stream[set]) {
            HIP_TRY(hipStreamCreateWithFlags(&s->pipe_stream[set], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->pipe_ready[set], hipEventDisableTiming));
        }
        ws = s->pipe_stream[set];
    }
    if (!s->pipe_free[set]) HIP_TRY(hipEventCreateWithFlags(&s->pipe_free[set], hipEventDisableTiming));
    // The buffers of this set are free once the resolve kernel of the frame that used them last is done.
    if (in_flight && s->pipe_free_set[set]) HIP_TRY(hipStreamWaitEvent(ws, s->pipe_free[set], 0));
    // Camera rays by projection, unless the frame is taller than wide (aspect 0). The older form
    // (RAYS_HIP_RASTER=atomic) steps one wave over a triangle's blocks of 8 x 8 samples and is not taken
    // when a triangle's box covers more than 16384 of them; by screen tile a large box costs little.
    bool raster = s->raster_ok && p->width >= p->height && p->width / p->height <= 4 &&
                  (uint64_t)p->width * p->grid < 65536u && (uint64_t)p->height * p->grid < 65536u &&  // 16-bit sample coordinates
                  opt.raster != '0';
    if (raster) {
        const float aspect = (float)(p->width / p->height);
        if (opt.raster == 'a' && s->n_rest_total != 0) raster = false;   // the older form does not test trailing triangles
        if (opt.raster == 'a') {
            const float bx = s->raster_dx * (float)(p->width * p->grid) * 0.5f / aspect / 8.0f + 2.0f;
            const float by = s->raster_dy * (float)(p->height * p->grid) * 0.5f / 8.0f + 2.0f;
            raster = bx * by <= 16384.0f;   // a block no sample of which passes the half-planes costs ~25 instructions
        }
        // Its work grows with the summed area of the triangles' boxes (every layer of depth is
        // tested), the tree walk's with the number of samples. Projection is taken while its
        // estimate stays at or below 160 sample slots per sample of the frame; measured camera
        // phases, projection by screen tile against tree walk: 1 M triangles at 3840 x 2160
        // (estimate 29) 0.82 / 3.13 ms, at 960 x 540 (69) 0.36 / 0.56, at 640 x 360 (104) 0.47 / 0.76;
        // 2 M at 1920 x 1080 (83) 0.59 / 1.26; 3 M at 3840 x 2160 (92) 1.73 / 2.85; 5 M (152) 2.68 /
        // 3.03. (The older form with atomic minima broke even near 64; 10 M triangles, estimate 291,
        // took it 13.2 ms against the walk's 5.2.)
        const double sx = (double)(p->width * p->grid) * 0.5 / aspect / 8.0, sy = (double)(p->height * p->grid) * 0.5 / 8.0;
        // Blocks of 8 x 8 samples: a box of w samples takes about w / 8 + 0.7 blocks a side.
        const double blocks = sx * sy * s->raster_sum_area + 0.7 * sx * s->raster_sum_dx + 0.7 * sy * s->raster_sum_dy +
                              0.49 * (double)s->n_tris_total;
        const double ratio = blocks * 64.0 / ((double)p->width * p->height * spp);
        const double limit = opt.raster_ratio >= 0.0 ? opt.raster_ratio : (opt.raster == 'a' ? 64.0 : 160.0);
        if (opt.verbose)
            std::fprintf(stderr, "rays_hip: projection would test %.1f sample slots per sample (limit %.1f)\n", ratio, limit);
        raster = raster && ratio <= limit;
    }
    const uint32_t n_packed = s->n_tris_total;
    if (projected) *projected = raster;
    const uint32_t total_tiles = dp.n_tiles;
    HIP_TRY(mark(0));
    // The library's own sort moves the keys that carry a shadow ray only and leaves their number on
    // the device; RAYS_HIP_SORT=rocprim sorts all of them with the library sort instead.
    const bool own_sort = !opt.sort_rocprim;
    // Shadow rays by light-space cells when the scene's triangles spread thinly enough over the
    // light-space grid of the sort (decided on the device, from the lists' lengths; the tree walk
    // takes the launch otherwise). Not worth the lists for small scenes, nor for scenes whose
    // lists are bound to be long: a triangle whose box has mean sides (a, b) across the light goes to
    // about (1 + a / cell) (1 + b / cell) cells, cell = the scene's side / cells a side (with the
    // cells a quarter of a triangle's size, as for ten million triangles in the shards of an 8K frame,
    // that is fifteen lists per triangle, and building them only to find them too long costs a shard
    // a quarter of its time). The lists depend on the scene and the sun only; they are built on a
    // stream of their own beside the camera rays.
    bool cells = false;
    uint32_t lc_cap_shift = 8, lc_mean = 0;
    if (mode == 0 && own_sort) {
        const LightProj lp = light_projection(s, p, (size_t)dp.n_tiles * tile_slots, true);
        const size_t lc_cells = (size_t)1 << lp.key_bits;
        // RAYS_HIP_SHADOW=walk: never; =cells: whenever no list outgrows its room, however long they are on average.
        const bool force = opt.shadow == 'c';
        const uint32_t mean_limit = cells_mean_limit(s);
        const double mean_list = cells_mean_list(s, lp, lp.key_bits);
        cells = lp.uscale > 0.0f && lp.vscale > 0.0f && lp.key_bits <= 18 && s->objects.size() <= 32 &&
                s->n_tris_total > 0 && s->n_tris_total < (1u << 26) && opt.shadow != 'w' &&
                (force || (s->n_tris_total >= 32768 && mean_list <= 1.4 * (double)mean_limit));   // the estimate runs 1.4 x high on the soups
        // Room per cell, 24 bytes an entry: the power of two from three times the estimated mean list up
        // (the longest list of the soups is about twice their mean; a scene that is denser than that in
        // places overflows a list and takes the tree walk), between 64 and 512 entries at 2^16 cells
        // (50 to 805 MB), 1024 when forced; at most 256 at 2^18.
        {
            const uint32_t top = lp.key_bits <= 16 ? (force ? 10u : 9u) : 8u;
            lc_cap_shift = 6u;
            while (lc_cap_shift < top && (double)(1u << lc_cap_shift) < 3.0 * mean_list) ++lc_cap_shift;
            if (force) lc_cap_shift = top;
        }
        if (cells) {
            if (w.lc_cells < lc_cells) {
                if (w.lc_counts) (void)hipFree(w.lc_counts);
                w.lc_counts = nullptr;
                w.lc_cells = 0;
                HIP_TRY(hipMalloc((void**)&w.lc_counts, (lc_cells + 1) * 4));
                w.lc_cells = lc_cells;
            }
            const size_t want = lc_cells << lc_cap_shift;
            if (w.lc_capacity < want) {
                if (w.lc_lists) (void)hipFree(w.lc_lists);
                w.lc_lists = nullptr;
                w.lc_capacity = 0;
                // the lists are read four entries at a time: room for a read that starts at the last one
                HIP_TRY(hipMalloc((void**)&w.lc_lists, (want + 4) * 24));   // patch, id, depth
                w.lc_capacity = want;
            }
            if (!w.lc_flags) HIP_TRY(hipMalloc((void**)&w.lc_flags, 256));
            if (!s->side_stream[set]) {
                HIP_TRY(hipStreamCreateWithFlags(&s->side_stream[set], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&s->side_fork[set], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&s->side_done[set], hipEventDisableTiming));
            }
            lc_mean = force ? 0xFFFFFFFFu : mean_limit;
        }
    }
    // Starts the list building on the side stream, after what `ws` holds so far (the scene and the sun
    // are all the lists depend on).
    auto start_lists = [&]() -> int {
        const LightProj lp = light_projection(s, p, (size_t)total_tiles * tile_slots, true);
        // After whatever used this set's lists last (the same stream `ws`, a frame or two ago).
        HIP_TRY(hipEventRecord(s->side_fork[set], ws));
        HIP_TRY(hipStreamWaitEvent(s->side_stream[set], s->side_fork[set], 0));
        HIP_TRY(rtk::lcell_lists(s->view, dp, lp, s->n_tris_total, w.lc_counts, w.lc_lists, lc_cap_shift, lc_mean,
                                 w.lc_flags, w.keys[0], opt.cells_redo_all, s->side_stream[set]));
        HIP_TRY(hipEventRecord(s->side_done[set], s->side_stream[set]));
        return RT_OK;
    };
    // Projection runs tile by tile (triangle lists per tile of 32 x 32 samples, nearest hits in LDS);
    // RAYS_HIP_RASTER=atomic keeps the older form with one nearest-hit word per sample in device memory.
    const bool raster_by_tile = raster && opt.raster != 'a';
    if (raster) {
        const size_t tab = (1 + 2 * s->objects.size()) * ((size_t)p->width + p->height) * p->grid;
        if (w.cam_table_floats < tab) {
            if (w.cam_table) (void)hipFree(w.cam_table);
            w.cam_table = nullptr;
            w.cam_table_floats = 0;
            HIP_TRY(hipMalloc((void**)&w.cam_table, tab * 4));
            w.cam_table_floats = tab;
        }
    }
    if (raster_by_tile) {
        const uint32_t n_rt = ceil_div(p->width * p->grid, 32u) * ceil_div(p->height * p->grid, 32u);
        if (w.bin_tiles < n_rt) {
            if (w.bin_counts) (void)hipFree(w.bin_counts);
            if (w.bin_offsets) (void)hipFree(w.bin_offsets);
            w.bin_counts = w.bin_offsets = nullptr;
            w.bin_tiles = 0;
            HIP_TRY(hipMalloc((void**)&w.bin_counts, ((size_t)n_rt + 1) * 4));
            HIP_TRY(hipMalloc((void**)&w.bin_offsets, ((size_t)n_rt + 1) * 4));
            w.bin_tiles = n_rt;
        }
        if (w.bin_recs_tris < n_packed) {
            if (w.bin_recs) (void)hipFree(w.bin_recs);
            w.bin_recs = nullptr;
            w.bin_recs_tris = 0;
            HIP_TRY(hipMalloc((void**)&w.bin_recs, (size_t)n_packed * 96));
            w.bin_recs_tris = n_packed;
        }
        HIP_TRY(rtk::raster_bin(s->view, dp, n_packed, w.cam_table, n_rt, w.bin_counts, w.bin_offsets, w.bin_list,
                                (uint32_t)w.bin_capacity, w.bin_recs, false, ws));
        const uint32_t key[5] = {p->width, p->height, p->grid, p->shard_index, p->shard_count};
        if (!w.bin_list || std::memcmp(key, w.bin_key, sizeof(key)) != 0) {
            // First frame of this shape: the lists' total length, read back once.
            uint32_t total = 0;
            HIP_TRY(hipStreamSynchronize(ws));
            HIP_TRY(hipMemcpy(&total, w.bin_offsets + n_rt, 4, hipMemcpyDeviceToHost));
            if (w.bin_capacity < total || !w.bin_list) {
                if (w.bin_list) (void)hipFree(w.bin_list);
                w.bin_list = nullptr;
                w.bin_capacity = 0;
                HIP_TRY(hipMalloc((void**)&w.bin_list, std::max<size_t>(total, 4) * 4));
                w.bin_capacity = std::max<size_t>(total, 4);
            }
            std::memcpy(w.bin_key, key, sizeof(key));
        }
        HIP_TRY(rtk::raster_bin(s->view, dp, n_packed, w.cam_table, n_rt, w.bin_counts, w.bin_offsets, w.bin_list,
                                (uint32_t)w.bin_capacity, w.bin_recs, true, ws));
    } else if (raster) {
        const size_t words = (size_t)p->width * p->height * spp;
        if (w.zbuf_words < words) {
            if (w.zbuf) (void)hipFree(w.zbuf);
            w.zbuf = nullptr;
            w.zbuf_words = 0;
            HIP_TRY(hipMalloc((void**)&w.zbuf, words * 8));
            w.zbuf_words = words;
            w.zbuf_clean = false;
        }
        const bool clear = !w.zbuf_clean;
        w.zbuf_clean = false;  // until the record pass has been through every tile of this frame
        HIP_TRY(rtk::raster_depth(s->view, dp, n_packed, w.cam_table, w.zbuf, clear, d_stats, ws));
    }
    // Beside the camera rays' main kernel, which is bound by arithmetic, not beside their binning, which
    // like the list building is bound by atomics (measured: 0.07 ms dearer camera rays here, 0.15 ms there).
    if (lists_built) *lists_built = cells;
    if (cells) {
        rc = start_lists();
        if (rc != RT_OK) return rc;
    }
    for (uint32_t first = 0; first < total_tiles; first += run_tiles) {
        const bool last_run = first + run_tiles >= total_tiles;
        dp.tile_first = first;
        dp.n_tiles = std::min(run_tiles, total_tiles - first);
        const size_t n_slots = (size_t)dp.n_tiles * tile_slots;
        const LightProj lp = light_projection(s, p, n_slots, cells);
        if (first == 0) HIP_TRY(mark(7));   // set-up and binning done: the camera rays' main kernel is next
        if (raster_by_tile) {
            HIP_TRY(rtk::raster_tiles(s->view, dp, lp, d_list, w.cam_table, w.bin_offsets, w.bin_list, w.bin_recs, w.samp,
                                      w.state, w.keys[0], d_stats, ws));
        } else if (raster) {
            HIP_TRY(rtk::raster_records(s->view, dp, lp, d_list, w.zbuf, w.samp, w.state, w.keys[0], d_stats, ws));
        } else {
            HIP_TRY(rtk::wf_primary(s->view, dp, lp, d_list, w.samp, w.state, w.keys[0], d_stats, ws));
        }
        if (last_run) HIP_TRY(mark(1));
        if (primary_stats && d_stats && last_run) {
            HIP_TRY(hipStreamSynchronize(ws));
            HIP_TRY(hipMemcpy(primary_stats, d_stats, sizeof(DevStats), hipMemcpyDeviceToHost));
        }
        if (own_sort) {
            HIP_TRY(rtk::radix_sort_valid(w.radix_work, w.keys[0], w.keys[1], w.order, n_slots, lp.key_bits, w.n_valid, ws));
        } else {
            size_t tb = w.temp_bytes;
            HIP_TRY(rtk::sort_pairs_u32(w.temp, &tb, w.keys[0], w.keys[1], nullptr, w.order, n_slots, lp.key_bits, ws));
        }
        if (last_run) HIP_TRY(mark(2));
        if (cells) {
            HIP_TRY(hipStreamWaitEvent(ws, s->side_done[set], 0));
            HIP_TRY(rtk::shadow_cells(s->view, dp, lp, (uint32_t)n_slots, w.keys[1], w.order, w.samp, w.state, d_stats,
                                      w.n_valid, w.lc_counts, w.lc_lists, lc_cap_shift, w.lc_flags, w.keys[0], ws));   // the sort is done with its input keys
        }
        HIP_TRY(rtk::wf_shadow(s->view, dp, (uint32_t)n_slots, lp.key_bits > 16 ? lp.key_bits - 16 : 0u, w.keys[1],
                               w.order, w.samp, w.state, d_stats, ws, own_sort ? w.n_valid : nullptr,
                               cells ? w.lc_flags : nullptr));
        if (mode == 0) {
            if (last_run) HIP_TRY(mark(3));
            if (in_flight) {
                HIP_TRY(hipEventRecord(s->pipe_ready[set], ws));
                HIP_TRY(hipStreamWaitEvent(stream, s->pipe_ready[set], 0));
            }
            HIP_TRY(rtk::wf_resolve(s->view, dp, d_list, w.samp, w.state, d_rgba8, d_hit_ids, stream));
        } else {
            tr.level_rays = (uint32_t)(n_slots * tr.branches);
            const size_t n_rays = (size_t)tr.level_rays * tr.levels;
            const LightProj lp2 = light_projection(s, p, n_rays);
            // Deeper levels only hold the rays that bounced off a mirror; the rest of their records
            // says "no ray" from the start. The lists of those rays use the sort's output array.
            HIP_TRY(hipMemsetAsync(bw.counts, 0, (RT_MAX_REFLECT + 1) * 4, stream));
            if (tr.levels > 1) {
                const size_t deeper = n_rays - tr.level_rays;
                HIP_TRY(hipMemsetAsync(bw.state + tr.level_rays, 0, deeper * 4, stream));
                HIP_TRY(hipMemsetAsync(bw.keys[0] + tr.level_rays, 0xFF, deeper * 4, stream));
            }
            if (last_run) HIP_TRY(mark(5));
            DevStats before_bounce{};
            if (bounce_stats && d_stats) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipMemcpy(&before_bounce, d_stats, sizeof(DevStats), hipMemcpyDeviceToHost));
            }
            for (uint32_t level = 0; level < tr.levels; ++level) {
                uint32_t* counter = nullptr;
                rc = take_counter(s->ctx, stream, &counter);
                if (rc != RT_OK) return rc;
                HIP_TRY(rtk::wf_bounce(s->view, dp, lp2, tr, level, d_list, counter, w.samp, w.state, bw.samp, bw.state,
                                       bw.keys[0], bw.order, bw.counts, d_stats, stream));
            }
            if (last_run) HIP_TRY(mark(6));
            if (bounce_stats && d_stats) {
                // What the bounce launches of this run of tiles added to the counters.
                DevStats after{};
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipMemcpy(&after, d_stats, sizeof(DevStats), hipMemcpyDeviceToHost));
                bounce_stats->rays += after.rays - before_bounce.rays;
                bounce_stats->box_tests += after.box_tests - before_bounce.box_tests;
                bounce_stats->tri_tests += after.tri_tests - before_bounce.tri_tests;
                bounce_stats->node_visits += after.node_visits - before_bounce.node_visits;
                bounce_stats->fetch_bytes += after.fetch_bytes - before_bounce.fetch_bytes;
            }
            if (own_sort) {
                HIP_TRY(rtk::radix_sort_valid(bw.radix_work, bw.keys[0], bw.keys[1], bw.order, n_rays, lp2.key_bits,
                                              bw.n_valid, stream));
            } else {
                size_t tb2 = bw.temp_bytes;
                HIP_TRY(rtk::sort_pairs_u32(bw.temp, &tb2, bw.keys[0], bw.keys[1], nullptr, bw.order, n_rays,
                                            lp2.key_bits, stream));
            }
            HIP_TRY(rtk::wf_shadow(s->view, dp, (uint32_t)n_rays, lp2.key_bits > 16 ? lp2.key_bits - 16 : 0u,
                                   bw.keys[1], bw.order, bw.samp, bw.state, d_stats, stream,
                                   own_sort ? bw.n_valid : nullptr));
            if (last_run) HIP_TRY(mark(3));
            HIP_TRY(rtk::wf_resolve_tree(s->view, dp, tr, d_list, w.samp, w.state, bw.samp, bw.state, d_rgba8,
                                         d_hit_ids, stream));
        }
    }
    HIP_TRY(mark(4));
    if (mode != 2) {
        HIP_TRY(mark(5));
        HIP_TRY(mark(6));
    }
    if (raster && !raster_by_tile) w.zbuf_clean = true;
    HIP_TRY(hipEventRecord(s->pipe_free[set], stream));
    s->pipe_free_set[set] = true;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }
const char* rt_version(void) { return "rays-hip 0.1 (gfx950)"; }

int rt_ctx_create(const int* devices, int n, rt_ctx** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "out is null");
    *out = nullptr;
    if (int g = rt_runtime_guard()) return g;
    if (!devices || n != 1) return rt_fail(RT_ERR_INVALID, "exactly one device per context (one process per GPU)");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (devices[0] < 0 || devices[0] >= count)
        return rt_fail(RT_ERR_DEVICE, "device %d not present (%d visible)", devices[0], count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, devices[0]));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rt_fail(RT_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", devices[0],
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(devices[0]));
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return rt_fail(RT_ERR_NOMEM, "out of host memory");
    c->device = devices[0];
    c->opt.read_env();
    // The allocation-free entry point's scratch array: its address on this device, looked up once.
    const int rc = rt_noalloc_setup(c);
    if (rc != RT_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return RT_OK;
}

int rt_ctx_reload_env(rt_ctx* c) {
    if (!c) return rt_fail(RT_ERR_INVALID, "context is null");
    std::lock_guard<std::mutex> lock(c->mu);
    c->opt.read_env();
    return RT_OK;
}

void rt_ctx_destroy(rt_ctx* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        if (c->live_scenes > 0) {
            // Scenes hold a pointer to their context: the last rt_scene_destroy deletes it.
            c->destroy_requested = true;
            return;
        }
    }
    (void)hipSetDevice(c->device);
    delete c;
}

int rt_scene_build(rt_ctx* ctx, const rt_object_desc* objs, uint32_t n_objs, rt_scene** out) {
    return rt_scene_build_ex(ctx, objs, n_objs, 0u, out);
}

int rt_scene_build_ex(rt_ctx* ctx, const rt_object_desc* objs, uint32_t n_objs, uint32_t flags, rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "out is null");
    if (flags & ~(uint32_t)(RT_BUILD_DEVICE | RT_BUILD_HOST)) return rt_fail(RT_ERR_INVALID, "unknown build flags %#x", flags);
    if ((flags & RT_BUILD_DEVICE) && (flags & RT_BUILD_HOST)) return rt_fail(RT_ERR_INVALID, "RT_BUILD_DEVICE and RT_BUILD_HOST together");
    *out = nullptr;
    if (!ctx || !objs || n_objs == 0) return rt_fail(RT_ERR_INVALID, "ctx / objs null or no objects");
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));

    // ---- validate, and lay all objects out in shared arrays --------------------------------
    uint64_t total_tris = 0, total_verts = 0;
    for (uint32_t o = 0; o < n_objs; ++o) {
        const rt_object_desc& d = objs[o];
        if (!d.verts || d.n_verts == 0) return rt_fail(RT_ERR_INVALID, "object %u has no vertices", o);
        if (d.n_tris > 0 && (!d.tris || !d.tri_data)) return rt_fail(RT_ERR_INVALID, "object %u: tris / tri_data null", o);
        for (uint64_t i = 0; i < (uint64_t)d.n_tris * 3; ++i)
            if (d.tris[i] >= d.n_verts)
                return rt_fail(RT_ERR_INVALID, "object %u: vertex index %u out of range (%u vertices)", o, d.tris[i],
                            d.n_verts);
        total_tris += d.n_tris;
        total_verts += d.n_verts;
    }
    if (total_tris >= (1ull << 27)) return rt_fail(RT_ERR_UNSUPPORTED, "more than 2^27 triangles");
    if (total_verts >= (1ull << 32)) return rt_fail(RT_ERR_UNSUPPORTED, "more than 2^32 vertices");

    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return rt_fail(RT_ERR_NOMEM, "out of host memory");
    struct Guard {
        rt_scene* s;
        ~Guard() { delete s; }
    } guard{s};
    s->ctx = ctx;
    s->n_tris = (uint32_t)total_tris;
    s->objects.resize(n_objs);
    s->order.resize(n_objs);

    std::vector<float> verts_all((size_t)total_verts * 3);
    std::vector<uint32_t> tris_all((size_t)total_tris * 3);  // sorted, vertex indices offset per object
    std::vector<float> data_all((size_t)total_tris * 6);     // sorted
    std::vector<uint32_t> leaf_first;                        // global sorted index of each leaf's first triangle
    std::vector<float> rest_verts;
    std::vector<float> bounds;                               // per packed triangle, for the tree build

    std::vector<float> tree_box((size_t)n_objs * 6);         // per object: finite bounds of its vertices
    uint32_t tri_base = 0, vert_base = 0;
    for (uint32_t o = 0; o < n_objs; ++o) {
        const rt_object_desc& d = objs[o];
        DevObject& ob = s->objects[o];
        std::memcpy(&verts_all[(size_t)vert_base * 3], d.verts, (size_t)d.n_verts * 12);
        // Object::from, src/objects.rs:48-53.
        float mn[3] = {d.verts[0], d.verts[1], d.verts[2]}, mx[3] = {d.verts[0], d.verts[1], d.verts[2]};
        for (uint32_t i = 0; i < d.n_verts; ++i)
            for (int k = 0; k < 3; ++k) {
                mn[k] = g_min(mn[k], d.verts[3 * (size_t)i + k]);
                mx[k] = g_max(mx[k], d.verts[3 * (size_t)i + k]);
            }
        std::memcpy(ob.bbmin, mn, 12);
        std::memcpy(ob.bbmax, mx, 12);
        // The same over the finite coordinates only: what the device tree build scales its Morton codes and
        // its padding by (the trees hold finite boxes whatever the vertices are, see the host build below).
        {
            float* f = &tree_box[6 * (size_t)o];
            for (int k = 0; k < 3; ++k) f[k] = INFINITY, f[3 + k] = -INFINITY;
            for (uint32_t i = 0; i < d.n_verts; ++i)
                for (int k = 0; k < 3; ++k) {
                    const float v = d.verts[3 * (size_t)i + k];
                    if (std::isfinite(v)) f[k] = std::min(f[k], v), f[3 + k] = std::max(f[3 + k], v);
                }
            for (int k = 0; k < 3; ++k)
                if (!(f[k] <= f[3 + k])) f[k] = f[3 + k] = 0.0f;
            // the point that stands in for a triangle with a vertex that is not finite
            for (int k = 0; k < 3; ++k) f[k] = std::min(f[k], 0.0f), f[3 + k] = std::max(f[3 + k], 0.0f);
        }
        // optimize_model, src/opt.rs:11-36: stable sort, smallest vertex index ascending, then
        // largest vertex index descending.
        std::vector<uint32_t>& ord = s->order[o];
        ord.resize(d.n_tris);
        for (uint32_t i = 0; i < d.n_tris; ++i) ord[i] = i;
        const uint32_t* T = d.tris;
        std::stable_sort(ord.begin(), ord.end(), [T](uint32_t a, uint32_t b) {
            const uint32_t *ta = T + 3 * (size_t)a, *tb = T + 3 * (size_t)b;
            const uint32_t amin = std::min(std::min(ta[0], ta[1]), ta[2]);
            const uint32_t bmin = std::min(std::min(tb[0], tb[1]), tb[2]);
            if (amin != bmin) return amin < bmin;
            const uint32_t amax = std::max(std::max(ta[0], ta[1]), ta[2]);
            const uint32_t bmax = std::max(std::max(tb[0], tb[1]), tb[2]);
            return amax > bmax;
        });
        for (uint32_t i = 0; i < d.n_tris; ++i) {
            const uint32_t src = ord[i];
            for (int k = 0; k < 3; ++k) tris_all[3 * (size_t)(tri_base + i) + k] = T[3 * (size_t)src + k] + vert_base;
            std::memcpy(&data_all[6 * (size_t)(tri_base + i)], d.tri_data + 6 * (size_t)src, 24);
        }
        ob.tri_base = tri_base;
        ob.n_tris = d.n_tris;
        const uint32_t n_packed = (d.n_tris / 8) * 8;
        ob.wide_root = 0;
        ob.n_rest = d.n_tris - n_packed;
        ob.rest_base = (uint32_t)(rest_verts.size() / 9);
        ob.leaf_base = (uint32_t)leaf_first.size();
        ob.albedo = d.albedo_over_pi;
        ob.reflectivity = d.reflectivity;
        ob.roughness = d.roughness;
        ob.root = rtbvh::EMPTY;
        // bit 31: one of the last n_leaves % 8 leaves, which stay out of the packets of eight (src/opt.rs:64-72)
        for (uint32_t l = 0; l < n_packed / 8; ++l)
            leaf_first.push_back((tri_base + l * 8) | (l < ((n_packed / 8) & ~7u) ? 0u : 0x80000000u));
        for (uint32_t i = n_packed; i < d.n_tris; ++i)
            for (int k = 0; k < 3; ++k) {
                const float* v = &verts_all[3 * (size_t)tris_all[3 * (size_t)(tri_base + i) + k]];
                rest_verts.insert(rest_verts.end(), v, v + 3);
            }
        tri_base += d.n_tris;
        vert_base += d.n_verts;
    }
    s->n_leaves = (uint32_t)leaf_first.size();
    s->n_tris_total = tri_base;
    s->raster_ok = n_objs <= 16;
    {
        // Trailing n % 8 triangles never lower `closest` (src/main.rs:157-169), so what they are compared
        // with depends on the objects before them. Projection takes the scenes in which that is the
        // nearest eight-wide hit of the whole scene: no eight-wide triangle after a trailing one in object
        // order (the reference's own scene: two teapots of whole leaves, then a floor of two triangles).
        uint32_t first_rest = n_objs, last_packed = 0;
        s->n_rest_total = 0;
        for (uint32_t o = 0; o < n_objs; ++o) {
            const DevObject& ob = s->objects[o];
            if (ob.n_rest != 0 && first_rest == n_objs) first_rest = o;
            if (ob.n_tris > ob.n_rest) last_packed = o;
            s->n_rest_total += ob.n_rest;
        }
        if (first_rest < last_packed) s->raster_ok = false;
        if (s->n_rest_total == s->n_tris_total) s->raster_ok = false;   // nothing to project
    }
    for (uint32_t o = 0; o < n_objs && s->raster_ok; ++o) {
        const DevObject& ob = s->objects[o];
        for (uint32_t i = 0; i < ob.n_tris - ob.n_rest && s->raster_ok; ++i) {
            const uint32_t* t = &tris_all[3 * (size_t)(ob.tri_base + i)];
            const float *a = &verts_all[3 * (size_t)t[0]], *b = &verts_all[3 * (size_t)t[1]], *c = &verts_all[3 * (size_t)t[2]];
            float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
            for (const float* v : {a, b, c}) {
                if (!(v[2] > 0.01f)) s->raster_ok = false;
                const float px = v[0] / v[2], py = v[1] / v[2];
                x0 = std::min(x0, px), x1 = std::max(x1, px);
                y0 = std::min(y0, py), y1 = std::max(y1, py);
            }
            // The projection kernel grows every triangle's box of samples by a bound on what the
            // exact test's rounding can accept outside the projected triangle (raster_setup in
            // rt_kernels.hip); the same bound here, so that the work estimate and the block limit
            // see the boxes the kernel will step over. A triangle seen edge-on from the camera has
            // no finite bound: such scenes take the tree walk.
            double ex = (double)x1 - x0, ey = (double)y1 - y0;
            const double grow = raster_grow(a, b, c);
            if (grow < 1.0e3) {
                ex *= 1.0 + 2.0 * grow;
                ey *= 1.0 + 2.0 * grow;
            } else {
                ex = ey = INFINITY;
            }
            // The kernel also keeps a triangle's samples inside the projection of its reference leaf
            // (the eight consecutive triangles it is packed with), which bounds edge-on triangles.
            const uint32_t leaf = i >> 3, n_leaves = ob.n_tris >> 3;
            if (leaf < (n_leaves & ~7u)) {
                float lx0 = INFINITY, lx1 = -INFINITY, ly0 = INFINITY, ly1 = -INFINITY;
                for (uint32_t j = leaf * 8; j < leaf * 8 + 8; ++j)
                    for (int k = 0; k < 3; ++k) {
                        const float* v = &verts_all[3 * (size_t)tris_all[3 * (size_t)(ob.tri_base + j) + k]];
                        // corners of the leaf's box project no farther out than its vertices taken at
                        // the leaf's nearest depth; the estimate only needs the extent
                        lx0 = std::min(lx0, v[0] / v[2]), lx1 = std::max(lx1, v[0] / v[2]);
                        ly0 = std::min(ly0, v[1] / v[2]), ly1 = std::max(ly1, v[1] / v[2]);
                    }
                ex = std::min(ex, 1.5 * ((double)lx1 - lx0));
                ey = std::min(ey, 1.5 * ((double)ly1 - ly0));
            }
            if (!(ex < INFINITY) || !(ey < INFINITY)) {
                s->raster_ok = false;
                break;
            }
            // the kernels clip a box to the frame: |dx| <= aspect <= 4, |dy| <= 1
            const float dx = (float)std::min(ex, 8.0), dy = (float)std::min(ey, 2.0);
            s->raster_dx = std::max(s->raster_dx, dx);
            s->raster_dy = std::max(s->raster_dy, dy);
            s->raster_sum_dx += dx;
            s->raster_sum_dy += dy;
            s->raster_sum_area += (double)dx * dy;
        }
    }
    {
        double sum[3] = {0.0, 0.0, 0.0};
        for (size_t i = 0; i < (size_t)s->n_tris_total; ++i) {
            const uint32_t* t = &tris_all[3 * i];
            const float *a = &verts_all[3 * (size_t)t[0]], *b = &verts_all[3 * (size_t)t[1]], *c = &verts_all[3 * (size_t)t[2]];
            for (int k = 0; k < 3; ++k) {
                const float side = std::max(a[k], std::max(b[k], c[k])) - std::min(a[k], std::min(b[k], c[k]));
                if (std::isfinite(side)) sum[k] += side;
            }
        }
        for (int k = 0; k < 3; ++k) s->mean_tri_side[k] = s->n_tris_total ? sum[k] / s->n_tris_total : 0.0;
    }

    // ---- trees: one per object, over its 8-wide triangles ----------------------------------
    std::vector<rtbvh::Node4> nodes_all;
    std::vector<rtbvh::Child> wide_all;
    std::vector<uint32_t> order_all;  // record -> global triangle id
    const int nt = host_threads();
    uint32_t wide_width = 64;
    if (const char* e = std::getenv("RAYS_HIP_WIDE")) wide_width = (uint32_t)std::atoi(e);
    // Where the trees are built. The host's surface-area build gives the walks 4 to 12 % faster trees on
    // scenes of one to three million triangles; from about ten million on the device's radix trees are as
    // fast to walk (measured on triangle soups, the camera, shadow and scatter walks alike) and are built in
    // under a second where the host needs several. So: on the device from DEVICE_BUILD_FROM triangles on,
    // unless the caller says which (RT_BUILD_DEVICE / RT_BUILD_HOST, or RAYS_HIP_BUILD=device|host).
    constexpr uint32_t DEVICE_BUILD_FROM = 4000000u;
    uint32_t n_in_trees = 0;
    for (uint32_t o = 0; o < n_objs; ++o) n_in_trees += (s->objects[o].n_tris / 8) * 8;
    bool device_build = n_in_trees >= DEVICE_BUILD_FROM;
    if (const char* e = std::getenv("RAYS_HIP_BUILD")) {
        if (std::strcmp(e, "device") == 0) device_build = true;
        if (std::strcmp(e, "host") == 0) device_build = false;
    }
    if (flags & RT_BUILD_DEVICE) device_build = true;
    if (flags & RT_BUILD_HOST) device_build = false;
    s->built_on_device = device_build;
    struct TreeSlot {
        uint32_t node_base = 0, wide_base = 0, rec_base = 0, n = 0;
    };
    std::vector<TreeSlot> slots(n_objs);
    size_t n_nodes_total = 0, n_wide_total = 0, n_recs_total = 0;
    uint32_t n_tree_max = 0;
    // Two device builders: cuts where the sorted Morton codes first differ (a binary radix tree,
    // collapsed top-down; the default) or every 4 / 64 records of the Morton curve (RAYS_HIP_LBVH=chunk,
    // and the fallback when the radix tree comes out too deep). The first needs room for n 4-wide
    // nodes and 2 n wide child records per object; what it uses is known after the build.
    bool chunked = false;
    if (const char* e = std::getenv("RAYS_HIP_LBVH")) chunked = std::strcmp(e, "chunk") == 0;
    const uint32_t fan = std::min(std::max(wide_width, 2u), rtbvh::MAX_WIDE);
    for (uint32_t o = 0; device_build && o < n_objs; ++o) {
        const uint32_t n_packed = (s->objects[o].n_tris / 8) * 8;
        if (n_packed == 0) continue;
        const rtk::LbvhLayout L = rtk::lbvh_layout(n_packed, fan);
        slots[o].node_base = (uint32_t)n_nodes_total;
        slots[o].wide_base = (uint32_t)n_wide_total;
        slots[o].rec_base = (uint32_t)n_recs_total;
        slots[o].n = n_packed;
        n_nodes_total += chunked ? L.n_nodes4 : n_packed;
        n_wide_total += chunked ? L.n_wide : 2 * (size_t)n_packed;
        n_recs_total += n_packed;
        n_tree_max = std::max(n_tree_max, n_packed);
        if (chunked) {
            s->max_stack = std::max(s->max_stack, 3u * L.depth4 + 1u);
            s->wide_depth = std::max(s->wide_depth, L.depth_wide);
        }
        // The reservation fixes the child indices the device builder hands out, so it is what must
        // fit an inner reference (the host build below checks the records it actually made).
        if (!rtbvh::wide_children_fit(n_wide_total))
            return rt_fail(RT_ERR_UNSUPPORTED, "wide tree too large for the device build (%zu child records reserved, limit %llu)",
                           n_wide_total, (unsigned long long)rtbvh::MAX_WIDE_CHILDREN);
    }
    if (s->max_stack > RT_STACK_MAX || s->wide_depth > RT_BEAM_LEVELS) return rt_fail(RT_ERR_UNSUPPORTED, "tree too deep");
    for (uint32_t o = 0; !device_build && o < n_objs; ++o) {
        DevObject& ob = s->objects[o];
        const uint32_t n_packed = (ob.n_tris / 8) * 8;
        if (n_packed == 0) continue;
        bounds.resize((size_t)n_packed * 6);
        for (uint32_t i = 0; i < n_packed; ++i) {
            const uint32_t* t = &tris_all[3 * (size_t)(ob.tri_base + i)];
            const float *a = &verts_all[3 * (size_t)t[0]], *b = &verts_all[3 * (size_t)t[1]],
                        *c = &verts_all[3 * (size_t)t[2]];
            bool finite = true;
            for (int k = 0; k < 3; ++k) {
                bounds[6 * (size_t)i + k] = std::min(std::min(a[k], b[k]), c[k]);
                bounds[6 * (size_t)i + 3 + k] = std::max(std::max(a[k], b[k]), c[k]);
                finite = finite && std::isfinite(a[k]) && std::isfinite(b[k]) && std::isfinite(c[k]);
            }
            // A triangle with a vertex that is not finite can be hit by no ray (its test yields infinities
            // and NaNs the arithmetic made, which are no hit in the reference either); it gets a point for
            // a box, so that the boxes above it stay finite and its neighbours in the tree stay reachable.
            if (!finite)
                for (int k = 0; k < 6; ++k) bounds[6 * (size_t)i + k] = 0.0f;
        }
        rtbvh::Result res;
        const uint32_t node_base = (uint32_t)nodes_all.size(), rec_base = (uint32_t)order_all.size();
        const uint32_t wide_base = (uint32_t)wide_all.size();
        rtbvh::build(bounds.data(), n_packed, node_base, rec_base, wide_width, wide_base, nt, false, &res);
        if (res.max_stack > RT_STACK_MAX || res.wide_depth > RT_BEAM_LEVELS)
            rtbvh::build(bounds.data(), n_packed, node_base, rec_base, wide_width, wide_base, nt, true, &res);
        if (res.max_stack > RT_STACK_MAX || res.wide_depth > RT_BEAM_LEVELS)
            return rt_fail(RT_ERR_UNSUPPORTED, "object %u: tree too deep (%u pending entries, %u wide levels)", o,
                        res.max_stack, res.wide_depth);
        if (!rtbvh::wide_children_fit(wide_all.size() + res.wide.size()))
            return rt_fail(RT_ERR_UNSUPPORTED, "wide tree too large");
        ob.wide_root = res.wide_root;
        wide_all.insert(wide_all.end(), res.wide.begin(), res.wide.end());
        s->max_stack = std::max(s->max_stack, res.max_stack);
        s->wide_depth = std::max(s->wide_depth, res.wide_depth);
        ob.root = node_base;
        nodes_all.insert(nodes_all.end(), res.nodes.begin(), res.nodes.end());
        for (uint32_t r : res.order) order_all.push_back(ob.tri_base + r);
    }
    if (!device_build) {
        n_nodes_total = nodes_all.size();
        n_wide_total = wide_all.size();
        n_recs_total = order_all.size();
    }
    s->n_nodes = (uint32_t)n_nodes_total;
    s->sizes[0] = n_wide_total * sizeof(rtbvh::Child);
    s->sizes[1] = n_nodes_total * sizeof(rtbvh::Node4);
    s->sizes[2] = n_recs_total * 48;
    s->sizes[3] = (uint64_t)s->n_tris * 24;
    {
        auto has_tree = [](const DevObject& ob) { return ob.n_tris >= 8u; };
        bool any_tree = false;
        for (const DevObject& ob : s->objects) any_tree = any_tree || has_tree(ob);
        bool first = true;
        for (const DevObject& ob : s->objects) {
            if (any_tree && !has_tree(ob)) continue;
            for (int k = 0; k < 3; ++k) {
                s->bounds_lo[k] = first ? ob.bbmin[k] : std::min(s->bounds_lo[k], ob.bbmin[k]);
                s->bounds_hi[k] = first ? ob.bbmax[k] : std::max(s->bounds_hi[k], ob.bbmax[k]);
            }
            first = false;
        }
    }
    if (std::getenv("RAYS_HIP_VERBOSE"))
        std::fprintf(stderr, "[rays_hip] %u tris, %zu nodes4 (stack bound %u), wide width %u: %zu children, depth %u%s\n",
                     s->n_tris, n_nodes_total, s->max_stack, wide_width, n_wide_total, s->wide_depth,
                     device_build ? " (device build)" : "");

    // ---- upload and pack on the device -----------------------------------------------------
    DevPool pool;
    float *d_verts, *d_tri9, *d_leaf_bounds, *d_data, *d_rest;
    uint32_t *d_tris, *d_leaf_first, *d_order;
    float4* d_recs;
    rtbvh::Node4* d_nodes;
    rtbvh::Child* d_wide;
    DevObject* d_objs;
    DevPool tmp;  // freed on return
    HIP_TRY(tmp.alloc((void**)&d_verts, verts_all.size() * 4));
    HIP_TRY(tmp.alloc((void**)&d_tris, tris_all.size() * 4));
    HIP_TRY(tmp.alloc((void**)&d_leaf_first, leaf_first.size() * 4));
    HIP_TRY(tmp.alloc((void**)&d_order, n_recs_total * 4));
    HIP_TRY(pool.alloc((void**)&d_tri9, (size_t)total_tris * 36));
    HIP_TRY(pool.alloc((void**)&d_leaf_bounds, leaf_first.size() * 24));
    float4* d_leaf_gate = nullptr;
    HIP_TRY(pool.alloc((void**)&d_leaf_gate, std::max<size_t>(leaf_first.size(), 1) * 32));
    HIP_TRY(pool.alloc((void**)&d_data, data_all.size() * 4));
    HIP_TRY(pool.alloc((void**)&d_rest, rest_verts.size() * 4));
    HIP_TRY(pool.alloc((void**)&d_recs, n_recs_total * 48 + 192));  // padded: a leaf is read as 192 B
    HIP_TRY(pool.alloc((void**)&d_nodes, n_nodes_total * sizeof(rtbvh::Node4)));
    HIP_TRY(pool.alloc((void**)&d_wide, n_wide_total * sizeof(rtbvh::Child)));
    float4* d_qnodes = nullptr;
    HIP_TRY(pool.alloc((void**)&d_qnodes, std::max<size_t>(n_nodes_total, 1) * 64));
    HIP_TRY(pool.alloc((void**)&d_objs, s->objects.size() * sizeof(DevObject)));
    HIP_TRY(hipMemcpy(d_verts, verts_all.data(), verts_all.size() * 4, hipMemcpyHostToDevice));
    if (!tris_all.empty()) HIP_TRY(hipMemcpy(d_tris, tris_all.data(), tris_all.size() * 4, hipMemcpyHostToDevice));
    if (!leaf_first.empty())
        HIP_TRY(hipMemcpy(d_leaf_first, leaf_first.data(), leaf_first.size() * 4, hipMemcpyHostToDevice));
    if (!order_all.empty())
        HIP_TRY(hipMemcpy(d_order, order_all.data(), order_all.size() * 4, hipMemcpyHostToDevice));
    if (!data_all.empty()) HIP_TRY(hipMemcpy(d_data, data_all.data(), data_all.size() * 4, hipMemcpyHostToDevice));
    if (!rest_verts.empty())
        HIP_TRY(hipMemcpy(d_rest, rest_verts.data(), rest_verts.size() * 4, hipMemcpyHostToDevice));
    if (!nodes_all.empty())
        HIP_TRY(hipMemcpy(d_nodes, nodes_all.data(), nodes_all.size() * sizeof(rtbvh::Node4), hipMemcpyHostToDevice));
    if (!wide_all.empty())
        HIP_TRY(hipMemcpy(d_wide, wide_all.data(), wide_all.size() * sizeof(rtbvh::Child), hipMemcpyHostToDevice));
    if (device_build && n_tree_max > 0) {
        uint32_t *keys[2], *vals[2];
        void* sort_tmp;
        size_t tb = 0;
        HIP_TRY(rtk::sort_pairs_u32(nullptr, &tb, nullptr, nullptr, nullptr, nullptr, n_tree_max, 32, 0));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(tmp.alloc((void**)&keys[k], (size_t)n_tree_max * 4));
            HIP_TRY(tmp.alloc((void**)&vals[k], (size_t)n_tree_max * 4));
        }
        HIP_TRY(tmp.alloc(&sort_tmp, std::max<size_t>(tb, 16)));
        void* work = nullptr;
        if (!chunked) HIP_TRY(tmp.alloc(&work, rtk::lbvh_radix_work_bytes(n_tree_max)));
        size_t used_nodes = 0, used_wide = 0;
        for (uint32_t o = 0; o < n_objs; ++o) {
            if (slots[o].n == 0) continue;
            DevObject& ob = s->objects[o];
            bool done = false;
            if (!chunked) {
                rtk::LbvhResult res{};
                HIP_TRY(rtk::lbvh_build_radix(d_verts, d_tris + 3 * (size_t)ob.tri_base, slots[o].n, &tree_box[6 * (size_t)o], &tree_box[6 * (size_t)o + 3],
                                              ob.tri_base, slots[o].rec_base, slots[o].node_base, slots[o].wide_base, fan,
                                              d_nodes, d_wide, d_order, keys, vals, sort_tmp, tb, work, &res, 0));
                if (3u * res.depth4 + 1u <= RT_STACK_MAX && res.depth_wide <= RT_BEAM_LEVELS) {
                    ob.root = res.root4;
                    ob.wide_root = res.wide_root;
                    s->max_stack = std::max(s->max_stack, 3u * res.depth4 + 1u);
                    s->wide_depth = std::max(s->wide_depth, res.depth_wide);
                    used_nodes += res.n_nodes4;
                    used_wide += res.n_wide;
                    done = true;
                }
            }
            if (!done) {
                const rtk::LbvhLayout L = rtk::lbvh_layout(slots[o].n, fan);
                HIP_TRY(rtk::lbvh_build(d_verts, d_tris + 3 * (size_t)ob.tri_base, slots[o].n, &tree_box[6 * (size_t)o], &tree_box[6 * (size_t)o + 3],
                                        ob.tri_base, slots[o].rec_base, slots[o].node_base, slots[o].wide_base, fan,
                                        d_nodes, d_wide, d_order, keys, vals, sort_tmp, tb, &ob.root, &ob.wide_root, 0));
                s->max_stack = std::max(s->max_stack, 3u * L.depth4 + 1u);
                s->wide_depth = std::max(s->wide_depth, L.depth_wide);
                used_nodes += L.n_nodes4;
                used_wide += L.n_wide;
            }
        }
        // The fixed fan-out fallback has a depth of its own (a small RAYS_HIP_WIDE makes it deep).
        if (s->max_stack > RT_STACK_MAX || s->wide_depth > RT_BEAM_LEVELS)
            return rt_fail(RT_ERR_UNSUPPORTED, "device-built tree too deep (%u pending entries, %u wide levels)",
                           s->max_stack, s->wide_depth);
        s->sizes[0] = used_wide * sizeof(rtbvh::Child);
        s->sizes[1] = used_nodes * sizeof(rtbvh::Node4);
        if (std::getenv("RAYS_HIP_VERBOSE"))
            std::fprintf(stderr, "[rays_hip] device build: %zu nodes4 (stack bound %u), %zu wide children, depth %u\n",
                         used_nodes, s->max_stack, used_wide, s->wide_depth);
    }
    HIP_TRY(hipMemcpy(d_objs, s->objects.data(), s->objects.size() * sizeof(DevObject), hipMemcpyHostToDevice));
    HIP_TRY(rtk::pack_tris(d_verts, d_tris, (uint32_t)total_tris, d_tri9, 0));
    HIP_TRY(rtk::leaf_bounds(d_verts, d_tris, d_leaf_first, (uint32_t)leaf_first.size(), d_leaf_bounds, d_leaf_gate, 0));
    HIP_TRY(rtk::gather_records(d_tri9, d_order, (uint32_t)n_recs_total, d_recs, 0));
    HIP_TRY(rtk::quantize_nodes(d_nodes, (uint32_t)n_nodes_total, d_qnodes, 0));
    HIP_TRY(hipDeviceSynchronize());

    s->view.objs = d_objs;
    s->view.n_objs = n_objs;
    {
        float ext = 0.0f;
        for (int k = 0; k < 3; ++k) ext = std::max(ext, s->bounds_hi[k] - s->bounds_lo[k]);
        s->view.scene_extent = ext;
        // Widest beam the beam walk takes (wider waves are split or take the mask walk). Measured on
        // soup-1M from 960x540 to 3840x2160: 1/64 of the scene is as good as any for shadow rays, and
        // camera rays only lose beyond 1/8; 1/128 sent the camera rays of small frames to the mask walk
        // (2.3 x slower).
        s->view.beam_limit = ext / 64.0f;
        if (const char* e = std::getenv("RAYS_HIP_BEAM_DIV")) s->view.beam_limit = ext / (float)std::atof(e);
    }
    s->view.nodes = d_nodes;
    s->view.qnodes = d_qnodes;
    s->view.wide = d_wide;
    s->view.recs = d_recs;
    s->view.tri9 = d_tri9;
    s->view.leaf_bounds = d_leaf_bounds;
    s->view.leaf_gate = d_leaf_gate;
    // triangle id lies in leaf id / 8 as long as no object with trailing triangles comes before it
    // How the persistent walks hand out the shared half of a launch's rays: through one counter where a
    // ray is a long walk, through RT_COUNTER_SHARDS of them where rays are cheap and the one word's
    // 88 requests a microsecond would bound the launch (the reference's scene with its own constants:
    // 5.3 -> 4.3 ms a frame, a 100,000-triangle soup's bounce frame 46.7 -> 35.0 ms; soups of three and
    // ten million triangles are 2 % slower that way, their waves then work on eight parts of the frame at
    // a time and share less of the tree in the caches; at one million the two are level).
    s->view.walk_counter_shards = s->n_tris_total < walk_shards_below() ? RT_COUNTER_SHARDS : 1u;
    s->view.leaves_uniform = 1u;
    s->view.n_rest_total = s->n_rest_total;
    for (size_t o = 0; o < s->objects.size(); ++o)
        for (size_t later = o + 1; later < s->objects.size() && s->objects[o].n_rest != 0; ++later)
            if (s->objects[later].n_tris > s->objects[later].n_rest) s->view.leaves_uniform = 0u;
    s->view.tri_data = d_data;
    s->view.rest_verts = d_rest;
    s->device_ptrs = pool.ptrs;
    pool.release();
    guard.s = nullptr;
    ctx->live_scenes += 1;
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    rt_ctx* c = s->ctx;
    (void)hipSetDevice(c->device);
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        delete s;
        c->live_scenes -= 1;
        last = c->destroy_requested && c->live_scenes == 0;
    }
    if (last) delete c;
}

uint32_t rt_scene_num_tris(const rt_scene* s) { return s ? s->n_tris : 0; }

int rt_scene_sizes(const rt_scene* s, uint64_t sizes[4]) {
    if (!s || !sizes) return rt_fail(RT_ERR_INVALID, "null argument");
    for (int k = 0; k < 4; ++k) sizes[k] = s->sizes[k];
    return RT_OK;
}

int rt_scene_order(const rt_scene* s, uint32_t obj, uint32_t* order) {
    if (!s || !order) return rt_fail(RT_ERR_INVALID, "null argument");
    if (obj >= s->order.size()) return rt_fail(RT_ERR_INVALID, "object %u out of range", obj);
    if (!s->order[obj].empty()) std::memcpy(order, s->order[obj].data(), s->order[obj].size() * 4);
    return RT_OK;
}

int rt_scene_packed(const rt_scene* s, uint32_t obj, float* tris9, float* leaf_bounds6) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (obj >= s->objects.size()) return rt_fail(RT_ERR_INVALID, "object %u out of range", obj);
    HIP_TRY(hipSetDevice(s->ctx->device));
    const DevObject& ob = s->objects[obj];
    if (tris9 && ob.n_tris)
        HIP_TRY(hipMemcpy(tris9, s->view.tri9 + 9 * (size_t)ob.tri_base, (size_t)ob.n_tris * 36, hipMemcpyDeviceToHost));
    if (leaf_bounds6 && ob.n_tris >= 8)
        HIP_TRY(hipMemcpy(leaf_bounds6, s->view.leaf_bounds + 6 * (size_t)ob.leaf_base, (size_t)(ob.n_tris / 8) * 24,
                          hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_trace_closest_device(rt_scene* s, const rt_ray* d_rays, uint64_t n, rt_hit* d_out, void* stream) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (int g = rt_runtime_guard()) return g;
    if (n == 0) return RT_OK;
    if (!d_rays || !d_out) return rt_fail(RT_ERR_INVALID, "null buffer");
    if (n > (1ull << 31) * RT_BLOCK) return rt_fail(RT_ERR_INVALID, "too many rays in one call");
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    uint32_t* counter = nullptr;
    int rc = take_counter(s->ctx, (hipStream_t)stream, &counter);
    if (rc != RT_OK) return rc;
    HIP_TRY(rtk::trace_closest(s->view, d_rays, n, d_out, counter, s->ctx->opt.trace, (hipStream_t)stream));
    return RT_OK;
}

int rt_trace_any_device(rt_scene* s, const rt_ray* d_rays, uint64_t n, uint8_t* d_occluded, void* stream) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (int g = rt_runtime_guard()) return g;
    if (n == 0) return RT_OK;
    if (!d_rays || !d_occluded) return rt_fail(RT_ERR_INVALID, "null buffer");
    if (n > (1ull << 31) * RT_BLOCK) return rt_fail(RT_ERR_INVALID, "too many rays in one call");
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    uint32_t* counter = nullptr;
    int rc = take_counter(s->ctx, (hipStream_t)stream, &counter);
    if (rc != RT_OK) return rc;
    HIP_TRY(rtk::trace_any(s->view, d_rays, n, d_occluded, counter, s->ctx->opt.trace, (hipStream_t)stream));
    return RT_OK;
}

}  // extern "C"

namespace {
template <class OutT, class Launch>
int trace_host(rt_scene* s, const rt_ray* rays, uint64_t n, OutT* out, Launch launch) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (n == 0) return RT_OK;
    if (!rays || !out) return rt_fail(RT_ERR_INVALID, "null buffer");
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    DevPool tmp;
    rt_ray* d_rays;
    OutT* d_out;
    HIP_TRY(tmp.alloc((void**)&d_rays, n * sizeof(rt_ray)));
    HIP_TRY(tmp.alloc((void**)&d_out, n * sizeof(OutT)));
    HIP_TRY(hipMemcpy(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    HIP_TRY(launch(d_rays, d_out));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out, n * sizeof(OutT), hipMemcpyDeviceToHost));
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_trace_closest(rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* out) {
    return trace_host<rt_hit>(s, rays, n, out, [&](const rt_ray* dr, rt_hit* dout) {
        uint32_t* counter = nullptr;
        if (take_counter(s->ctx, 0, &counter) != RT_OK) return hipErrorOutOfMemory;
        return rtk::trace_closest(s->view, dr, n, dout, counter, s->ctx->opt.trace, 0);
    });
}

int rt_trace_any(rt_scene* s, const rt_ray* rays, uint64_t n, uint8_t* occluded) {
    return trace_host<uint8_t>(s, rays, n, occluded, [&](const rt_ray* dr, uint8_t* dout) {
        uint32_t* counter = nullptr;
        if (take_counter(s->ctx, 0, &counter) != RT_OK) return hipErrorOutOfMemory;
        return rtk::trace_any(s->view, dr, n, dout, counter, s->ctx->opt.trace, 0);
    });
}

int rt_trace_closest_linear(rt_scene* s, const rt_ray* rays, uint64_t n, rt_hit* out) {
    return trace_host<rt_hit>(s, rays, n, out, [&](const rt_ray* dr, rt_hit* dout) {
        return rtk::trace_closest_linear(s->view, dr, n, dout, 0);
    });
}

uint32_t rt_shard_tiles(uint32_t width, uint32_t height, uint32_t shard_index, uint32_t shard_count) {
    if (shard_count == 0 || shard_index >= shard_count) return 0;
    const uint32_t tx = ceil_div(width, RT_TILE), ty = ceil_div(height, RT_TILE);
    uint32_t n = 0;
    for (uint32_t y = 0; y < ty; ++y)
        for (uint32_t x = 0; x < tx; ++x)
            if ((x + y) % shard_count == shard_index) ++n;
    return n;
}

int rt_render_device(rt_scene* s, const rt_render_params* p, uint8_t* d_rgba8, uint32_t* d_hit_ids, void* stream) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (int g = rt_runtime_guard()) return g;
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    return render_common(s, p, d_rgba8, d_hit_ids, nullptr, (hipStream_t)stream);
}

int rt_render_device_events(rt_scene* s, const rt_render_params* p, uint8_t* d_rgba8, uint32_t* d_hit_ids, void* stream,
                            void* const* events, int n_events) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (int g = rt_runtime_guard()) return g;
    if (!events || n_events != RT_RENDER_EVENTS) return rt_fail(RT_ERR_INVALID, "need %d events", RT_RENDER_EVENTS);
    for (int k = 0; k < n_events; ++k)
        if (!events[k]) return rt_fail(RT_ERR_INVALID, "event %d is null", k);
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    return render_common(s, p, d_rgba8, d_hit_ids, nullptr, (hipStream_t)stream, events);
}

int rt_render_device_stats(rt_scene* s, const rt_render_params* p, uint8_t* d_rgba8, uint32_t* d_hit_ids,
                           rt_stats* stats) {
    if (!s || !stats) return rt_fail(RT_ERR_INVALID, "null argument");
    if (int g = rt_runtime_guard()) return g;
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    DevPool tmp;
    DevStats* d_stats;
    HIP_TRY(tmp.alloc((void**)&d_stats, sizeof(DevStats)));
    HIP_TRY(hipMemset(d_stats, 0, sizeof(DevStats)));
    DevStats hp{}, hb{};
    bool projected = false;
    bool lists = false;
    int rc = render_common(s, p, d_rgba8, d_hit_ids, d_stats, 0, nullptr, &hp, &projected, &hb, &lists);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    uint32_t lf[6] = {0, 0, 0, 0, 0, 0};   // usable, entries, longest, cells with a list (k_lcell_check), -, rays done over
    if (lists) HIP_TRY(hipMemcpy(lf, s->work.lc_flags, sizeof(lf), hipMemcpyDeviceToHost));
    stats->shadow_by_cells = lf[0];
    stats->shadow_list_entries = lf[1];
    stats->shadow_list_longest = lf[2];
    stats->shadow_list_cells = lf[3];
    stats->shadow_redone = lf[5];
    stats->tree_built_on_device = s->built_on_device ? 1u : 0u;
    DevStats h;
    HIP_TRY(hipMemcpy(&h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
    stats->rays = h.rays;
    stats->box_tests = h.box_tests;
    stats->tri_tests = h.tri_tests;
    stats->node_visits = h.node_visits;
    stats->fetch_bytes = h.fetch_bytes;
    stats->primary_rays = hp.rays;
    stats->primary_box_tests = hp.box_tests;
    stats->primary_tri_tests = hp.tri_tests;
    stats->primary_node_visits = hp.node_visits;
    stats->primary_fetch_bytes = hp.fetch_bytes;
    stats->camera_by_projection = projected ? 1u : 0u;
    stats->bounce_rays = hb.rays;
    stats->bounce_box_tests = hb.box_tests;
    stats->bounce_tri_tests = hb.tri_tests;
    stats->bounce_node_visits = hb.node_visits;
    stats->bounce_fetch_bytes = hb.fetch_bytes;
    if (std::getenv("RAYS_HIP_VERBOSE"))
        std::fprintf(stderr,
                     "rays_hip: camera kernel: %llu beam walks, %llu mask walks, %llu refits, %llu mask-walk nodes; "
                     "frame: %llu / %llu / %llu / %llu\n",
                     hp.walks[0], hp.walks[1], hp.walks[2], hp.walks[3], h.walks[0], h.walks[1], h.walks[2], h.walks[3]);
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_render_params* p, uint8_t* rgba8, uint32_t* hit_ids) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    uint32_t spp = 0;
    int rc = check_render_params(p, &spp);
    if (rc != RT_OK) return rc;
    if (!rgba8) return rt_fail(RT_ERR_INVALID, "rgba8 buffer is null");
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    HIP_TRY(hipSetDevice(s->ctx->device));
    const size_t n_pix = (size_t)p->width * p->height;
    const bool sharded = p->shard_count > 1;
    const uint32_t my_tiles = sharded ? rt_shard_tiles(p->width, p->height, p->shard_index, p->shard_count) : 0;
    const size_t frame_bytes = sharded ? (size_t)my_tiles * RT_TILE * RT_TILE * 4 : n_pix * 4;
    DevPool tmp;
    uint8_t* d_rgba;
    uint32_t* d_ids = nullptr;
    HIP_TRY(tmp.alloc((void**)&d_rgba, frame_bytes));
    HIP_TRY(hipMemset(d_rgba, 0, std::max<size_t>(frame_bytes, 16)));
    if (hit_ids) {
        HIP_TRY(tmp.alloc((void**)&d_ids, n_pix * spp * 4));
        HIP_TRY(hipMemcpy(d_ids, hit_ids, n_pix * spp * 4, hipMemcpyHostToDevice));
    }
    rc = render_common(s, p, d_rgba, d_ids, nullptr, 0);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (!sharded) {
        HIP_TRY(hipMemcpy(rgba8, d_rgba, frame_bytes, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint8_t> compact(frame_bytes);
        if (frame_bytes) HIP_TRY(hipMemcpy(compact.data(), d_rgba, frame_bytes, hipMemcpyDeviceToHost));
        TileMap* tm = nullptr;
        rc = get_tile_map(s->ctx, p->width, p->height, p->shard_index, p->shard_count, &tm);
        if (rc != RT_OK) return rc;
        for (size_t k = 0; k < tm->list.size(); ++k) {
            const uint32_t t = tm->list[k], tx = t % tm->tiles_x, ty = t / tm->tiles_x;
            for (uint32_t ly = 0; ly < RT_TILE; ++ly) {
                const uint32_t y = ty * RT_TILE + ly;
                if (y >= p->height) break;
                const uint32_t x0 = tx * RT_TILE;
                const uint32_t w = std::min<uint32_t>(RT_TILE, p->width - x0);
                std::memcpy(rgba8 + ((size_t)y * p->width + x0) * 4,
                            compact.data() + (k * RT_TILE * RT_TILE + (size_t)ly * RT_TILE) * 4, (size_t)w * 4);
            }
        }
    }
    if (hit_ids) HIP_TRY(hipMemcpy(hit_ids, d_ids, n_pix * spp * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_deinterleave_tiles_device(rt_ctx* ctx, const uint8_t* d_gathered, uint32_t slab_tiles, uint32_t width,
                                 uint32_t height, uint32_t shard_count, uint8_t* d_rgba8, void* stream) {
    if (!ctx || !d_gathered || !d_rgba8) return rt_fail(RT_ERR_INVALID, "null argument");
    if (width == 0 || height == 0 || shard_count == 0) return rt_fail(RT_ERR_INVALID, "bad frame or shard count");
    if (int g = rt_runtime_guard()) return g;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    TileMap* tm = nullptr;
    int rc = get_tile_map(ctx, width, height, 0, shard_count, &tm);
    if (rc != RT_OK) return rc;
    for (uint32_t sh = 0; sh < shard_count; ++sh)
        if (rt_shard_tiles(width, height, sh, shard_count) > slab_tiles)
            return rt_fail(RT_ERR_INVALID, "slab_tiles %u is smaller than shard %u's tile count", slab_tiles, sh);
    HIP_TRY(rtk::deinterleave(d_gathered, tm->d_slot, slab_tiles, width, height, tm->tiles_x,
                              tm->tiles_x * tm->tiles_y, shard_count, d_rgba8, (hipStream_t)stream));
    return RT_OK;
}

}  // extern "C"
