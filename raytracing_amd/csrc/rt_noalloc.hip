// rt_noalloc.hip -- rt_trace_closest_device_noalloc: the nearest-hit batch search for callers that
// cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc). rt_trace_closest_device takes the work counter of its persistent kernel from a pool
// the context allocates on first use; this entry point keeps that counter in a file-scope device
// array instead and calls no allocating routine at all.
//
// The array exists once per GPU, at an address of its own on each, and keeps whatever the last call
// left in it: the counters are cleared on the caller's stream before every launch. Two calls
// that overlap on one GPU (different streams, or host threads using different contexts on one
// device) share the counter and race; calls on one stream, or ordered by the caller, do not.

#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_api_internal.h"
#include "rt_kernels.h"

// Device memory of the entry point beyond its inputs and outputs: the ray hand-out counters of
// k_trace_closest_persistent (eight words, one per 64-byte line). RT_NOALLOC_SCRATCH_BYTES = 512 per GPU,
// held until the process ends.
#define RT_NOALLOC_SCRATCH_BYTES RT_COUNTER_BYTES
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

// Most rays one call takes: the persistent kernel's 32-bit ray index with the counter's overshoot
// at the end of a launch taken off (one chunk of 64 rays per wave, 8192 waves), rounded down to 2^31.
#define RT_NOALLOC_MAX_RAYS (1ull << 31)

// The array's address on each device, looked up the first time a context is created there.
#define RT_NOALLOC_MAX_DEVICES 64
static std::mutex g_setup_mu;
static void* g_scratch_ptr[RT_NOALLOC_MAX_DEVICES];

int rt_noalloc_setup(rt_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= RT_NOALLOC_MAX_DEVICES)
        return rt_fail(RT_ERR_UNSUPPORTED, "device index %d is beyond %d", ctx->device, RT_NOALLOC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_setup_mu);
    if (!g_scratch_ptr[ctx->device]) {
        // In host code `scratch` is only a handle for the symbol; the array's address on the
        // current device (the caller has made ctx->device current) comes from the runtime, and
        // differs from device to device.
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return rt_fail(RT_ERR_DEVICE, "scratch array has no address on device %d", ctx->device);
        g_scratch_ptr[ctx->device] = ptr;
    }
    ctx->noalloc_scratch = g_scratch_ptr[ctx->device];
    return RT_OK;
}

extern "C" int rt_trace_closest_device_noalloc(rt_scene* s, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                                               void* stream) {
    if (!s) return rt_fail(RT_ERR_INVALID, "scene is null");
    if (int g = rt_runtime_guard()) return g;
    if (n == 0) return RT_OK;
    if (!d_rays || !d_out) return rt_fail(RT_ERR_INVALID, "null buffer");
    if (n > RT_NOALLOC_MAX_RAYS)
        return rt_fail(RT_ERR_UNSUPPORTED, "rt_trace_closest_device_noalloc takes at most %llu rays per call (got %llu)",
                       (unsigned long long)RT_NOALLOC_MAX_RAYS, (unsigned long long)n);
    rt_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    uint32_t* counter = static_cast<uint32_t*>(ctx->noalloc_scratch);
    if (!counter) return rt_fail(RT_ERR_DEVICE, "scratch array was not set up for device %d", ctx->device);
    HIP_TRY(hipMemsetAsync(counter, 0, RT_NOALLOC_SCRATCH_BYTES, (hipStream_t)stream));
    HIP_TRY(rtk::trace_closest(s->view, d_rays, n, d_out, counter, ctx->opt.trace, (hipStream_t)stream));
    return RT_OK;
}
