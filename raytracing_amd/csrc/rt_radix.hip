// rt_radix.hip -- the sort that puts shadow rays into light-space order, shaped for its input: keys
// of 16 (at most 32) significant bits, of which about half say "no shadow ray" (all ones) and need
// not be moved at all, values that are the keys' own positions, and a count of valid keys that only
// the device knows. A stable least-significant-digit radix sort, eight bits a pass:
//   k_radix_hist     one wave per block of RT_RADIX_BLOCK keys: digit histogram of the valid keys
//   k_radix_groups   sums of the histograms over groups of 256 blocks
//   k_radix_bases    where every (digit, group) starts: digits in order, groups in order within a digit
//                    (folded into k_radix_offsets while the groups are few)
//   k_radix_offsets  where every (digit, block) starts
//   k_radix_scatter  one wave per block goes through its keys in order, 64 at a time: the lanes that
//                    share a digit find each other with eight ballots (one per bit), take consecutive
//                    places from the block's running cursor for that digit, and write key and value
// The first pass reads the keys where the producer left them (position = value) and drops the keys
// that are all ones; the later passes run over the valid keys only, whose number the first pass leaves
// in device memory (their launches are sized for the most there could be; the blocks beyond return).
// Stable in every pass, so the result is the order a library radix sort of the same keys gives.
#include <cstdint>

#include <hip/hip_runtime.h>

#include "rt_kernels.h"

namespace {

#ifndef RT_RADIX_BLOCK
#define RT_RADIX_BLOCK 2048u     // keys per block (one wave): 32 rounds of 64
#endif
#define RT_RADIX_GROUP 256u      // blocks per group of the offset computation
#define RT_RADIX_NONE 0xFFFFFFFFu
#ifndef RT_RADIX_AHEAD
#define RT_RADIX_AHEAD 4u        // rounds whose keys a wave loads before it works on them
#endif

__device__ __forceinline__ void radix_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Keys of block `block`, round r, for this lane; FIRST: all n positions, valid = not all ones; later
// passes: the first *n_valid positions.
template <bool FIRST>
__device__ __forceinline__ uint32_t radix_limit(uint32_t n, const uint32_t* __restrict__ n_valid) {
    return FIRST ? n : min(n, *n_valid);
}

template <bool FIRST>
__global__ __launch_bounds__(64) void k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n,
                                                   const uint32_t* __restrict__ n_valid, uint32_t shift,
                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t limit = radix_limit<FIRST>(n, n_valid);
    const uint32_t first = blockIdx.x * RT_RADIX_BLOCK;
    for (uint32_t k = lane; k < 256u; k += 64u) h[k] = 0u;
    radix_lds_sync();
    if (first < limit) {
        // Counting does not care about the order: four keys a lane and load (16 bytes), two loads ahead.
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + first);
        for (uint32_t r = 0; r < RT_RADIX_BLOCK / 256u; r += 2u) {
            uint4 q[2];
#pragma unroll
            for (uint32_t j = 0; j < 2u; ++j) {
                const uint32_t i = first + (r + j) * 256u + 4u * lane;
                q[j] = make_uint4(RT_RADIX_NONE, RT_RADIX_NONE, RT_RADIX_NONE, RT_RADIX_NONE);
                if (i + 3u < limit) {
                    q[j] = k4[(r + j) * 64u + lane];
                } else {
                    if (i < limit) q[j].x = keys[i];
                    if (i + 1u < limit) q[j].y = keys[i + 1u];
                    if (i + 2u < limit) q[j].z = keys[i + 2u];
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 2u; ++j) {
                const uint32_t i = first + (r + j) * 256u + 4u * lane;
                const uint32_t kk[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (uint32_t c = 0; c < 4u; ++c) {
                    const bool valid = i + c < limit && (!FIRST || kk[c] != RT_RADIX_NONE);
                    if (valid) atomicAdd(&h[(kk[c] >> shift) & 255u], 1u);
                }
            }
        }
        radix_lds_sync();
    }
    uint4* out = reinterpret_cast<uint4*>(hist + (size_t)blockIdx.x * 256u);
    out[lane] = make_uint4(h[4u * lane], h[4u * lane + 1u], h[4u * lane + 2u], h[4u * lane + 3u]);
}

// groups[g][d] = sum over the blocks of group g of hist[block][d]. One workgroup of 256 per group.
__global__ __launch_bounds__(256) void k_radix_groups(const uint32_t* __restrict__ hist, uint32_t n_blocks,
                                                      uint32_t* __restrict__ groups) {
    const uint32_t d = threadIdx.x, g = blockIdx.x;
    const uint32_t b0 = g * RT_RADIX_GROUP, b1 = min(n_blocks, b0 + RT_RADIX_GROUP);
    uint32_t sum = 0;
#pragma unroll 16
    for (uint32_t b = b0; b < b1; ++b) sum += hist[(size_t)b * 256u + d];
    groups[(size_t)g * 256u + d] = sum;
}

// bases[g][d] = the place where digit d of group g starts (digits in order, groups in order within a
// digit); *n_valid = the number of keys. One workgroup of 256 (one thread per digit).
__global__ __launch_bounds__(256) void k_radix_bases(const uint32_t* __restrict__ groups, uint32_t n_groups,
                                                     uint32_t* __restrict__ bases, uint32_t* __restrict__ n_valid,
                                                     bool set_count) {
    __shared__ uint32_t total[256];
    const uint32_t d = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll 16
    for (uint32_t g = 0; g < n_groups; ++g) sum += groups[(size_t)g * 256u + d];
    total[d] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t v = total[k];
        if (k < d) run += v;
        all += v;
    }
#pragma unroll 16
    for (uint32_t g = 0; g < n_groups; ++g) {
        bases[(size_t)g * 256u + d] = run;
        run += groups[(size_t)g * 256u + d];
    }
    if (d == 0 && set_count) *n_valid = all;
}

// offsets[block][d] = the place where digit d of that block starts. OWN_BASES: the workgroup works its
// group's bases out of `groups` itself (what k_radix_bases does, once per group instead of once: a few
// hundred reads a thread while there are few groups, and one launch less per pass); workgroup 0 then
// leaves the number of keys in *n_valid if set_count.
#define RT_RADIX_OWN_BASES_GROUPS 128u   // most groups for which the offsets' workgroups do so
template <bool OWN_BASES>
__global__ __launch_bounds__(256) void k_radix_offsets(const uint32_t* __restrict__ hist, uint32_t n_blocks,
                                                       const uint32_t* __restrict__ bases, uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ groups, uint32_t n_groups,
                                                       uint32_t* __restrict__ n_valid, bool set_count) {
    const uint32_t d = threadIdx.x, g = blockIdx.x;
    const uint32_t b0 = g * RT_RADIX_GROUP, b1 = min(n_blocks, b0 + RT_RADIX_GROUP);
    uint32_t run;
    if (OWN_BASES) {
        __shared__ uint32_t total[256];
        uint32_t before = 0, sum = 0;   // this digit in the groups before this one, and in all of them
#pragma unroll 16
        for (uint32_t k = 0; k < n_groups; ++k) {
            const uint32_t v = groups[(size_t)k * 256u + d];
            before += k < g ? v : 0u;
            sum += v;
        }
        total[d] = sum;
        __syncthreads();
        uint32_t lower = 0, all = 0;    // the digits below this one, and all keys
        for (uint32_t k = 0; k < 256u; ++k) {
            const uint32_t v = total[k];
            lower += k < d ? v : 0u;
            all += v;
        }
        run = lower + before;
        if (g == 0 && d == 0 && set_count) *n_valid = all;
    } else {
        run = bases[(size_t)g * 256u + d];
    }
#pragma unroll 16
    for (uint32_t b = b0; b < b1; ++b) {
        offsets[(size_t)b * 256u + d] = run;
        run += hist[(size_t)b * 256u + d];
    }
}

template <bool FIRST>
__global__ __launch_bounds__(64) void k_radix_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                      uint32_t n, const uint32_t* __restrict__ n_valid, uint32_t shift,
                                                      const uint32_t* __restrict__ offsets, uint32_t* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t cursor[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t limit = radix_limit<FIRST>(n, n_valid);
    const uint32_t first = blockIdx.x * RT_RADIX_BLOCK;
    if (first >= limit) return;
    {
        const uint4 o = reinterpret_cast<const uint4*>(offsets + (size_t)blockIdx.x * 256u)[lane];
        cursor[4u * lane] = o.x, cursor[4u * lane + 1u] = o.y, cursor[4u * lane + 2u] = o.z, cursor[4u * lane + 3u] = o.w;
    }
    radix_lds_sync();
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < RT_RADIX_BLOCK / 64u; r += RT_RADIX_AHEAD) {
        uint32_t key[RT_RADIX_AHEAD], val[RT_RADIX_AHEAD];
#pragma unroll
        for (uint32_t j = 0; j < RT_RADIX_AHEAD; ++j) {
            const uint32_t i = first + (r + j) * 64u + lane;
            key[j] = i < limit ? keys[i] : RT_RADIX_NONE;
            val[j] = FIRST ? i : (i < limit ? vals[i] : 0u);
        }
#pragma unroll
        for (uint32_t j = 0; j < RT_RADIX_AHEAD; ++j) {
            const uint32_t i = first + (r + j) * 64u + lane;
            const bool valid = i < limit && (!FIRST || key[j] != RT_RADIX_NONE);
            const unsigned long long any = __ballot(valid);
            if (any == 0ull) continue;
            const uint32_t digit = (key[j] >> shift) & 255u;
            // the lanes with this lane's digit
            unsigned long long peers = any;
#pragma unroll
            for (uint32_t b = 0; b < 8u; ++b) {
                const bool bit = ((digit >> b) & 1u) != 0u;
                const unsigned long long m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const uint32_t rank = (uint32_t)__builtin_popcountll(peers & below);
                const uint32_t pos = cursor[digit] + rank;
                keys_out[pos] = key[j];
                vals_out[pos] = val[j];
            }
            radix_lds_sync();
            if (valid && (peers & below) == 0ull) cursor[digit] += (uint32_t)__builtin_popcountll(peers);
            radix_lds_sync();
        }
    }
}

struct RadixPlan {
    uint32_t n_blocks, n_groups;
    size_t hist_off, offsets_off, groups_off, bases_off, tmp_keys_off, tmp_vals_off, total;
};
RadixPlan radix_plan(size_t n) {
    RadixPlan p;
    p.n_blocks = (uint32_t)((n + RT_RADIX_BLOCK - 1) / RT_RADIX_BLOCK);
    p.n_groups = (p.n_blocks + RT_RADIX_GROUP - 1) / RT_RADIX_GROUP;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    p.hist_off = take((size_t)p.n_blocks * 256 * 4);
    p.offsets_off = take((size_t)p.n_blocks * 256 * 4);
    p.groups_off = take((size_t)p.n_groups * 256 * 4);
    p.bases_off = take((size_t)p.n_groups * 256 * 4);
    p.tmp_keys_off = take(n * 4);
    p.tmp_vals_off = take(n * 4);
    p.total = off;
    return p;
}

}  // namespace

namespace rtk {

size_t radix_workspace_bytes(size_t n) { return radix_plan(n).total; }

hipError_t radix_sort_valid(void* work, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, size_t n,
                            unsigned key_bits, uint32_t* d_n_valid, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(d_n_valid, 0, 4, s);
    const RadixPlan p = radix_plan(n);
    char* base = static_cast<char*>(work);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + p.hist_off);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(base + p.offsets_off);
    uint32_t* groups = reinterpret_cast<uint32_t*>(base + p.groups_off);
    uint32_t* bases = reinterpret_cast<uint32_t*>(base + p.bases_off);
    uint32_t* tmp_keys = reinterpret_cast<uint32_t*>(base + p.tmp_keys_off);
    uint32_t* tmp_vals = reinterpret_cast<uint32_t*>(base + p.tmp_vals_off);
    const unsigned passes = key_bits == 0 ? 1u : (key_bits + 7u) / 8u;
    const uint32_t* src_k = keys_in;
    const uint32_t* src_v = nullptr;
    for (unsigned pass = 0; pass < passes; ++pass) {
        // the last pass writes the caller's arrays; the passes before it alternate so that it can
        const bool to_out = ((passes - 1u - pass) & 1u) == 0u;
        uint32_t* dst_k = to_out ? keys_out : tmp_keys;
        uint32_t* dst_v = to_out ? vals_out : tmp_vals;
        const uint32_t shift = 8u * pass;
        if (pass == 0) {
            hipLaunchKernelGGL(k_radix_hist<true>, dim3(p.n_blocks), dim3(64), 0, s, src_k, (uint32_t)n, d_n_valid, shift, hist);
        } else {
            hipLaunchKernelGGL(k_radix_hist<false>, dim3(p.n_blocks), dim3(64), 0, s, src_k, (uint32_t)n, d_n_valid, shift, hist);
        }
        hipLaunchKernelGGL(k_radix_groups, dim3(p.n_groups), dim3(256), 0, s, hist, p.n_blocks, groups);
        if (p.n_groups <= RT_RADIX_OWN_BASES_GROUPS) {
            hipLaunchKernelGGL(k_radix_offsets<true>, dim3(p.n_groups), dim3(256), 0, s, hist, p.n_blocks, bases, offsets, groups,
                               p.n_groups, d_n_valid, pass == 0);
        } else {
            hipLaunchKernelGGL(k_radix_bases, dim3(1), dim3(256), 0, s, groups, p.n_groups, bases, d_n_valid, pass == 0);
            hipLaunchKernelGGL(k_radix_offsets<false>, dim3(p.n_groups), dim3(256), 0, s, hist, p.n_blocks, bases, offsets, groups,
                               p.n_groups, d_n_valid, false);
        }
        if (pass == 0) {
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(p.n_blocks), dim3(64), 0, s, src_k, src_v, (uint32_t)n, d_n_valid,
                               shift, offsets, dst_k, dst_v);
        } else {
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(p.n_blocks), dim3(64), 0, s, src_k, src_v, (uint32_t)n, d_n_valid,
                               shift, offsets, dst_k, dst_v);
        }
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}

}  // namespace rtk
