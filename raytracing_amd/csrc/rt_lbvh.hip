// rt_lbvh.hip -- tree build on the device: triangles in Morton order of their centroids, then both
// trees bottom-up with a fixed fan-out (4 for the per-lane and mask walks, 64 for the beam walk).
//
// A level is an array of child records; a node is a run of `fan` consecutive records of the level
// below, so every level is one kernel over independent nodes and the whole build is a sort plus a
// few short kernels. The boxes are looser than those of the host's surface-area build (the cuts
// fall every `fan` records of the Morton curve, not where the geometry has gaps); the build is for
// scenes that are built often or are too large to wait for the host. Any conservative tree gives
// the reference's answer, because the nearest hit does not depend on the visiting order.
//
// Leaves hold one triangle. Record r of an object holds its triangle order[r] (Morton order).

#include <hip/hip_runtime.h>

#include "rt_kernels.h"

using rtbvh::Child;
using rtbvh::EMPTY;
using rtbvh::LEAF_BIT;
using rtbvh::Node4;

namespace {

constexpr float PAD_REL = 6e-6f;  // relative widening of every triangle box, as in the host build

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Exact bounds of triangle t's three vertices.
__device__ __forceinline__ void tri_bounds(const float* __restrict__ verts, const uint32_t* __restrict__ tris, uint32_t t,
                                           float lo[3], float hi[3]) {
    const uint32_t ia = tris[3 * (size_t)t], ib = tris[3 * (size_t)t + 1], ic = tris[3 * (size_t)t + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = verts[3 * (size_t)ia + k], b = verts[3 * (size_t)ib + k], c = verts[3 * (size_t)ic + k];
        lo[k] = fminf(fminf(a, b), c);
        hi[k] = fmaxf(fmaxf(a, b), c);
    }
}

// 30-bit Morton code of every triangle's box centre inside the object's box.
__global__ void k_lbvh_codes(const float* __restrict__ verts, const uint32_t* __restrict__ tris, uint32_t n, float3 bbmin,
                             float3 scale, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo[3], hi[3];
    tri_bounds(verts, tris, i, lo, hi);
    const float cx = (0.5f * lo[0] + 0.5f * hi[0] - bbmin.x) * scale.x;
    const float cy = (0.5f * lo[1] + 0.5f * hi[1] - bbmin.y) * scale.y;
    const float cz = (0.5f * lo[2] + 0.5f * hi[2] - bbmin.z) * scale.z;
    const uint32_t qx = (uint32_t)fminf(fmaxf(cx, 0.0f), 1023.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf(cy, 0.0f), 1023.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf(cz, 0.0f), 1023.0f);
    keys[i] = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
    vals[i] = i;
}

// Leaf level of both trees: record p of the object holds triangle sorted[p].
__global__ void k_lbvh_leaves(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                              const uint32_t* __restrict__ sorted, uint32_t n, uint32_t tri_base, uint32_t rec_base,
                              Child* __restrict__ wide0, Node4* __restrict__ nodes0, uint32_t* __restrict__ order) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = sorted[p];
    float lo[3], hi[3];
    tri_bounds(verts, tris, t, lo, hi);
    Child c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pad = fmaxf(fabsf(lo[k]), fabsf(hi[k])) * PAD_REL + 1e-30f;
        c.lo[k] = lo[k] - pad;
        c.hi[k] = hi[k] + pad;
    }
    c.ref = LEAF_BIT | ((rec_base + p) << 3);
    c.pad = 0;
    wide0[p] = c;
    nodes0[p >> 2].c[p & 3u] = c;
    order[p] = tri_base + t;
    if (p == n - 1) {
        Child e;
        for (int k = 0; k < 3; ++k) {
            e.lo[k] = INFINITY;
            e.hi[k] = -INFINITY;
        }
        e.ref = EMPTY;
        e.pad = 0;
        for (uint32_t s = (p & 3u) + 1u; s < 4u; ++s) nodes0[p >> 2].c[s] = e;
    }
}

// One level up in the wide tree: entry i of `out` is the node made of records [fan * i, ...) of `in`.
__global__ void k_lbvh_up_wide(const Child* __restrict__ in, uint32_t in_base, uint32_t count, uint32_t fan,
                               Child* __restrict__ out, uint32_t n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const uint32_t first = i * fan, last = min(first + fan, count);
    Child c;
    for (int k = 0; k < 3; ++k) {
        c.lo[k] = INFINITY;
        c.hi[k] = -INFINITY;
    }
    for (uint32_t j = first; j < last; ++j) {
        const Child& d = in[j];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c.lo[k] = fminf(c.lo[k], d.lo[k]);
            c.hi[k] = fmaxf(c.hi[k], d.hi[k]);
        }
    }
    c.ref = ((in_base + first) << 6) | (last - first - 1u);
    c.pad = 0;
    out[i] = c;
}

// One level up in the 4-wide tree: node j of `out` has the nodes 4j .. 4j+3 of `in` as children.
__global__ void k_lbvh_up_4(const Node4* __restrict__ in, uint32_t in_index_base, uint32_t count,
                            Node4* __restrict__ out, uint32_t n_out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    Node4 nd;
    for (uint32_t s = 0; s < 4u; ++s) {
        Child c;
        for (int k = 0; k < 3; ++k) {
            c.lo[k] = INFINITY;
            c.hi[k] = -INFINITY;
        }
        c.ref = EMPTY;
        c.pad = 0;
        const uint32_t src = 4u * j + s;
        if (src < count) {
            const Node4& d = in[src];
            for (uint32_t q = 0; q < 4u; ++q) {
                if (d.c[q].ref == EMPTY) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    c.lo[k] = fminf(c.lo[k], d.c[q].lo[k]);
                    c.hi[k] = fmaxf(c.hi[k], d.c[q].hi[k]);
                }
            }
            c.ref = in_index_base + src;
        }
        nd.c[s] = c;
    }
    out[j] = nd;
}

inline unsigned blocks_for(uint32_t n) { return (n + 255u) / 256u; }

}  // namespace

namespace rtk {

LbvhLayout lbvh_layout(uint32_t n, uint32_t fan) {
    LbvhLayout L{};
    // 4-wide tree: level 0 holds the leaf records four to a node.
    uint32_t count = (n + 3u) / 4u;
    L.n_nodes4 = count;
    L.depth4 = 1;
    while (count > 1u) {
        count = (count + 3u) / 4u;
        L.n_nodes4 += count;
        ++L.depth4;
    }
    // Wide tree: level 0 holds the n leaf records; the top node's records are the last level.
    count = n;
    L.n_wide = n;
    L.depth_wide = 1;
    while (count > fan) {
        count = (count + fan - 1u) / fan;
        L.n_wide += count;
        ++L.depth_wide;
    }
    L.top_count = count;
    return L;
}

hipError_t lbvh_build(const float* d_verts, const uint32_t* d_tris, uint32_t n, const float bbmin[3], const float bbmax[3],
                      uint32_t tri_base, uint32_t rec_base, uint32_t node_base, uint32_t wide_base, uint32_t fan,
                      Node4* d_nodes_all, Child* d_wide_all, uint32_t* d_order_all, uint32_t* d_keys[2],
                      uint32_t* d_vals[2], void* d_temp, size_t temp_bytes, uint32_t* root4, uint32_t* wide_root,
                      hipStream_t s) {
    if (n == 0) return hipSuccess;
    float3 mn = make_float3(bbmin[0], bbmin[1], bbmin[2]), sc;
    const float ex = bbmax[0] - bbmin[0], ey = bbmax[1] - bbmin[1], ez = bbmax[2] - bbmin[2];
    sc.x = ex > 0.0f ? 1024.0f / ex : 0.0f;
    sc.y = ey > 0.0f ? 1024.0f / ey : 0.0f;
    sc.z = ez > 0.0f ? 1024.0f / ez : 0.0f;
    hipLaunchKernelGGL(k_lbvh_codes, dim3(blocks_for(n)), dim3(256), 0, s, d_verts, d_tris, n, mn, sc, d_keys[0],
                       d_vals[0]);
    size_t tb = temp_bytes;
    hipError_t e = sort_pairs_u32(d_temp, &tb, d_keys[0], d_keys[1], d_vals[0], d_vals[1], n, 30, s);
    if (e != hipSuccess) return e;
    Node4* nodes = d_nodes_all + node_base;
    Child* wide = d_wide_all + wide_base;
    hipLaunchKernelGGL(k_lbvh_leaves, dim3(blocks_for(n)), dim3(256), 0, s, d_verts, d_tris, d_vals[1], n, tri_base,
                       rec_base, wide, nodes, d_order_all + rec_base);
    // 4-wide levels.
    uint32_t in_off = 0, count = (n + 3u) / 4u;
    while (count > 1u) {
        const uint32_t n_out = (count + 3u) / 4u, out_off = in_off + count;
        hipLaunchKernelGGL(k_lbvh_up_4, dim3(blocks_for(n_out)), dim3(256), 0, s, nodes + in_off, node_base + in_off,
                           count, nodes + out_off, n_out);
        in_off = out_off;
        count = n_out;
    }
    *root4 = node_base + in_off;
    // Wide levels.
    in_off = 0;
    count = n;
    while (count > fan) {
        const uint32_t n_out = (count + fan - 1u) / fan, out_off = in_off + count;
        hipLaunchKernelGGL(k_lbvh_up_wide, dim3(blocks_for(n_out)), dim3(256), 0, s, wide + in_off, wide_base + in_off,
                           count, fan, wide + out_off, n_out);
        in_off = out_off;
        count = n_out;
    }
    *wide_root = ((wide_base + in_off) << 6) | (count - 1u);
    return hipGetLastError();
}

}  // namespace rtk
