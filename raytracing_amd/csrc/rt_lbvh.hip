// rt_lbvh.hip -- tree build on the device. Triangles are put in Morton order of their box centres
// (one radix sort); then one of two builders makes both trees (4-wide for the per-lane and mask
// walks, up to 64-wide for the beam walk):
//  * radix tree (default): a binary tree whose cuts fall where the sorted codes first differ, i.e.
//    on the spatial mid-planes (every inner node finds its range and cut on its own), boxes fitted
//    bottom-up (every leaf climbs, the second arrival at a node joins its children), then both trees
//    collapsed from it top-down, one launch per level: a 4-wide node opens its largest inner child
//    until four slots are used; a wide node opens children, largest first, until each holds at most
//    fan^(levels - 1) leaves, so that the levels below come out full.
//  * fixed fan-out: a level is an array of child records and a node is a run of `fan` consecutive
//    records of the level below. No host synchronisation and no work memory, but looser boxes (the
//    cuts fall every `fan` records of the Morton curve, not where the geometry has gaps). Used when
//    the radix tree comes out too deep for the walks' stacks.
// Both are for scenes that are built often or are too large to wait for the host's surface-area
// build. Any conservative tree gives the reference's answer, because the nearest hit does not
// depend on the visiting order.
//
// Leaves hold one triangle. Record r of an object holds its triangle order[r] (Morton order).

#include <hip/hip_runtime.h>

#include "rt_kernels.h"

using rtbvh::Child;
using rtbvh::EMPTY;
using rtbvh::LEAF_BIT;
using rtbvh::Node4;

namespace {

// Widening of every triangle box, as in the host build: relative to the box's own coordinates plus
// an absolute part tied to the object's largest coordinate (see bvh_build.cpp).
constexpr float PAD_REL = 6e-6f;
// PAD_ABS is ten times what those roundings need: the nearest-hit walks rely on reaching every triangle
// whose hit lies less than half the padding (as a distance along the ray) behind the nearest one (tie_note
// in rt_kernels.hip, which reads the same constant).
constexpr float PAD_ABS = RT_PAD_ABS;   // bvh_build.h

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Exact bounds of triangle t's three vertices. A triangle with a vertex that is not finite gets the
// point (0, 0, 0): its test can accept no ray (every product with its data is infinite or NaN, and a NaN
// the arithmetic makes is no hit in the reference either), so it only needs a place in the tree, and an
// infinite box would make every box above it infinite.
__device__ __forceinline__ void tri_bounds(const float* __restrict__ verts, const uint32_t* __restrict__ tris, uint32_t t,
                                           float lo[3], float hi[3]) {
    const uint32_t ia = tris[3 * (size_t)t], ib = tris[3 * (size_t)t + 1], ic = tris[3 * (size_t)t + 2];
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = verts[3 * (size_t)ia + k], b = verts[3 * (size_t)ib + k], c = verts[3 * (size_t)ic + k];
        lo[k] = fminf(fminf(a, b), c);
        hi[k] = fmaxf(fmaxf(a, b), c);
        finite = finite && fabsf(a) < INFINITY && fabsf(b) < INFINITY && fabsf(c) < INFINITY;
    }
    if (!finite) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
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
                              float abs_pad, Child* __restrict__ wide0, Node4* __restrict__ nodes0, uint32_t* __restrict__ order) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = sorted[p];
    float lo[3], hi[3];
    tri_bounds(verts, tris, t, lo, hi);
    Child c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pad = fmaxf(fabsf(lo[k]), fabsf(hi[k])) * PAD_REL + abs_pad;
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

// ---------------------------------------------------------------------------------------------
// Radix-tree build: a binary tree whose cuts fall where the sorted Morton codes first differ
// (the spatial mid-planes), boxes fitted bottom-up, then collapsed top-down into both trees, one
// launch per level of the result. Binary nodes: internal i in [0, n-1), leaf p as (n-1) + p;
// internal 0 is the root.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int code_delta(const uint32_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a == b ? 32 + __clz((uint32_t)(i ^ j)) : __clz(a ^ b);
}

__global__ void k_radix_tree(const uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ left,
                             uint32_t* __restrict__ right, uint32_t* __restrict__ parent, uint32_t* __restrict__ n_leaves) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = code_delta(keys, n, i, i + 1) - code_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = code_delta(keys, n, i, i - d);
    int lmax = 2;
    while (code_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (code_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = code_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (code_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t lc = lo == gamma ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    const uint32_t rc = hi == gamma + 1 ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    left[i] = lc;
    right[i] = rc;
    parent[lc] = (uint32_t)i;
    parent[rc] = (uint32_t)i;
    n_leaves[i] = (uint32_t)(hi - lo + 1);
    if (i == 0) parent[0] = 0xFFFFFFFFu;
}

// Boxes of the binary nodes: every leaf climbs; the second arrival at a node joins its children.
__global__ void k_radix_fit(const Child* __restrict__ leaf_recs, int n, const uint32_t* __restrict__ left,
                            const uint32_t* __restrict__ right, const uint32_t* __restrict__ parent,
                            uint32_t* __restrict__ arrived, float* __restrict__ boxes) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n) return;
    {
        const Child c = leaf_recs[p];
        float* b = boxes + 6 * (size_t)(n - 1 + p);
        for (int k = 0; k < 3; ++k) {
            b[k] = c.lo[k];
            b[3 + k] = c.hi[k];
        }
    }
    __threadfence();
    uint32_t cur = parent[n - 1 + p];
    while (cur != 0xFFFFFFFFu) {
        if (atomicAdd(&arrived[cur], 1u) == 0u) return;
        __threadfence();
        const volatile float* a = boxes + 6 * (size_t)left[cur];
        const volatile float* c = boxes + 6 * (size_t)right[cur];
        float* o = boxes + 6 * (size_t)cur;
        for (int k = 0; k < 3; ++k) {
            o[k] = fminf(a[k], c[k]);
            o[3 + k] = fmaxf(a[3 + k], c[3 + k]);
        }
        __threadfence();
        cur = parent[cur];
    }
}

struct Collapse {
    const uint32_t* left;
    const uint32_t* right;
    const uint32_t* n_leaves;  // per internal node
    const float* boxes;        // 6 floats per binary node
    int n;                     // leaves
    uint32_t rec_base;
};

__device__ __forceinline__ Child child_of(const Collapse& c, uint32_t node) {
    Child out;
    const float* b = c.boxes + 6 * (size_t)node;
    for (int k = 0; k < 3; ++k) {
        out.lo[k] = b[k];
        out.hi[k] = b[3 + k];
    }
    out.ref = node >= (uint32_t)(c.n - 1) ? (LEAF_BIT | ((c.rec_base + (node - (uint32_t)(c.n - 1))) << 3)) : 0u;
    out.pad = 0;
    return out;
}

// One wide node per wave. Entry: (binary node, index of the child record in `wide` whose reference
// waits for this node, or all ones for the root). Children are opened, largest first, until each
// holds at most `fan`^(levels - 1) leaves, so that the levels below come out full.
__global__ __launch_bounds__(64) void k_collapse_wide(Collapse c, const uint2* __restrict__ frontier, uint32_t n_front,
                                                       uint32_t fan, Child* __restrict__ wide, uint32_t wide_base,
                                                       uint32_t* __restrict__ counters, uint2* __restrict__ next,
                                                       uint32_t* __restrict__ root_ref) {
    const uint32_t e = blockIdx.x;
    if (e >= n_front) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t b = frontier[e].x, patch = frontier[e].y;
    const uint32_t none = 0xFFFFFFFFu;
    uint32_t cand = lane == 0 ? c.left[b] : (lane == 1 ? c.right[b] : none);
    uint32_t count = 2;
    const uint32_t total = c.n_leaves[b];
    unsigned long long target = 1;
    while (target * fan < total) target *= fan;
    while (count < fan) {
        const bool inner = cand != none && cand < (uint32_t)(c.n - 1);
        const uint32_t size = inner ? c.n_leaves[cand] : 0u;
        unsigned long long key = (inner && size > target) ? (((unsigned long long)size << 8) | (63u - lane)) : 0ull;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
        }
        if (key == 0ull) break;
        const uint32_t pick = 63u - (uint32_t)(key & 63ull);
        const uint32_t opened = (uint32_t)__shfl((int)cand, (int)pick, 64);
        if (lane == pick) cand = c.left[opened];
        if (lane == count) cand = c.right[opened];
        ++count;
    }
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[0], count);
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (lane < count) {
        Child ch = child_of(c, cand);
        wide[base + lane] = ch;
        if (cand < (uint32_t)(c.n - 1)) {
            const uint32_t slot = atomicAdd(&counters[2], 1u);
            next[slot] = make_uint2(cand, base + lane);
        }
    }
    if (lane == 0) {
        const uint32_t ref = ((wide_base + base) << 6) | (count - 1u);
        if (patch == none) *root_ref = ref;
        else wide[patch].ref = ref;
    }
}

// One 4-wide node per thread. Entry: (binary node, index of the 4-wide node to fill). The largest
// inner child is opened until four slots are used.
__global__ void k_collapse_4(Collapse c, const uint2* __restrict__ frontier, uint32_t n_front, Node4* __restrict__ nodes,
                             uint32_t node_base, uint32_t* __restrict__ counters, uint2* __restrict__ next) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_front) return;
    const uint32_t b = frontier[e].x, mine = frontier[e].y;
    uint32_t cand[4] = {c.left[b], c.right[b], 0xFFFFFFFFu, 0xFFFFFFFFu};
    int count = 2;
    while (count < 4) {
        int pick = -1;
        float best = -1.0f;
        for (int q = 0; q < count; ++q) {
            if (cand[q] >= (uint32_t)(c.n - 1)) continue;
            const float* x = c.boxes + 6 * (size_t)cand[q];
            const float dx = x[3] - x[0], dy = x[4] - x[1], dz = x[5] - x[2];
            const float a = dx * dy + dy * dz + dz * dx;
            if (a > best) {
                best = a;
                pick = q;
            }
        }
        if (pick < 0) break;
        const uint32_t opened = cand[pick];
        cand[pick] = c.left[opened];
        cand[count++] = c.right[opened];
    }
    Node4 nd;
    for (int q = 0; q < 4; ++q) {
        Child ch;
        if (q < count) {
            ch = child_of(c, cand[q]);
            if (cand[q] < (uint32_t)(c.n - 1)) {
                const uint32_t idx = atomicAdd(&counters[1], 1u);
                ch.ref = node_base + idx;
                const uint32_t slot = atomicAdd(&counters[2], 1u);
                next[slot] = make_uint2(cand[q], idx);
            }
        } else {
            for (int k = 0; k < 3; ++k) {
                ch.lo[k] = INFINITY;
                ch.hi[k] = -INFINITY;
            }
            ch.ref = EMPTY;
            ch.pad = 0;
        }
        nd.c[q] = ch;
    }
    nodes[mine] = nd;
}

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

// PAD_ABS times the object's largest coordinate (its box holds every vertex).
static float abs_pad_of(const float bbmin[3], const float bbmax[3]) {
    float m = 0.0f;
    for (int k = 0; k < 3; ++k) m = fmaxf(m, fmaxf(fabsf(bbmin[k]), fabsf(bbmax[k])));
    return m * PAD_ABS + 1e-30f;
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
                       rec_base, abs_pad_of(bbmin, bbmax), wide, nodes, d_order_all + rec_base);
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

// Radix-tree build of one object. Capacities: d_nodes_all + node_base holds up to n 4-wide nodes,
// d_wide_all + wide_base up to 2 n child records. d_work: 2 n - 1 boxes of 6 floats, then five
// arrays of 2 n words (children, parents, leaf counts, arrival flags), then two frontiers of n
// pairs, then 4 counter words. Synchronises with the host once per level of each tree.
size_t lbvh_radix_work_bytes(uint32_t n) {
    return (size_t)(2 * (size_t)n) * 24 + 5 * (size_t)(2 * (size_t)n) * 4 + 2 * (size_t)n * 8 + 64;
}

hipError_t lbvh_build_radix(const float* d_verts, const uint32_t* d_tris, uint32_t n, const float bbmin[3],
                            const float bbmax[3], uint32_t tri_base, uint32_t rec_base, uint32_t node_base,
                            uint32_t wide_base, uint32_t fan, Node4* d_nodes_all, Child* d_wide_all,
                            uint32_t* d_order_all, uint32_t* d_keys[2], uint32_t* d_vals[2], void* d_temp,
                            size_t temp_bytes, void* d_work, LbvhResult* out, hipStream_t s) {
    if (n < 2) return hipErrorInvalidValue;
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
    // Work arrays.
    char* w = static_cast<char*>(d_work);
    float* boxes = reinterpret_cast<float*>(w);
    w += (size_t)(2 * (size_t)n) * 24;
    uint32_t* left = reinterpret_cast<uint32_t*>(w);
    uint32_t* right = left + 2 * (size_t)n;
    uint32_t* parent = right + 2 * (size_t)n;
    uint32_t* n_leaves = parent + 2 * (size_t)n;
    uint32_t* arrived = n_leaves + 2 * (size_t)n;
    uint2* front[2];
    front[0] = reinterpret_cast<uint2*>(arrived + 2 * (size_t)n);
    front[1] = front[0] + n;
    uint32_t* counters = reinterpret_cast<uint32_t*>(front[1] + n);  // wide records, 4-wide nodes, next frontier, root
    Node4* nodes = d_nodes_all + node_base;
    Child* wide = d_wide_all + wide_base;
    // Leaf records: the wide array's first n entries are used as staging for the leaf boxes (they
    // are rewritten by the collapse, which reads boxes[] only), the 4-wide level-0 writes land in
    // nodes[0 .. n/4) and are overwritten likewise.
    hipLaunchKernelGGL(k_lbvh_leaves, dim3(blocks_for(n)), dim3(256), 0, s, d_verts, d_tris, d_vals[1], n, tri_base,
                       rec_base, abs_pad_of(bbmin, bbmax), wide, nodes, d_order_all + rec_base);
    e = hipMemsetAsync(arrived, 0, (size_t)n * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_radix_tree, dim3(blocks_for(n)), dim3(256), 0, s, d_keys[1], (int)n, left, right, parent,
                       n_leaves);
    hipLaunchKernelGGL(k_radix_fit, dim3(blocks_for(n)), dim3(256), 0, s, wide, (int)n, left, right, parent, arrived,
                       boxes);
    const Collapse c{left, right, n_leaves, boxes, (int)n, rec_base};
    const uint2 root_entry = make_uint2(0u, 0xFFFFFFFFu);
    uint32_t init[4] = {0u, 1u, 0u, 0u};
    // ---- wide tree ----
    e = hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(front[0], &root_entry, sizeof(root_entry), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    uint32_t n_front = 1, levels = 0;
    int cur = 0;
    while (n_front > 0) {
        hipLaunchKernelGGL(k_collapse_wide, dim3(n_front), dim3(64), 0, s, c, front[cur], n_front, fan, wide, wide_base,
                           counters, front[cur ^ 1], counters + 3);
        e = hipMemcpyAsync(&n_front, counters + 2, 4, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(counters + 2, 0, 4, s);
        if (e != hipSuccess) return e;
        cur ^= 1;
        ++levels;
    }
    out->depth_wide = levels;
    uint32_t tail[4];
    e = hipMemcpyAsync(tail, counters, sizeof(tail), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    out->n_wide = tail[0];
    out->wide_root = tail[3];
    // ---- 4-wide tree ----
    const uint2 root4 = make_uint2(0u, 0u);
    e = hipMemcpyAsync(front[0], &root4, sizeof(root4), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    n_front = 1;
    levels = 0;
    cur = 0;
    while (n_front > 0) {
        hipLaunchKernelGGL(k_collapse_4, dim3(blocks_for(n_front)), dim3(256), 0, s, c, front[cur], n_front, nodes,
                           node_base, counters, front[cur ^ 1]);
        e = hipMemcpyAsync(&n_front, counters + 2, 4, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(counters + 2, 0, 4, s);
        if (e != hipSuccess) return e;
        cur ^= 1;
        ++levels;
    }
    out->depth4 = levels;
    e = hipMemcpyAsync(tail, counters, sizeof(tail), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    out->n_nodes4 = tail[1];
    out->root4 = node_base;
    return hipGetLastError();
}

}  // namespace rtk
