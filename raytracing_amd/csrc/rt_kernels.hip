// rt_kernels.hip -- gfx950 kernels for the trace_ray path: one ray per lane, wave64.
//
// Parity-critical arithmetic lives in rt_exact.h and is used unchanged here. The tree walk
// itself only has to be conservative (it decides which triangles get tested, never the result
// of a test), so its box test is free to use fused arithmetic.
//
// Compile with -ffp-contract=off (see rt_exact.h).

#include "rt_kernels.h"

#include <algorithm>
#include <cstdlib>

#include "rt_exact.h"

using rtbvh::EMPTY;
using rtbvh::LEAF_BIT;
using rtbvh::Node4;

namespace {

struct LaneStats {
    uint32_t box = 0, tri = 0, node = 0, rays = 0;
    uint32_t walks[4] = {0, 0, 0, 0};  // lane 0 only: beam walks, mask walks, beam refits, mask-walk node fetches
    uint32_t bytes = 0;  // structure bytes requested: a scalar load counts once per wave, a vector load per lane
};

// A nearest-hit result that stands for "not decided here": the wave-wide and the persistent walks
// hand the rare ray whose answer may depend on the order of the reference's search (closest_needs_resolve)
// to a second kernel, which runs the per-lane search with closest_resolve for it.
#define RT_REDO 0xFFFFFFFEu
struct HitD {
    uint32_t tri;
    uint32_t obj;
    float t, u, v;
    float closest;
    uint32_t near;   // the current object's walk: least distance, in units of the last place, between an accepted
                     // triangle's t and the nearest distance of that moment (tie_note); a tie when <= RT_TIE_WALK
};
// Walks note when an accepted triangle's t lies within RT_TIE_WALK units of the last place of the nearest
// distance of that moment, on either side (+inf, the distance before any hit, is near nothing). The nearest
// distance only falls, so every accepted triangle that near the final one meets it, or a value between the
// two, in one of these comparisons -- if the walk tests it at all. The trees' boxes are padded by
// RT_PAD_ABS of the object's largest coordinate (PAD_ABS of the builders), so a triangle whose hit lies less
// than pad / |d| behind the nearest one is never culled: its box begins that much before its hit along
// whichever axis it is entered by. closest_needs_resolve relies on half of that.
#define RT_TIE_WALK 1024u
// (The note is a minimum kept in a vector register: a flag per lane that changes inside the walks' loops is kept
// as a mask in scalar registers and merged at every join of the control flow, which cost the camera rays' walk
// 0.2 ms of 3.1.)
__device__ __forceinline__ void tie_note(float t, float closest, bool accepted, uint32_t& near) {
    uint32_t apart;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(apart) : "v"(__float_as_uint(t)), "v"(__float_as_uint(closest)));
    near = min(near, accepted ? apart : 0xFFFFFFFFu);
}

__device__ __forceinline__ rt_f3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }

// The reference's own gate in front of a triangle of the 8-wide path: trace_ray tests the triangles of
// a leaf (eight consecutive triangles) only for rays that pass the slab test of the leaf's box --
// PackedBoxes::intersect for the leaves that fill packets of eight (src/main.rs:103-111,
// src/simd_accel.rs:342-377), box_intersection_check for an object's last n_leaves % 8 leaves
// (src/main.rs:134-136). A triangle test that rounds badly (a needle whose determinant cancels) can
// accept a ray that passes the leaf's box by a thousandth of its size, which the reference never
// tests; the searches here reach triangles by other routes (a tree of their own boxes, projection,
// light-space cells), so a hit they are about to accept is put through this gate first. The shadow
// search passes +inf for the ray's length (src/main.rs:185); the nearest-hit search passes the nearest
// distance of the moment (src/main.rs:98,127), which makes its answer depend on the order of the
// search where two triangles lie within rounding of each other: the nearest-hit searches here test
// their winner with the ray cut at the winner's own distance and play the reference's order back where
// that fails (closest_needs_resolve and closest_resolve below; the record loop of k_raster_tile).
// `ob` is the triangle's object, `id` its global number (not a trailing triangle).
// lo4, hi4: the leaf's entry of SceneView::leaf_gate (its box, and which of the two tests it takes).
__device__ __forceinline__ bool ref_leaf_gate(float4 lo4, float4 hi4, rt_f3 o, rt_f3 inv, float ray_length = RT_INF) {
    const rt_f3 lo = f3(lo4.x, lo4.y, lo4.z), hi = f3(hi4.x, hi4.y, hi4.z);
    return lo4.w != 0.0f ? packed_box_hit(lo, hi, o, inv, ray_length) : box_intersection_check(o, inv, lo, hi);
}
__device__ __forceinline__ bool ref_leaf_gate(const SceneView& sv, const DevObject& ob, uint32_t id, rt_f3 o, rt_f3 inv) {
    const float4* lg = sv.leaf_gate + 2 * (size_t)(ob.leaf_base + ((id - ob.tri_base) >> 3));
    return ref_leaf_gate(lg[0], lg[1], o, inv);
}
// The same for callers that do not know the triangle's object.
__device__ __forceinline__ bool ref_leaf_gate(const SceneView& sv, uint32_t id, rt_f3 o, rt_f3 inv) {
    if (sv.leaves_uniform) {
        const float4* lg = sv.leaf_gate + 2 * (size_t)(id >> 3);
        return ref_leaf_gate(lg[0], lg[1], o, inv);
    }
    uint32_t oi = 0;
    while (oi + 1 < sv.n_objs && id >= sv.objs[oi].tri_base + sv.objs[oi].n_tris) ++oi;
    return ref_leaf_gate(sv, sv.objs[oi], id, o, inv);
}

// ---------------------------------------------------------------------------------------------
// Per-lane traversal stack: RT_STACK_LDS entries in LDS (lane-interleaved, conflict-free
// 8-byte accesses), the rest in private memory. An entry is (node reference, entry distance).
// ---------------------------------------------------------------------------------------------
template <int NLDS, int STRIDE>
struct LaneStackT {
    static_assert(NLDS <= RT_STACK_MAX, "stack split");
    uint2* lds;    // &s_stack[threadIdx.x]; entry e lives at lds[e * STRIDE] (STRIDE = workgroup size)
    uint2* spill;  // RT_STACK_MAX - NLDS entries of the kernel's, in private memory (an array inside this
                   // struct would take sp and lds into private memory with it)
    int sp;
    static constexpr int REST = RT_STACK_MAX - NLDS;
    __device__ __forceinline__ void push(uint32_t node, float tnear) {
        uint2 e = make_uint2(node, __float_as_uint(tnear));
        if (sp < NLDS) lds[sp * STRIDE] = e;
        else if (sp < RT_STACK_MAX) spill[sp - NLDS] = e;
        // The host refuses trees whose pending-entry bound exceeds RT_STACK_MAX, so the
        // stack cannot overflow; the guard above only keeps a corrupt tree from writing wild.
        sp = sp < RT_STACK_MAX ? sp + 1 : sp;
    }
    __device__ __forceinline__ uint2 pop() {
        --sp;
        return sp < NLDS ? lds[sp * STRIDE] : spill[sp - NLDS];
    }
};
using LaneStack = LaneStackT<RT_STACK_LDS, RT_BLOCK>;

#define RT_CSWAP(ka, kb, ca, cb)            \
    {                                       \
        bool sw_ = (ka) > (kb);             \
        float tk_ = sw_ ? (kb) : (ka);      \
        float tl_ = sw_ ? (ka) : (kb);      \
        uint32_t tc_ = sw_ ? (cb) : (ca);   \
        uint32_t td_ = sw_ ? (ca) : (cb);   \
        (ka) = tk_; (kb) = tl_; (ca) = tc_; (cb) = td_; \
    }

// Walks one object's tree. Nearest-hit mode (ANY == false) keeps the reference's acceptance rule
// for 8-wide triangles (src/main.rs:117, src/simd_accel.rs:307-322): not failed, t > 0,
// closest > t; ties on t go to the lower triangle id because the reference scans ids upward
// with a strict compare. `local_best` starts at 0 so that a tie with the incoming `closest`
// (a hit in an earlier object) is rejected, as the strict compare does.
// Shadow mode (ANY == true) returns at the first triangle whose lane of the reference's result
// has a clear sign bit (src/main.rs:199-204). A NaN t is treated as a miss: x86 produces
// NaNs with the sign bit set, gfx950 with it clear.
// One inner node for one ray: tests the four children, pushes the far ones (farthest first) and
// returns the nearest in *next; false when the ray meets none of them.
template <bool STATS, class Stack>
__device__ __forceinline__ bool node_step(const SceneView& sv, uint32_t cur, rt_f3 inv, float nox, float noy, float noz,
                                          float closest, Stack& stk, LaneStats& st, uint32_t* next) {
    const float4* np = reinterpret_cast<const float4*>(sv.nodes + cur);
    // child k = (lo.x lo.y lo.z hi.x) (hi.y hi.z ref pad)
    const float4 a0 = np[0], b0 = np[1], a1 = np[2], b1 = np[3];
    const float4 a2 = np[4], b2 = np[5], a3 = np[6], b3 = np[7];
    if (STATS) {
        st.node += 1;
        st.box += 4;
        st.bytes += 128;
    }
    float k0, k1, k2, k3;
    uint32_t c0 = __float_as_uint(b0.z), c1 = __float_as_uint(b1.z);
    uint32_t c2 = __float_as_uint(b2.z), c3 = __float_as_uint(b3.z);
#define RT_BOX(K, C, A, B)                                                                           \
    {                                                                                                \
        float ax = __builtin_fmaf(A.x, inv.x, nox), bx = __builtin_fmaf(A.w, inv.x, nox);            \
        float ay = __builtin_fmaf(A.y, inv.y, noy), by = __builtin_fmaf(B.x, inv.y, noy);            \
        float az = __builtin_fmaf(A.z, inv.z, noz), bz = __builtin_fmaf(B.y, inv.z, noz);            \
        float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));           \
        float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), closest));        \
        K = ((C) != EMPTY && tn <= tf) ? tn : RT_INF;                                                \
    }
    RT_BOX(k0, c0, a0, b0)
    RT_BOX(k1, c1, a1, b1)
    RT_BOX(k2, c2, a2, b2)
    RT_BOX(k3, c3, a3, b3)
#undef RT_BOX
    RT_CSWAP(k0, k1, c0, c1)
    RT_CSWAP(k2, k3, c2, c3)
    RT_CSWAP(k0, k2, c0, c2)
    RT_CSWAP(k1, k3, c1, c3)
    RT_CSWAP(k1, k2, c1, c2)
    if (k3 < RT_INF) stk.push(c3, k3);
    if (k2 < RT_INF) stk.push(c2, k2);
    if (k1 < RT_INF) stk.push(c1, k1);
    *next = c0;
    return k0 < RT_INF;
}

// node_step on the compact form of the nodes (64 bytes instead of 128: half the vector-memory
// instructions per step, which is what bounds the per-lane walk on incoherent rays). A compact node
// holds its own box corner `origin` and a step `scale` per axis, and every child box as six bytes:
// lo = origin + q_lo * scale, hi = origin + q_hi * scale, rounded outwards by k_quantize_nodes with a
// margin that covers the roundings below. Layout, four float4:
//   0: origin.xyz scale.x   1: scale.yz ref0 ref1   2: ref2 ref3 lo.x hi.x   3: lo.y hi.y lo.z hi.z
// where each of the last six words holds that plane of the four children, child k in byte k. The slab distances come out as
// t = q * (scale * inv) + (origin * inv - o * inv): one conversion and one fused multiply-add each.
#define RT_UB(w, k) ((float)(((w) >> (8 * (k))) & 0xFFu))   // one v_cvt_f32_ubyteK
template <bool STATS, class Stack>
__device__ __forceinline__ bool node_step_q(const float4 q0, const float4 q1, const float4 q2, const float4 q3, rt_f3 inv,
                                            float nox, float noy, float noz, float closest, Stack& stk, LaneStats& st,
                                            uint32_t* next) {
    if (STATS) {
        st.node += 1;
        st.box += 4;
        st.bytes += 64;
    }
    const float sx = q0.w * inv.x, sy = q1.x * inv.y, sz = q1.y * inv.z;
    const float ox = __builtin_fmaf(q0.x, inv.x, nox), oy = __builtin_fmaf(q0.y, inv.y, noy), oz = __builtin_fmaf(q0.z, inv.z, noz);
    uint32_t c0 = __float_as_uint(q1.z), c1 = __float_as_uint(q1.w);
    uint32_t c2 = __float_as_uint(q2.x), c3 = __float_as_uint(q2.y);
    // The ray enters a slab through its low plane when it runs upwards along the axis and through its
    // high plane otherwise: choosing the word of entry planes per axis here (one select each way) saves
    // the minimum and maximum of every pair of slab distances (two operations per child and axis).
    // Where the sign cannot tell (a slab distance that is NaN: 0 * inf), maxima and minima skip the NaN
    // as they did, which keeps the child.
    const bool upx = !__builtin_signbit(inv.x), upy = !__builtin_signbit(inv.y), upz = !__builtin_signbit(inv.z);
    const uint32_t xl = __float_as_uint(q2.z), xh = __float_as_uint(q2.w), yl = __float_as_uint(q3.x);
    const uint32_t yh = __float_as_uint(q3.y), zl = __float_as_uint(q3.z), zh = __float_as_uint(q3.w);
    const uint32_t nx = upx ? xl : xh, fx = upx ? xh : xl;
    const uint32_t ny = upy ? yl : yh, fy = upy ? yh : yl;
    const uint32_t nz = upz ? zl : zh, fz = upz ? zh : zl;
    float k0, k1, k2, k3;
#define RT_QBOX(K, C, B)                                                                              \
    {                                                                                                 \
        const float ax = __builtin_fmaf(RT_UB(nx, B), sx, ox), bx = __builtin_fmaf(RT_UB(fx, B), sx, ox);  \
        const float ay = __builtin_fmaf(RT_UB(ny, B), sy, oy), by = __builtin_fmaf(RT_UB(fy, B), sy, oy);  \
        const float az = __builtin_fmaf(RT_UB(nz, B), sz, oz), bz = __builtin_fmaf(RT_UB(fz, B), sz, oz);  \
        const float tn = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.0f));                                       \
        const float tf = fminf(fminf(bx, by), fminf(bz, closest));                                    \
        K = ((C) != EMPTY && tn <= tf) ? tn : RT_INF;                                                 \
    }
    RT_QBOX(k0, c0, 0)
    RT_QBOX(k1, c1, 1)
    RT_QBOX(k2, c2, 2)
    RT_QBOX(k3, c3, 3)
#undef RT_QBOX
    RT_CSWAP(k0, k1, c0, c1)
    RT_CSWAP(k2, k3, c2, c3)
    RT_CSWAP(k0, k2, c0, c2)
    RT_CSWAP(k1, k3, c1, c3)
    RT_CSWAP(k1, k2, c1, c2)
    if (k3 < RT_INF) stk.push(c3, k3);
    if (k2 < RT_INF) stk.push(c2, k2);
    if (k1 < RT_INF) stk.push(c1, k1);
    *next = c0;
    return k0 < RT_INF;
}
template <bool STATS, class Stack>
__device__ __forceinline__ bool node_step_q(const SceneView& sv, uint32_t cur, rt_f3 inv, float nox, float noy, float noz,
                                            float closest, Stack& stk, LaneStats& st, uint32_t* next) {
    const float4* np = sv.qnodes + 4 * (size_t)cur;
    return node_step_q<STATS>(np[0], np[1], np[2], np[3], inv, nox, noy, noz, closest, stk, st, next);
}

// One triangle record for one ray. Shadow mode returns true if it occludes.
template <bool ANY>
__device__ __forceinline__ bool record_test(float4 ra, float4 rb, float4 rc, uint32_t obj_index, rt_f3 o, rt_f3 d,
                                            float& closest, uint32_t& local_best, HitD& hit) {
    const rt_f3 v0 = f3(ra.x, ra.y, ra.z);
    const rt_f3 e1 = f3(ra.w, rb.x, rb.y);
    const rt_f3 e2 = f3(rb.z, rb.w, rc.x);
    const uint32_t id = __float_as_uint(rc.y);
    const rt_tri_eval ev = tri_intersect(e1, e2, v0, o, d);
    if (ANY) return !ev.failed && !__builtin_signbit(ev.t) && !(ev.t != ev.t);
    tie_note(ev.t, closest, !ev.failed && ev.t > 0.0f, hit.near);
    const bool better = (ev.t < closest) || (ev.t == closest && id < local_best);
    if (!ev.failed && ev.t > 0.0f && better) {
        closest = ev.t;
        local_best = id;
        hit.tri = id;
        hit.obj = obj_index;
        hit.t = ev.t;
        hit.u = ev.u;
        hit.v = ev.v;
    }
    return false;
}

// One leaf for one ray. Shadow mode returns true at the first occluder.
template <bool ANY, bool STATS>
__device__ __forceinline__ bool leaf_step(const SceneView& sv, uint32_t cur, uint32_t obj_index, rt_f3 o, rt_f3 d,
                                          float& closest, uint32_t& local_best, HitD& hit, LaneStats& st) {
    const uint32_t first = (cur & 0x7FFFFFFFu) >> 3;
    const uint32_t n = (cur & 7u) + 1u;
    for (uint32_t i = 0; i < n; ++i) {
        const float4* r = sv.recs + 3 * (size_t)(first + i);
        if (STATS) st.tri += 1, st.bytes += 48;
        if (record_test<ANY>(r[0], r[1], r[2], obj_index, o, d, closest, local_best, hit)) return true;
    }
    return false;
}

template <bool ANY, bool STATS, class Stack>
__device__ __forceinline__ bool bvh_search(const SceneView& sv, uint32_t root, uint32_t obj_index, rt_f3 o,
                                           rt_f3 d, rt_f3 inv, float& closest, HitD& hit, Stack& stk,
                                           LaneStats& st) {
    const float nox = -(o.x * inv.x), noy = -(o.y * inv.y), noz = -(o.z * inv.z);
    uint32_t local_best = 0;
    hit.near = 0xFFFFFFFFu;
    stk.sp = 0;
    uint32_t cur = root;
    for (;;) {
        if (!(cur & LEAF_BIT)) {
            uint32_t next;
            if (node_step<STATS>(sv, cur, inv, nox, noy, noz, closest, stk, st, &next)) {
                cur = next;
                continue;
            }
        } else {
            if (leaf_step<ANY, STATS>(sv, cur, obj_index, o, d, closest, local_best, hit, st)) return true;
        }
        // Next pending node whose entry distance is not beyond the current nearest hit.
        for (;;) {
            if (stk.sp == 0) return false;
            const uint2 e = stk.pop();
            if (!(__uint_as_float(e.y) > closest)) {
                cur = e.x;
                break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The order of the reference's search, where it decides. The reference goes through an object's
// leaves in order, eight at a time, and tests a packet's eight boxes with the ray cut at the nearest
// distance found so far (src/main.rs:98-111,127); a triangle then replaces the nearest only if strictly
// nearer. The walks here return the least (t, id) among the triangles the exact test accepts, W. That
// is the reference's answer whenever the box of W's leaf passes with the ray cut at W's own distance:
// every distance the reference can hold before it reaches W's packet belongs to an accepted triangle,
// so it is at least t_W, the box test only gets easier with a longer ray, and of the triangles at the
// least distance W has the lowest number. It need not be when that test fails: W lies in the very
// plane of its leaf's box (an axis-aligned wall) and the box's entry distance, rounded as the box test
// rounds, comes out a few units of the last place behind W's t, rounded as the triangle test rounds; if
// an earlier-numbered triangle E has its t in between (a coplanar neighbour, an overlapping layer), the
// reference holds t_E when it comes to W's packet, culls the packet and keeps E.
// closest_needs_resolve is the test, run once per ray and object that found a hit; closest_resolve plays
// the reference back for the (rare) rays that fail it.
// ---------------------------------------------------------------------------------------------
// Half the hits on an axis-aligned wall fail the first test by rounding alone. Nothing can differ for them
// unless a second accepted triangle lies between the hit and the entry distance of its leaf's box: none does
// when the walk has noted no tie (hit.tie, tie_note) and the box passes with the ray cut RT_TIE_WALK ulps
// behind the hit.
// (tri_base, leaf_base: the object's, which the callers hold already: one read fewer ahead of the gate's)
__device__ __forceinline__ bool closest_needs_resolve(const SceneView& sv, uint32_t oi, uint32_t tri_base, uint32_t leaf_base, rt_f3 o,
                                                      rt_f3 d, rt_f3 inv, const HitD& hit) {
    const float4* lg = sv.leaf_gate + 2 * (size_t)(leaf_base + ((hit.tri - tri_base) >> 3));
    const float4 lo4 = lg[0], hi4 = lg[1];
    if (ref_leaf_gate(lo4, hi4, o, inv, hit.t)) return false;
    if (hit.near <= RT_TIE_WALK) return true;
    const DevObject& ob = sv.objs[oi];
    // how far behind the hit the walk is sure to have tested everything: half the boxes' padding, as a distance
    // along the ray, and no farther than the ties are noted
    float largest = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) largest = fmaxf(largest, fmaxf(__builtin_fabsf(ob.bbmin[k]), __builtin_fabsf(ob.bbmax[k])));
    const float sure = 0.5f * RT_PAD_ABS * largest / __builtin_sqrtf(g_dot(d, d));
    const float t_far = fminf(hit.t + sure, __uint_as_float(__float_as_uint(hit.t) + RT_TIE_WALK));
    return !ref_leaf_gate(lo4, hi4, o, inv, t_far);
}
__device__ __forceinline__ bool closest_needs_resolve(const SceneView& sv, uint32_t oi, rt_f3 o, rt_f3 d, rt_f3 inv, const HitD& hit) {
    const DevObject& ob = sv.objs[oi];
    return closest_needs_resolve(sv, oi, ob.tri_base, ob.leaf_base, o, d, inv, hit);
}

// These run for one ray in many thousands and are called, not inlined: a called function's registers count
// towards every kernel that calls it, and the values it keeps across its own calls sit in the high
// registers the calling convention preserves, so the ray and its result travel as one block of private
// memory and each function keeps little.
struct ResolveRay {
    rt_f3 o, d;
    float closest;   // in: the nearest distance as the object's walk left it; out: as the reference leaves it
    HitD hit;        // likewise
};

// The packed part of the reference's search of one object (src/main.rs:100-153), r.closest and r.hit as the
// objects before it left them.
__device__ __noinline__ void ref_object_packed(const SceneView& sv, uint32_t oi, ResolveRay& r) {
    const DevObject& ob = sv.objs[oi];
    const rt_f3 o = r.o, d = r.d, inv = g_recip(d);
    float closest = r.closest;
    const uint32_t n_leaves = ob.n_tris / 8;
    for (uint32_t l0 = 0; l0 < n_leaves; l0 += 8u) {
        const float ray_length = closest;   // closest_splat when the packet's boxes are tested
        uint32_t mask = 0;
        for (uint32_t b = 0; b < 8u && l0 + b < n_leaves; ++b) {
            const float4* lg = sv.leaf_gate + 2 * (size_t)(ob.leaf_base + l0 + b);
            if (ref_leaf_gate(lg[0], lg[1], o, inv, ray_length)) mask |= 1u << b;
        }
        for (uint32_t b = 0; b < 8u; ++b) {
            if (!(mask & (1u << b))) continue;
            for (uint32_t j = 0; j < 8u; ++j) {
                const uint32_t id = ob.tri_base + (l0 + b) * 8u + j;
                const float* p = sv.tri9 + 9 * (size_t)id;
                const rt_tri_eval ev = tri_intersect(ld3(p), ld3(p + 3), ld3(p + 6), o, d);
                if (!ev.failed && ev.t > 0.0f && closest > ev.t) {
                    closest = ev.t;
                    r.hit.tri = id, r.hit.obj = oi, r.hit.t = ev.t, r.hit.u = ev.u, r.hit.v = ev.v;
                }
            }
        }
    }
    r.closest = closest;
}

// An object's trailing n % 8 triangles as the reference tests them (src/main.rs:157-169): `closest` stays.
__device__ __noinline__ void ref_object_rest(const SceneView& sv, uint32_t oi, ResolveRay& r) {
    const DevObject& ob = sv.objs[oi];
    const rt_f3 o = r.o, d = r.d;
    const float closest = r.closest;
    for (uint32_t k = 0; k < ob.n_rest; ++k) {
        const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + k);
        float t, u, v;
        if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, closest, &t, &u, &v) && closest > t)
            r.hit.tri = ob.tri_base + ob.n_tris - ob.n_rest + k, r.hit.obj = oi, r.hit.t = t, r.hit.u = u, r.hit.v = v;
    }
}

// The whole of the reference's search up to and including the packed part of object `upto`
// (src/main.rs:93-171): the last resort of closest_resolve.
__device__ __noinline__ void ref_search_upto(const SceneView& sv, uint32_t upto, ResolveRay& r) {
    r.closest = RT_INF;
    r.hit.tri = RT_MISS, r.hit.obj = 0, r.hit.t = r.hit.u = r.hit.v = 0.0f;
    for (uint32_t oi = 0; oi <= upto; ++oi) {
        const DevObject& ob = sv.objs[oi];
        if (!box_intersection_check(r.o, g_recip(r.d), ld3(ob.bbmin), ld3(ob.bbmax))) continue;
        ref_object_packed(sv, oi, r);
        if (oi != upto) ref_object_rest(sv, oi, r);   // (the caller goes on with the last object's trailing triangles)
    }
}

// Entry distance of PackedBoxes::intersect (src/simd_accel.rs:342-370): the tmin of packed_box_hit.
__device__ __forceinline__ float packed_box_entry(rt_f3 bmin, rt_f3 bmax, rt_f3 o, rt_f3 inv) {
    const rt_f3 a = (bmin - o) * inv, b = (bmax - o) * inv;
    const bool kx = signum_is_one(inv.x), ky = signum_is_one(inv.y), kz = signum_is_one(inv.z);
    return x86_max(x86_max(kx ? a.x : b.x, ky ? a.y : b.y), kz ? a.z : b.z);
}

#define RT_RESOLVE_MAX 8
#define RT_RESOLVE_STACK 96    // up to three children left waiting per level of a tree; a deeper one goes to the last resort
struct NearList {
    uint32_t n;
    uint32_t ids[RT_RESOLVE_MAX];
    float ts[RT_RESOLVE_MAX];
};
// Every triangle of the tree under `root` that the exact test accepts with t <= far, in the order of their
// numbers. False when there are more than RT_RESOLVE_MAX (or the tree is deeper than any the host builds).
// The walk of bvh_search with the ray's length fixed, a child at a time.
__device__ __noinline__ bool collect_near(const SceneView& sv, uint32_t root, const ResolveRay& r, float far, NearList& out) {
    uint32_t stack[RT_RESOLVE_STACK];
    uint32_t sp = 0, n = 0;
    uint32_t cur = root;
    const rt_f3 o = r.o, d = r.d, inv = g_recip(d);
    for (;;) {
        if (!(cur & LEAF_BIT)) {
            const float4* np = reinterpret_cast<const float4*>(sv.nodes + cur);
            for (uint32_t k = 0; k < 4u; ++k) {
                const float4 A = np[2 * k], B = np[2 * k + 1];   // (lo.x lo.y lo.z hi.x) (hi.y hi.z ref pad)
                const uint32_t c = __float_as_uint(B.z);
                if (c == EMPTY) continue;
                const float ax = (A.x - o.x) * inv.x, bx = (A.w - o.x) * inv.x;
                const float ay = (A.y - o.y) * inv.y, by = (B.x - o.y) * inv.y;
                const float az = (A.z - o.z) * inv.z, bz = (B.y - o.z) * inv.z;
                // (minima and maxima that skip a NaN, as node_step's do: a child is kept when in doubt)
                const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), far));
                // the walks' boxes are padded for the roundings of fma(plane, inv, -o * inv); this form
                // rounds differently by a few units of the last place, far inside that padding
                if (tn <= tf * 1.00001f + 1e-30f) {
                    if (sp == RT_RESOLVE_STACK) return false;
                    stack[sp++] = c;
                }
            }
        } else {
            const uint32_t first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float4* rec = sv.recs + 3 * (size_t)(first + i);
                const float4 ra = rec[0], rb = rec[1], rc = rec[2];
                const rt_tri_eval ev = tri_intersect(f3(ra.w, rb.x, rb.y), f3(rb.z, rb.w, rc.x), f3(ra.x, ra.y, ra.z), o, d);
                if (ev.failed || !(ev.t > 0.0f) || !(ev.t <= far)) continue;
                if (n == RT_RESOLVE_MAX) return false;
                const uint32_t id = __float_as_uint(rc.y);
                uint32_t k = n++;
                while (k > 0 && out.ids[k - 1] > id) out.ids[k] = out.ids[k - 1], out.ts[k] = out.ts[k - 1], --k;
                out.ids[k] = id, out.ts[k] = ev.t;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    out.n = n;
    return true;
}

// Object oi's search has ended at hit W = (r.hit.tri, r.hit.t) with r.closest = r.hit.t, `before` was the
// nearest distance when it began, and closest_needs_resolve said that the order may matter. Leaves r.hit and
// r.closest as the reference's search of the object's packed triangles leaves them.
// Which triangles can matter: the reference's answer F is an accepted triangle, so t_F >= t_W, and if
// t_F > t_W the reference has culled W's packet with a distance below the entry distance of W's leaf, so
// t_F < entry. The accepted triangles with t up to `far` = max(entry, t_W) are collected by a second walk
// with the ray cut there (a handful), in the order of their numbers, and gone through as the reference
// goes through them. Triangles farther than `far` take part in the reference's search only through the
// nearest distance they leave behind, some value above `far`: a collected triangle's box test gives the
// same answer for every such value unless its own entry distance lies above `far`, and then, or when
// there are too many to collect, or when none of the collected ones is accepted, the reference's search
// is run as it stands (ref_search_upto).
__device__ __forceinline__ const float4* resolve_gate_of(const SceneView& sv, const DevObject& ob, uint32_t id) {
    return sv.leaf_gate + 2 * (size_t)(ob.leaf_base + ((id - ob.tri_base) >> 3));
}
// The nearest hit of object oi among the triangles whose leaf the reference opens at all for this ray (the
// gate with the ray's length at +inf), nearer than `before`: what a walk's answer should have been when its
// winner sits in a leaf that is closed to the ray (a needle whose test rounds badly, a hit on the very edge
// of a wall). The least (t, id) as the walks take it, ties noted as they note them; a child at a time,
// as in collect_near. False when the tree is too deep for the stack.
__device__ __noinline__ bool gated_nearest(const SceneView& sv, uint32_t oi, float before, ResolveRay& r) {
    const DevObject& ob = sv.objs[oi];
    uint32_t stack[RT_RESOLVE_STACK];
    uint32_t sp = 0, best = RT_MISS;
    uint32_t cur = ob.root;
    const rt_f3 o = r.o, d = r.d, inv = g_recip(d);
    float closest = before;
    uint32_t near = 0xFFFFFFFFu;
    for (;;) {
        if (!(cur & LEAF_BIT)) {
            const float4* np = reinterpret_cast<const float4*>(sv.nodes + cur);
            for (uint32_t k = 0; k < 4u; ++k) {
                const float4 A = np[2 * k], B = np[2 * k + 1];
                const uint32_t c = __float_as_uint(B.z);
                if (c == EMPTY) continue;
                const float ax = (A.x - o.x) * inv.x, bx = (A.w - o.x) * inv.x;
                const float ay = (A.y - o.y) * inv.y, by = (B.x - o.y) * inv.y;
                const float az = (A.z - o.z) * inv.z, bz = (B.y - o.z) * inv.z;
                const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), closest));
                if (tn <= tf * 1.00001f + 1e-30f) {
                    if (sp == RT_RESOLVE_STACK) return false;
                    stack[sp++] = c;
                }
            }
        } else {
            const uint32_t first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float4* rec = sv.recs + 3 * (size_t)(first + i);
                const float4 ra = rec[0], rb = rec[1], rc = rec[2];
                const rt_tri_eval ev = tri_intersect(f3(ra.w, rb.x, rb.y), f3(rb.z, rb.w, rc.x), f3(ra.x, ra.y, ra.z), o, d);
                if (ev.failed || !(ev.t > 0.0f)) continue;
                const uint32_t id = __float_as_uint(rc.y);
                const float4* lg = resolve_gate_of(sv, ob, id);
                if (!ref_leaf_gate(lg[0], lg[1], o, inv)) continue;
                tie_note(ev.t, closest, true, near);
                if (ev.t < closest || (ev.t == closest && best != RT_MISS && id < best)) {
                    closest = ev.t, best = id;
                    r.hit.tri = id, r.hit.obj = oi, r.hit.t = ev.t, r.hit.u = ev.u, r.hit.v = ev.v;
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    r.hit.near = near;
    r.closest = closest;
    if (best == RT_MISS) r.hit.tri = RT_MISS;
    return true;
}

// (in three called parts, so that none of them holds much across a call)
// `far` for the hit in r; NaN when the reference never looks into W's leaf (its answer may then lie anywhere behind W).
__device__ __noinline__ float resolve_far(const SceneView& sv, uint32_t oi, const ResolveRay& r) {
    const DevObject& ob = sv.objs[oi];
    const rt_f3 o = r.o, inv = g_recip(r.d);
    const float4* lg = resolve_gate_of(sv, ob, r.hit.tri);
    const float4 lo4 = lg[0], hi4 = lg[1];
    if (!ref_leaf_gate(lo4, hi4, o, inv)) return __uint_as_float(0x7FC00000u);
    const float entry = packed_box_entry(f3(lo4.x, lo4.y, lo4.z), f3(hi4.x, hi4.y, hi4.z), o, inv);
    return entry > r.hit.t ? entry : r.hit.t;   // (an entry that is NaN: t_W)
}
// The collected triangles gone through in the reference's order. False: the reference's search as it stands is needed.
__device__ __noinline__ bool resolve_replay(const SceneView& sv, uint32_t oi, float before, float far, const NearList& near,
                                            ResolveRay& r) {
    const DevObject& ob = sv.objs[oi];
    const rt_f3 o = r.o, inv = g_recip(r.d);
    uint32_t best = RT_MISS;
    float c = before;
    uint32_t packet = 0xFFFFFFFFu;
    float packet_closest = before;
    for (uint32_t k = 0; k < near.n; ++k) {
        const uint32_t pk = (near.ids[k] - ob.tri_base) >> 6;
        if (pk != packet) packet = pk, packet_closest = c;
        const float4* lg = resolve_gate_of(sv, ob, near.ids[k]);
        const float4 lo4 = lg[0], hi4 = lg[1];
        const bool open = ref_leaf_gate(lo4, hi4, o, inv), open_far = ref_leaf_gate(lo4, hi4, o, inv, far);
        if (open != open_far) return false;   // depends on distances beyond `far`
        const bool pass = packet_closest > far ? open_far : ref_leaf_gate(lo4, hi4, o, inv, packet_closest);
        if (pass && c > near.ts[k]) c = near.ts[k], best = near.ids[k];
    }
    if (best == RT_MISS) return false;
    if (best != r.hit.tri) {
        const float* p = sv.tri9 + 9 * (size_t)best;
        const rt_tri_eval ev = tri_intersect(ld3(p), ld3(p + 3), ld3(p + 6), r.o, r.d);
        r.hit.tri = best, r.hit.obj = oi, r.hit.t = ev.t, r.hit.u = ev.u, r.hit.v = ev.v;
        r.closest = ev.t;
    }
    return true;
}
__device__ __noinline__ void closest_resolve(const SceneView& sv, uint32_t oi, float before, ResolveRay& r) {
    float far = resolve_far(sv, oi, r);
    bool done = false;
    if (far != far && oi == 0u) {
        // W's leaf is closed to the ray. In the first object (nothing before it to fall back on) the search is
        // done over without such triangles; its winner is then looked at as any walk's is.
        if (gated_nearest(sv, oi, before, r)) {
            if (r.hit.tri == RT_MISS) return;
            const rt_f3 inv = g_recip(r.d);
            if (!closest_needs_resolve(sv, oi, r.o, r.d, inv, r.hit)) return;
            far = resolve_far(sv, oi, r);
        }
    }
    if (far == far) {
        NearList near;
        near.n = 0;
        done = collect_near(sv, sv.objs[oi].root, r, far, near) && resolve_replay(sv, oi, before, far, near, r);
    }
    if (!done) ref_search_upto(sv, oi, r);
}
// The call, from a walk that has just ended object oi's tree with a nearer hit than `before`.
__device__ __forceinline__ void closest_confirm(const SceneView& sv, uint32_t oi, rt_f3 o, rt_f3 d, rt_f3 inv, float before,
                                                HitD& hit, float& closest) {
    if (!closest_needs_resolve(sv, oi, o, d, inv, hit)) return;
    ResolveRay r;
    r.o = o, r.d = d, r.closest = closest, r.hit = hit;
    closest_resolve(sv, oi, before, r);
    hit = r.hit, closest = r.closest;
}

// trace_ray's nearest-hit search over the whole World, src/main.rs:93-171.
template <bool STATS, class Stack = LaneStack>
__device__ __forceinline__ HitD closest_all(const SceneView& sv, rt_f3 o, rt_f3 d, Stack& stk,
                                            LaneStats& st) {
    const rt_f3 inv = g_recip(d);
    HitD h;
    h.tri = RT_MISS;
    h.obj = 0;
    h.t = h.u = h.v = 0.0f;
    float closest = RT_INF;
    if (STATS) st.rays += 1;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        const DevObject& ob = sv.objs[oi];
        if (STATS) st.box += 1;
        if (!box_intersection_check(o, inv, ld3(ob.bbmin), ld3(ob.bbmax))) continue;
        if (ob.root != EMPTY) {
            const float before = closest;
            bvh_search<false, STATS>(sv, ob.root, oi, o, d, inv, closest, h, stk, st);
            if (closest < before) closest_confirm(sv, oi, o, d, inv, before, h, closest);
        }
        // Trailing n % 8 triangles, src/main.rs:157-169: scalar arithmetic, and `closest` is NOT
        // lowered by a hit here, so the last qualifying one wins and later objects still compare
        // against the old distance.
        for (uint32_t r = 0; r < ob.n_rest; ++r) {
            const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
            float t, u, v;
            if (STATS) st.tri += 1;
            if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, closest, &t, &u, &v)) {
                if (closest > t) {
                    h.tri = ob.tri_base + ob.n_tris - ob.n_rest + r;
                    h.obj = oi;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                }
            }
        }
    }
    h.closest = closest;
    return h;
}

// trace_ray's shadow search, src/main.rs:185-233. True when anything occludes.
template <bool STATS>
__device__ __forceinline__ bool any_all(const SceneView& sv, rt_f3 o, rt_f3 d, LaneStack& stk, LaneStats& st) {
    const rt_f3 inv = g_recip(d);
    if (STATS) st.rays += 1;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        const DevObject& ob = sv.objs[oi];
        if (STATS) st.box += 1;
        if (!box_intersection_check(o, inv, ld3(ob.bbmin), ld3(ob.bbmax))) continue;
        if (ob.root != EMPTY) {
            float closest = RT_INF;
            HitD dummy;
            if (bvh_search<true, STATS>(sv, ob.root, oi, o, d, inv, closest, dummy, stk, st)) return true;
        }
        for (uint32_t r = 0; r < ob.n_rest; ++r) {
            const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
            float t, u, v;
            if (STATS) st.tri += 1;
            if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, RT_INF, &t, &u, &v)) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Wave-coherent ("packet") traversal: the 64 rays of a wave walk one object's tree together.
// The node is fetched once per wave through the scalar cache (uniform address, constant address
// space -> s_load), each lane tests its own ray against the four child boxes with the node held
// in SGPRs, and wave ballots decide which children the wave visits. The pending-node stack is
// shared by the wave and lives in two VGPRs (entry i in lane i), so there is no LDS or memory
// traffic for it. Every lane tests every triangle of every leaf the wave reaches whose box its
// own ray hits; results per lane are the same as in the per-lane walk because the nearest hit
// does not depend on the visiting order.
// Meant for rays that start close together and point the same way (camera rays of neighbouring
// pixels, their sun shadow rays); correct for any rays, slower than the per-lane walk for
// incoherent ones.
// ---------------------------------------------------------------------------------------------
#define RT_CONST __attribute__((address_space(4)))
// Declared with 4-byte alignment: triangle records (48 B) and trailing-triangle vertices (36 B)
// are not multiples of the vector size, and scalar loads only need dword alignment.
typedef float f32x4_t __attribute__((ext_vector_type(4))) __attribute__((aligned(4)));
typedef float f32x8_t __attribute__((ext_vector_type(8))) __attribute__((aligned(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16))) __attribute__((aligned(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ f32x4_t sload4(const void* p) { return *(const RT_CONST f32x4_t*)p; }
__device__ __forceinline__ f32x8_t sload8(const void* p) { return *(const RT_CONST f32x8_t*)p; }
__device__ __forceinline__ f32x16_t sload16(const void* p) { return *(const RT_CONST f32x16_t*)p; }
__device__ __forceinline__ uint32_t sload1u(const void* p) { return *(const RT_CONST uint32_t*)p; }
#pragma clang diagnostic pop

#define RT_INF_BITS 0x7f800000
// Shadow rays: stop a triangle test after its first coordinate when that fails for every lane.
#ifndef RT_ANY_EARLY_OUT
#define RT_ANY_EARLY_OUT 1
#endif
#define RT_BALLOT(pred) __ballot(pred)


// Wave-shared stack of at most 128 node references: entry i is lane (i & 63) of r0 / r1.
struct WaveStack {
    int r0 = 0, r1 = 0;
    int sp = 0;  // wave-uniform
    __device__ __forceinline__ void push(int lane, uint32_t v) {
        if (sp < 64) r0 = (lane == sp) ? (int)v : r0;
        else r1 = (lane == sp - 64) ? (int)v : r1;
        sp = sp < 2 * 64 - 1 ? sp + 1 : sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return (uint32_t)(sp < 64 ? __builtin_amdgcn_readlane(r0, sp) : __builtin_amdgcn_readlane(r1, sp - 64));
    }
};

#define RT_SSWAP(ka, kb, ca, cb)  \
    if ((ka) > (kb)) {            \
        int tk_ = (ka);           \
        (ka) = (kb);              \
        (kb) = tk_;               \
        uint32_t tc_ = (ca);      \
        (ca) = (cb);              \
        (cb) = tc_;               \
    }

// One triangle record (wave-uniform, in scalar registers) against every lane with `pred` set.
template <bool ANY, bool STATS>
__device__ __forceinline__ void packet_tri(const f32x8_t ra, const f32x4_t rb, uint32_t obj_index, bool& pred, rt_f3 o,
                                           rt_f3 d, float& closest, uint32_t& local_best, HitD& hit, bool& active,
                                           bool& occluded, LaneStats& st, bool sv_bounded) {
    {
        const rt_f3 v0 = f3(ra[0], ra[1], ra[2]);
        const rt_f3 e1 = f3(ra[3], ra[4], ra[5]);
        const rt_f3 e2 = f3(ra[6], ra[7], rb[0]);
        const uint32_t id = __float_as_uint(rb[1]);
        if (STATS && pred) st.tri += 1;
        if (STATS && (threadIdx.x & 63u) == 0u) st.bytes += 48;
        if (ANY) {
            // Shadow rays: most triangles a wave fetches hit none of its rays, and most of those fail
            // on the first coordinate for every lane: the wave then skips the division and the rest.
            const rt_tri_head hd = tri_begin(e1, e2, v0, o, d);
            if (RT_ANY_EARLY_OUT && sv_bounded && __ballot(pred && !tri_head_fails(hd)) == 0ull) return;
            const rt_tri_eval ev = tri_finish(hd, e1, e2, d);
            if (pred && !ev.failed && !__builtin_signbit(ev.t) && !(ev.t != ev.t)) {
                occluded = true;
                active = false;
                pred = false;
            }
        } else {
            const rt_tri_eval ev = tri_intersect(e1, e2, v0, o, d);
            const bool accepted = pred && !ev.failed && ev.t > 0.0f;
            tie_note(ev.t, closest, accepted, hit.near);
            const bool better = (ev.t < closest) || (ev.t == closest && id < local_best);
            if (accepted && better) {
                closest = ev.t;
                local_best = id;
                hit.tri = id;
                hit.obj = obj_index;
                hit.t = ev.t;
                hit.u = ev.u;
                hit.v = ev.v;
            }
        }
    }
}

// One leaf (wave-uniform reference) against every lane with `pred` set. With `preloaded`, ra0 / rb0
// hold the leaf's first record already.
template <bool ANY, bool STATS>
__device__ __forceinline__ void packet_leaf(const SceneView& sv, uint32_t ref, uint32_t obj_index, bool pred, rt_f3 o,
                                            rt_f3 d, float& closest, uint32_t& local_best, HitD& hit, bool& active,
                                            bool& occluded, LaneStats& st, bool preloaded = false,
                                            f32x8_t ra0 = f32x8_t{}, f32x4_t rb0 = f32x4_t{}) {
    const uint32_t first = (ref & 0x7FFFFFFFu) >> 3;
    const uint32_t n = (ref & 7u) + 1u;
    // Products of coordinate differences stay far from overflow (see tri_head_fails).
    const bool bounded = sv.scene_extent < 1.0e12f;
    for (uint32_t i = 0; i < n; ++i) {
        f32x8_t ra = ra0;
        f32x4_t rb = rb0;
        if (!(preloaded && i == 0)) {
            const char* rp = reinterpret_cast<const char*>(sv.recs) + 48 * (size_t)(first + i);
            ra = sload8(rp);
            rb = sload4(rp + 32);
        }
        packet_tri<ANY, STATS>(ra, rb, obj_index, pred, o, d, closest, local_best, hit, active, occluded, st, bounded);
    }
}

// One leaf (wave-uniform reference) in shadow mode, with the wave's unoccluded rays kept as one
// 64-bit mask instead of a flag per lane: every lane runs the test, a ballot says which rays the
// triangle occludes, and the mask loses them. (Per-lane flags that change inside the walk's loops
// cost a merge of three scalar instructions at every join of the control flow.)
template <bool STATS>
__device__ __forceinline__ void leaf_any(const SceneView& sv, uint32_t ref, rt_f3 o, rt_f3 d, unsigned long long& alive,
                                         LaneStats& st) {
    const uint32_t first = (ref & 0x7FFFFFFFu) >> 3;
    const uint32_t n = (ref & 7u) + 1u;
    const bool bounded = sv.scene_extent < 1.0e12f;   // see tri_head_fails
    for (uint32_t i = 0; i < n; ++i) {
        const char* rp = reinterpret_cast<const char*>(sv.recs) + 48 * (size_t)(first + i);
        const f32x8_t ra = sload8(rp);
        const f32x4_t rb = sload4(rp + 32);
        const rt_f3 v0 = f3(ra[0], ra[1], ra[2]), e1 = f3(ra[3], ra[4], ra[5]), e2 = f3(ra[6], ra[7], rb[0]);
        if (STATS) {
            if ((alive >> (threadIdx.x & 63u)) & 1ull) st.tri += 1;
            if ((threadIdx.x & 63u) == 0u) st.bytes += 48;
        }
        const rt_tri_head hd = tri_begin(e1, e2, v0, o, d);
        if (RT_ANY_EARLY_OUT && bounded && (__ballot(!tri_head_fails(hd)) & alive) == 0ull) continue;
        const rt_tri_eval ev = tri_finish(hd, e1, e2, d);
        // src/main.rs:185-193: any lane whose t has a clear sign bit occludes
        alive &= ~__ballot(!ev.failed && !__builtin_signbit(ev.t) && !(ev.t != ev.t));
        if (alive == 0ull) return;
    }
}

// Walks one object's tree for the whole wave. `active` marks the lanes whose rays take part.
// Acceptance rules are those of bvh_search.
template <bool ANY, bool STATS>
__device__ __forceinline__ void packet_search(const SceneView& sv, uint32_t root, uint32_t obj_index, rt_f3 o, rt_f3 d,
                                              rt_f3 inv, bool active, float& closest, HitD& hit, bool& occluded,
                                              LaneStats& st) {
    const int lane = (int)(threadIdx.x & 63u);
    const float nox = -(o.x * inv.x), noy = -(o.y * inv.y), noz = -(o.z * inv.z);
    uint32_t local_best = 0;
    hit.near = 0xFFFFFFFFu;
    WaveStack stk;
    uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)root);
    for (;;) {
        if (ANY) {
            if (RT_BALLOT(active) == 0ull) break;
        }
        const char* np = reinterpret_cast<const char*>(sv.nodes) + 128 * (size_t)cur;
        const f32x16_t A = sload16(np);
        const f32x16_t B = sload16(np + 64);
        if (STATS && active) {
            st.node += 1;
            st.box += 4;
        }
        if (STATS && (threadIdx.x & 63u) == 0u) {
            st.bytes += 128;
            st.walks[3] += 1;
        }
        float tn0, tn1, tn2, tn3;
        bool h0, h1, h2, h3;
        uint32_t c0 = __float_as_uint(A[6]), c1 = __float_as_uint(A[14]);
        uint32_t c2 = __float_as_uint(B[6]), c3 = __float_as_uint(B[14]);
#define RT_PBOX(TN, H, C, V, O)                                                                          \
    {                                                                                                    \
        float ax = __builtin_fmaf(V[O + 0], inv.x, nox), bx = __builtin_fmaf(V[O + 3], inv.x, nox);      \
        float ay = __builtin_fmaf(V[O + 1], inv.y, noy), by = __builtin_fmaf(V[O + 4], inv.y, noy);      \
        float az = __builtin_fmaf(V[O + 2], inv.z, noz), bz = __builtin_fmaf(V[O + 5], inv.z, noz);      \
        TN = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));                     \
        float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), closest));            \
        H = active && (C) != EMPTY && TN <= tf;                                                          \
    }
        RT_PBOX(tn0, h0, c0, A, 0)
        RT_PBOX(tn1, h1, c1, A, 8)
        RT_PBOX(tn2, h2, c2, B, 0)
        RT_PBOX(tn3, h3, c3, B, 8)
#undef RT_PBOX
        // Leaves first: they are cheap and shorten `closest` before any subtree is entered.
        if ((c0 & LEAF_BIT) && c0 != EMPTY && RT_BALLOT(h0) != 0ull)
            packet_leaf<ANY, STATS>(sv, c0, obj_index, h0, o, d, closest, local_best, hit, active, occluded, st);
        if ((c1 & LEAF_BIT) && c1 != EMPTY && RT_BALLOT(h1) != 0ull)
            packet_leaf<ANY, STATS>(sv, c1, obj_index, h1 && active, o, d, closest, local_best, hit, active, occluded, st);
        if ((c2 & LEAF_BIT) && c2 != EMPTY && RT_BALLOT(h2) != 0ull)
            packet_leaf<ANY, STATS>(sv, c2, obj_index, h2 && active, o, d, closest, local_best, hit, active, occluded, st);
        if ((c3 & LEAF_BIT) && c3 != EMPTY && RT_BALLOT(h3) != 0ull)
            packet_leaf<ANY, STATS>(sv, c3, obj_index, h3 && active, o, d, closest, local_best, hit, active, occluded, st);
        // Inner children some lane still needs, keyed by the entry distance of the first such lane.
        int k0 = RT_INF_BITS, k1 = RT_INF_BITS, k2 = RT_INF_BITS, k3 = RT_INF_BITS;
        {
            unsigned long long m;
            if (!(c0 & LEAF_BIT) && (m = RT_BALLOT(h0 && active && tn0 <= closest)) != 0ull)
                k0 = __builtin_amdgcn_readlane(__float_as_int(tn0), __builtin_ctzll(m));
            if (!(c1 & LEAF_BIT) && (m = RT_BALLOT(h1 && active && tn1 <= closest)) != 0ull)
                k1 = __builtin_amdgcn_readlane(__float_as_int(tn1), __builtin_ctzll(m));
            if (!(c2 & LEAF_BIT) && (m = RT_BALLOT(h2 && active && tn2 <= closest)) != 0ull)
                k2 = __builtin_amdgcn_readlane(__float_as_int(tn2), __builtin_ctzll(m));
            if (!(c3 & LEAF_BIT) && (m = RT_BALLOT(h3 && active && tn3 <= closest)) != 0ull)
                k3 = __builtin_amdgcn_readlane(__float_as_int(tn3), __builtin_ctzll(m));
        }
        // Entry distances are >= 0, so their bit patterns order like integers.
        RT_SSWAP(k0, k1, c0, c1)
        RT_SSWAP(k2, k3, c2, c3)
        RT_SSWAP(k0, k2, c0, c2)
        RT_SSWAP(k1, k3, c1, c3)
        RT_SSWAP(k1, k2, c1, c2)
        if (k3 < RT_INF_BITS) stk.push(lane, c3);
        if (k2 < RT_INF_BITS) stk.push(lane, c2);
        if (k1 < RT_INF_BITS) stk.push(lane, c1);
        if (k0 < RT_INF_BITS) {
            cur = c0;
            continue;
        }
        if (stk.sp == 0) break;
        cur = stk.pop();
    }
}

// ---------------------------------------------------------------------------------------------
// Beam walk: for a wave whose rays all point into the same octant, inner nodes are tested once per
// wave against the beam that bounds the wave's rays (origin interval and reciprocal-direction
// interval per axis), with ONE LANE PER CHILD of a wide node instead of one lane per ray. The
// test is conservative (interval arithmetic plus a relative margin), so the beam reaches every
// leaf some ray of the wave could reach; at a leaf every lane tests its own ray against the
// leaf's triangles with the exact arithmetic, as in the other walks.
// Pending children of each level stay in registers (current level) or LDS (outer levels) as
// (key, reference) per lane; the nearest pending child is found with a DPP wave-minimum, and a
// child is dropped without a visit once its entry distance lies beyond every ray's nearest hit.
// ---------------------------------------------------------------------------------------------
// Wave reductions with DPP-fused VALU ops: row_shr:1,2,4,8 leave each row's result in its last
// lane, row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3) carry it to lane 63. A lane whose DPP
// source is out of range is skipped by the hardware and keeps its value. The s_nop 1 between the
// steps are the two wait states a DPP read needs after a VALU write of the same register.
#define RT_DPP_REDUCE(OP)                                                       \
    "s_nop 1\n\t"                                                               \
    OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
    OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
    OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
    OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"        \
    OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"     \
    OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    asm volatile(RT_DPP_REDUCE("v_min_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_min_f32(float v) {
    asm volatile(RT_DPP_REDUCE("v_min_f32_dpp") : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    asm volatile(RT_DPP_REDUCE("v_max_f32_dpp") : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef RT_DPP_REDUCE
// Four reductions at once (minimum of a and c, maximum of b and d): the four chains are independent,
// so each step's four instructions stand between a register's write and its next DPP read and no
// wait states have to be spent on s_nop.
#define RT_DPP4(STEP)                                     \
    "v_min_f32_dpp %0, %0, %0 " STEP "\n\t"               \
    "v_max_f32_dpp %1, %1, %1 " STEP "\n\t"               \
    "v_min_f32_dpp %2, %2, %2 " STEP "\n\t"               \
    "v_max_f32_dpp %3, %3, %3 " STEP "\n\t"
__device__ __forceinline__ void wave_minmax4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t"
                 RT_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf")
                 RT_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 RT_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf")
                 RT_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 RT_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RT_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
    d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}
#undef RT_DPP4

// The beam of a wave, wave-uniform. Rays are written along their dominant axis w (the axis with
// the largest direction component, mirrored so that every ray travels towards +w): on each of
// the two lateral axes a, a ray is the line  coord_a = A + B * w  with B = d_a / d_w and
// A = o_a - B * o_w. The beam keeps the intervals of A and B over the wave's rays. This form is
// tight both for rays leaving one point (camera rays: A = 0, B spans the pixel footprint) and for
// parallel rays starting anywhere along a line (sun shadow rays: B is one value and A barely
// moves, however far apart the origins are in depth).
struct Beam {
    float sgn;        // +1 or -1: w = sgn * coord_M (M = dominant axis, a template parameter)
    float a0[2], a1[2], b0[2], b1[2];  // over the lateral axes (M+1)%3... see lat_axis
    float wbase;      // w of the first origin when the walk started: keys are w - wbase
    float wstart;     // smallest origin w among the lanes the beam covers
    float wfar;       // largest origin w among them
};

#define RT_MARGIN 9.5367431640625e-7f  // 2^-20
#ifndef RT_ANY_FIRST_PENDING
#define RT_ANY_FIRST_PENDING 1
#endif
#ifndef RT_ANY_PREFER_FAR
#define RT_ANY_PREFER_FAR 1
#endif
#define RT_MAX_SLOPE 8.0f

template <int K>
__device__ __forceinline__ float comp(const rt_f3& v) { return K == 0 ? v.x : (K == 1 ? v.y : v.z); }
__device__ __forceinline__ float pick3(const rt_f3& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
// Lateral axes of dominant axis M: (1,2), (0,2), (0,1).
template <int M, int K>
struct lat_axis {
    static constexpr int value = K == 0 ? (M == 0 ? 1 : 0) : (M == 2 ? 1 : 2);
};

// Per-lane line coefficients of the beam form.
struct BeamLane {
    float ow, dw;  // origin and direction along w (dw > 0)
};

// Dominant axis of the first active lane's direction (wave-uniform), and its sign.
__device__ __forceinline__ int beam_axis(rt_f3 d, unsigned long long act, float* sgn) {
    const int l0 = __builtin_ctzll(act);
    const float dx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.x), l0));
    const float dy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.y), l0));
    const float dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.z), l0));
    const float ax = __builtin_fabsf(dx), ay = __builtin_fabsf(dy), az = __builtin_fabsf(dz);
    const int m = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    *sgn = (m == 0 ? dx : (m == 1 ? dy : dz)) < 0.0f ? -1.0f : 1.0f;
    return m;
}

// Builds the beam of the active lanes along dominant axis M (b.sgn set by the caller). Returns
// false (wave-uniform) when some ray does not travel towards +w, a lateral slope exceeds
// RT_MAX_SLOPE, a value is not finite, or the beam is too wide to be worth a beam walk; the caller
// then uses packet_search. `first` = false keeps wbase of an earlier build (a refit over fewer lanes).
template <int M>
__device__ __forceinline__ bool make_beam(const SceneView& sv, rt_f3 o, rt_f3 d, bool active, Beam& b, BeamLane& bl,
                                          bool first) {
    bl.ow = b.sgn * comp<M>(o);
    bl.dw = b.sgn * comp<M>(d);
    bool lane_ok = (bl.dw > 0.0f) & (bl.dw < RT_INF) & (__builtin_fabsf(bl.ow) < RT_INF);
    const float rdw = 1.0f / bl.dw;
    {
        const float oa = comp<lat_axis<M, 0>::value>(o), da = comp<lat_axis<M, 0>::value>(d);
        const float ob = comp<lat_axis<M, 1>::value>(o), db = comp<lat_axis<M, 1>::value>(d);
        const float B0 = da * rdw, B1 = db * rdw;
        const float A0 = oa - B0 * bl.ow, A1 = ob - B1 * bl.ow;
        lane_ok = lane_ok & (__builtin_fabsf(B0) <= RT_MAX_SLOPE) & (__builtin_fabsf(A0) < RT_INF) &
                  (__builtin_fabsf(B1) <= RT_MAX_SLOPE) & (__builtin_fabsf(A1) < RT_INF);
        // rounding of B and of A = oa - B * ow, plus a relative cushion that also covers the
        // roundings of the node test (products and sums, 2^-24 relative each)
        const float eB0 = __builtin_fabsf(B0) * (4.0f * RT_MARGIN), eB1 = __builtin_fabsf(B1) * (4.0f * RT_MARGIN);
        const float eA0 = (__builtin_fabsf(oa) + __builtin_fabsf(B0 * bl.ow)) * (4.0f * RT_MARGIN) + 1e-30f;
        const float eA1 = (__builtin_fabsf(ob) + __builtin_fabsf(B1 * bl.ow)) * (4.0f * RT_MARGIN) + 1e-30f;
        const bool use = active & lane_ok;
        float q0 = use ? A0 - eA0 : RT_INF, q1 = use ? A0 + eA0 : -RT_INF, q2 = use ? B0 - eB0 : RT_INF,
              q3 = use ? B0 + eB0 : -RT_INF;
        wave_minmax4(q0, q1, q2, q3);
        b.a0[0] = q0, b.a1[0] = q1, b.b0[0] = q2, b.b1[0] = q3;
        q0 = use ? A1 - eA1 : RT_INF, q1 = use ? A1 + eA1 : -RT_INF, q2 = use ? B1 - eB1 : RT_INF,
        q3 = use ? B1 + eB1 : -RT_INF;
        wave_minmax4(q0, q1, q2, q3);
        b.a0[1] = q0, b.a1[1] = q1, b.b0[1] = q2, b.b1[1] = q3;
    }
    float ws = active ? bl.ow : RT_INF, wf = active ? bl.ow : -RT_INF, ws2 = ws, wf2 = wf;
    wave_minmax4(ws, wf, ws2, wf2);
    b.wstart = ws - __builtin_fabsf(ws) * RT_MARGIN;
    b.wfar = wf;
    if (first) b.wbase = b.wstart;
    // A beam much wider than the geometry it meets makes every lane test every triangle of a
    // large region (rays that happen to share a wave but not a neighbourhood): leave those waves
    // to the per-ray walk. Footprint = spread of the lines at w = 0 plus their fanning-out
    // across the scene.
    bool thin = true;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        thin = thin && ((b.a1[k] - b.a0[k]) + (b.b1[k] - b.b0[k]) * sv.scene_extent) <= sv.beam_limit;
    return thin && __ballot(active && !lane_ok) == 0ull;
}

// Tests the children of wide node `ref` (one per lane) against the beam, between the beam's start
// and wend_rel (the farthest w, relative to wbase, at which any ray can still gain a hit).
// key = entry w relative to wbase with the lane number in its low six bits (0xFFFFFFFF: no
// visit), cref = the child's reference.
template <int M, bool STATS, bool ANYMODE = false>
__device__ __forceinline__ void beam_node(const SceneView& sv, uint32_t ref, const Beam& b, float wend_rel,
                                          uint32_t lane, uint32_t& key, uint32_t& cref, LaneStats& st) {
    const uint32_t first = ref >> 6, count = (ref & 63u) + 1u;
    key = 0xFFFFFFFFu;
    cref = 0;
    if (STATS && lane == 0) {
        st.node += 1;
        st.box += count;
        st.bytes += 32 * count;
    }
    if (lane < count) {
        const float4* cp = reinterpret_cast<const float4*>(sv.wide + first + lane);
        const float4 A = cp[0], B = cp[1];  // (lo.x lo.y lo.z hi.x) (hi.y hi.z ref pad)
        const rt_f3 lo = f3(A.x, A.y, A.z), hi = f3(A.w, B.x, B.y);
        const float lm = comp<M>(lo), hm = comp<M>(hi);
        const float z0 = fmaxf(b.sgn > 0.0f ? lm : -hm, b.wstart);
        const float z1 = fminf(b.sgn > 0.0f ? hm : -lm, b.wbase + wend_rel);
        bool hit = z0 <= z1;
        {
            const float p00 = b.b0[0] * z0, p01 = b.b0[0] * z1, p10 = b.b1[0] * z0, p11 = b.b1[0] * z1;
            const float blo = b.a0[0] + fminf(fminf(p00, p01), fminf(p10, p11));
            const float bhi = b.a1[0] + fmaxf(fmaxf(p00, p01), fmaxf(p10, p11));
            hit = hit & (blo <= comp<lat_axis<M, 0>::value>(hi)) & (bhi >= comp<lat_axis<M, 0>::value>(lo));
        }
        {
            const float p00 = b.b0[1] * z0, p01 = b.b0[1] * z1, p10 = b.b1[1] * z0, p11 = b.b1[1] * z1;
            const float blo = b.a0[1] + fminf(fminf(p00, p01), fminf(p10, p11));
            const float bhi = b.a1[1] + fmaxf(fmaxf(p00, p01), fmaxf(p10, p11));
            hit = hit & (blo <= comp<lat_axis<M, 1>::value>(hi)) & (bhi >= comp<lat_axis<M, 1>::value>(lo));
        }
        key = hit ? ((__float_as_uint(fmaxf(z0 - b.wbase, 0.0f)) & ~63u) | lane) : 0xFFFFFFFFu;
#if RT_ANY_PREFER_FAR
        // Shadow mode reads the key only as "pending" (any value but all ones) and "preferred" (0):
        // a child that starts beyond every origin can occlude every ray of the wave.
        if (ANYMODE) key = hit ? (((b.sgn > 0.0f ? lm : -hm) >= b.wfar) ? 0u : 1u) : 0xFFFFFFFFu;
#endif
        cref = __float_as_uint(B.z);
    }
}

// Beam walk of one object's wide tree along dominant axis M. Returns false without doing
// anything when the wave's rays do not form a usable beam.
template <bool ANY, bool STATS, int M>
__device__ __forceinline__ bool beam_search(const SceneView& sv, uint32_t root_ref, uint32_t obj_index, rt_f3 o, rt_f3 d,
                                            bool active, float sgn, float& closest, HitD& hit, bool& occluded,
                                            uint2* lds, LaneStats& st) {
    Beam b;
    BeamLane bl;
    b.sgn = sgn;
    if (!make_beam<M>(sv, o, d, active, b, bl, true)) {
        if (STATS && (threadIdx.x & 63u) == 0u) st.walks[1] += 1;
        return false;
    }
    const uint32_t lane = threadIdx.x & 63u;
    if (STATS && lane == 0) st.walks[0] += 1;
    uint32_t local_best = 0;
    hit.near = 0xFFFFFFFFu;
    // Farthest useful w per lane, relative to wbase: where the ray's current nearest hit lies.
    auto far_rel = [&](float c) {
        const float z = __builtin_fmaf(c, bl.dw, bl.ow);
        return (z + __builtin_fabsf(z) * (4.0f * RT_MARGIN)) - b.wbase;
    };
    float zc = ANY ? RT_INF : far_rel(closest);
    float wend_rel = ANY ? RT_INF : wave_max_f32(active ? zc : -RT_INF);
    // Shadow mode keeps the rays still without an occluder as one mask (see leaf_any).
    unsigned long long alive = ANY ? __ballot(active) : 0ull;
    int n_beam = __builtin_popcountll(__ballot(active));  // lanes the current beam was built from
    int level = 0;
    uint32_t key = 0xFFFFFFFFu, cref = 0;
    // `expand` is the inner node to open next. Its children are tested against the beam, its leaf
    // children are tested against the rays at once (in lane order: the result does not depend on
    // the order, and a leaf costs far less this way than as one more turn of the ordered walk),
    // its inner children become the pending set of a new level.
    uint32_t expand = root_ref;
    bool running = true;
    while (running) {
        if (expand != 0xFFFFFFFFu) {
            // The level's remaining children are kept only if there are any.
            if (__ballot(key != 0xFFFFFFFFu) != 0ull) {
                lds[level * 64 + lane] = make_uint2(key, cref);
                ++level;
            }
            beam_node<M, STATS, ANY>(sv, expand, b, wend_rel, lane, key, cref, st);
            expand = 0xFFFFFFFFu;
            const bool leaf_child = (cref & LEAF_BIT) != 0u;
            unsigned long long lm = __ballot(key != 0xFFFFFFFFu && leaf_child);
            bool improved = false;
            while (lm != 0ull) {
                const int l = __builtin_ctzll(lm);
                lm &= lm - 1ull;
                if (ANY) {
                    if (alive == 0ull) break;
                    leaf_any<STATS>(sv, (uint32_t)__builtin_amdgcn_readlane((int)cref, l), o, d, alive, st);
                    continue;
                } else {
                    const float kf = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)key, l) & ~63u);
                    if (__ballot(active && zc > kf) == 0ull) continue;
                }
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)cref, l);
                const float before = closest;
                packet_leaf<ANY, STATS>(sv, r, obj_index, active, o, d, closest, local_best, hit, active, occluded, st);
                if (!ANY) {
                    if (__ballot(closest < before) != 0ull) {
                        zc = far_rel(closest);
                        improved = true;
                    }
                }
            }
            if (leaf_child) key = 0xFFFFFFFFu;
            if (!ANY && improved) wend_rel = wave_max_f32(active ? zc : -RT_INF);
        }
        uint32_t kmin;
        if (ANY && RT_ANY_FIRST_PENDING) {
            // Any occluder will do, so the order of the visits is free: take the first pending lane
            // (one ballot) instead of the nearest child (a wave-wide minimum); children that start
            // beyond every origin first.
            const unsigned long long pm = __ballot(key != 0xFFFFFFFFu);
#if RT_ANY_PREFER_FAR
            const unsigned long long pp = __ballot(key == 0u);
            kmin = pm ? (uint32_t)__builtin_ctzll(pp ? pp : pm) : 0xFFFFFFFFu;
#else
            kmin = pm ? (uint32_t)__builtin_ctzll(pm) : 0xFFFFFFFFu;
#endif
        } else {
            kmin = wave_min_u32(key);
        }
        const float kf = __uint_as_float(kmin & ~63u);
        // Lanes that can still gain from pending children: all unoccluded lanes in shadow mode;
        // in nearest-hit mode those whose nearest hit lies beyond the next child's entry (the
        // walk is front to back along w, so the others are finished for this level).
        const unsigned long long rel = ANY ? alive : __ballot(active && zc > kf);
        if (ANY && rel == 0ull) {
            running = false;
        } else if (kmin == 0xFFFFFFFFu || (!ANY && rel == 0ull)) {
            // Nothing left at this level: back to the level above.
            if (level == 0) {
                running = false;
            } else {
                --level;
                const uint2 e = lds[level * 64 + lane];
                key = e.x;
                cref = e.y;
            }
        } else {
            // Shadow mode: shrink the beam to the unoccluded lanes once half of them have dropped
            // out; the children tested from here on are culled against the thinner beam. (An
            // occluded lane is finished for good. In nearest-hit mode a lane that is finished for
            // this level can still be needed by a nearer child pending further out, so the beam is
            // kept.)
            const int n_rel = __builtin_popcountll(rel);
            if (ANY && 2 * n_rel <= n_beam) {
                BeamLane tmp;
                make_beam<M>(sv, o, d, ((alive >> lane) & 1ull) != 0ull, b, tmp, false);
                n_beam = n_rel;
                if (STATS && lane == 0) st.walks[2] += 1;
            }
            const uint32_t sel = kmin & 63u;
            expand = (uint32_t)__builtin_amdgcn_readlane((int)cref, (int)sel);
            if (lane == sel) key = 0xFFFFFFFFu;
        }
    }
    if (ANY && active && !((alive >> lane) & 1ull)) occluded = true;
    return true;
}

// Picks the dominant axis and runs the beam walk; false when the wave needs the per-ray walk.
template <bool ANY, bool STATS>
__device__ __forceinline__ bool beam_try(const SceneView& sv, uint32_t root_ref, uint32_t obj_index, rt_f3 o, rt_f3 d,
                                         bool active, float& closest, HitD& hit, bool& occluded, uint2* lds,
                                         LaneStats& st) {
    const unsigned long long act = __ballot(active);
    if (act == 0ull) return true;
    float sgn;
    const int m = beam_axis(d, act, &sgn);
    if (m == 0) return beam_search<ANY, STATS, 0>(sv, root_ref, obj_index, o, d, active, sgn, closest, hit, occluded, lds, st);
    if (m == 1) return beam_search<ANY, STATS, 1>(sv, root_ref, obj_index, o, d, active, sgn, closest, hit, occluded, lds, st);
    return beam_search<ANY, STATS, 2>(sv, root_ref, obj_index, o, d, active, sgn, closest, hit, occluded, lds, st);
}

// The wave-coherent forms of closest_all / any_all. Every lane of the wave must call them
// together; `active` says whether the lane carries a ray.
// DEFER: a lane whose ray needs closest_resolve comes back with tri = RT_REDO instead of calling it (a
// call inside a kernel that is held to 64 registers costs the kernel its register allocation).
template <bool STATS, bool DEFER = false>
__device__ __forceinline__ HitD closest_all_packet(const SceneView& sv, rt_f3 o, rt_f3 d, bool active, uint2* beam_lds,
                                                   LaneStats& st) {
    bool redo = false;
    const rt_f3 inv = g_recip(d);
    HitD h;
    h.tri = RT_MISS;
    h.obj = 0;
    h.t = h.u = h.v = 0.0f;
    float closest = RT_INF;
    if (STATS && active) st.rays += 1;
    bool unused = false;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        const char* op = reinterpret_cast<const char*>(sv.objs) + 64 * (size_t)oi;
        const f32x16_t ob = sload16(op);
        const rt_f3 bbmin = f3(ob[0], ob[1], ob[2]), bbmax = f3(ob[3], ob[4], ob[5]);
        const uint32_t tri_base = __float_as_uint(ob[6]), n_tris = __float_as_uint(ob[7]);
        const uint32_t wide_root = __float_as_uint(ob[8]), root = __float_as_uint(ob[9]);
        const uint32_t rest_base = __float_as_uint(ob[10]), n_rest = __float_as_uint(ob[11]);
        if (STATS && active) st.box += 1;
        const bool in_obj = active && box_intersection_check(o, inv, bbmin, bbmax);
        if (RT_BALLOT(in_obj) == 0ull) continue;
        if (root != EMPTY) {
            const float before = closest;
            if (!beam_try<false, STATS>(sv, wide_root, oi, o, d, in_obj, closest, h, unused, beam_lds, st))
                packet_search<false, STATS>(sv, root, oi, o, d, inv, in_obj, closest, h, unused, st);
            if (closest < before) {
                if (DEFER) redo = redo || closest_needs_resolve(sv, oi, tri_base, __float_as_uint(ob[15]), o, d, inv, h);
                else closest_confirm(sv, oi, o, d, inv, before, h, closest);
            }
        }
        for (uint32_t r = 0; r < n_rest; ++r) {
            const char* pv = reinterpret_cast<const char*>(sv.rest_verts) + 36 * (size_t)(rest_base + r);
            const f32x8_t va = sload8(pv);
            const float vc2 = __uint_as_float(sload1u(pv + 32));
            float t, u, v;
            if (STATS && in_obj) st.tri += 1;
            if (in_obj && triangle_ray_hits(f3(va[0], va[1], va[2]), f3(va[3], va[4], va[5]), f3(va[6], va[7], vc2), o,
                                            d, closest, &t, &u, &v)) {
                if (closest > t) {
                    h.tri = tri_base + n_tris - n_rest + r;
                    h.obj = oi;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                }
            }
        }
    }
    h.closest = closest;
    if (DEFER && redo) h.tri = RT_REDO;
    return h;
}

// Any-hit search of one object's tree for the lanes `in_obj` (one ray per lane): first the whole
// wave as one beam; failing that, one cell's lanes at a time, each as a beam if it forms one and by
// the mask walk otherwise. Returns the lane's "occluded".
template <bool STATS>
__device__ __forceinline__ bool any_object_tree(const SceneView& sv, uint32_t wide_root, uint32_t root, uint32_t oi, rt_f3 o,
                                                rt_f3 d, rt_f3 inv, bool in_obj, uint32_t cell, uint2* beam_lds,
                                                LaneStats& st) {
    float closest = RT_INF;
    HitD dummy;
    bool occ = false;
    bool todo = in_obj, whole = true;
    for (;;) {
        const unsigned long long tm = __ballot(todo);
        if (tm == 0ull) break;
        bool grp = todo;
        if (!whole) {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cell, __builtin_ctzll(tm));
            grp = todo && cell == c0;
        }
        // A group equal to the whole wave has already failed as a beam.
        const bool same = !whole && __ballot(grp) == __ballot(in_obj);
        bool done = !same && beam_try<true, STATS>(sv, wide_root, oi, o, d, grp, closest, dummy, occ, beam_lds, st);
        if (!done && !whole) {
            packet_search<true, STATS>(sv, root, oi, o, d, inv, grp, closest, dummy, occ, st);
            done = true;
        }
        if (done) todo = todo && !grp;
        whole = false;
    }
    return occ;
}

// `cell` groups the lanes: when the whole wave does not form a usable beam, the lanes of one
// cell at a time are tried as a beam of their own (rays sorted by a spatial key share a cell with
// their neighbours; a wave that straddles two distant cells splits into two thin beams).
template <bool STATS>
__device__ __forceinline__ bool any_all_packet(const SceneView& sv, rt_f3 o, rt_f3 d, bool active, uint2* beam_lds,
                                               LaneStats& st, uint32_t cell = 0) {
    const rt_f3 inv = g_recip(d);
    if (STATS && active) st.rays += 1;
    bool occluded = false;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        if (RT_BALLOT(active) == 0ull) break;
        const char* op = reinterpret_cast<const char*>(sv.objs) + 64 * (size_t)oi;
        const f32x16_t ob = sload16(op);
        const rt_f3 bbmin = f3(ob[0], ob[1], ob[2]), bbmax = f3(ob[3], ob[4], ob[5]);
        const uint32_t wide_root = __float_as_uint(ob[8]), root = __float_as_uint(ob[9]);
        const uint32_t rest_base = __float_as_uint(ob[10]), n_rest = __float_as_uint(ob[11]);
        if (STATS && active) st.box += 1;
        const bool in_obj = active && box_intersection_check(o, inv, bbmin, bbmax);
        if (RT_BALLOT(in_obj) == 0ull) continue;
        if (root != EMPTY) {
            if (any_object_tree<STATS>(sv, wide_root, root, oi, o, d, inv, in_obj, cell, beam_lds, st)) {
                occluded = true;
                active = false;
            }
        }
        for (uint32_t r = 0; r < n_rest; ++r) {
            const char* pv = reinterpret_cast<const char*>(sv.rest_verts) + 36 * (size_t)(rest_base + r);
            const f32x8_t va = sload8(pv);
            const float vc2 = __uint_as_float(sload1u(pv + 32));
            float t, u, v;
            const bool test = in_obj && active;
            if (STATS && test) st.tri += 1;
            if (test && triangle_ray_hits(f3(va[0], va[1], va[2]), f3(va[3], va[4], va[5]), f3(va[6], va[7], vc2), o, d,
                                          RT_INF, &t, &u, &v)) {
                occluded = true;
                active = false;
            }
        }
    }
    return occluded;
}

__device__ __forceinline__ void flush_stats(const LaneStats& st, DevStats* out) {
    // One atomic per counter per lane would be heavy; reduce over the wave first.
    unsigned long long v[9] = {st.rays, st.box, st.tri, st.node, st.bytes, st.walks[0], st.walks[1], st.walks[2], st.walks[3]};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[k] = x;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out->rays, v[0]);
        atomicAdd(&out->box_tests, v[1]);
        atomicAdd(&out->tri_tests, v[2]);
        atomicAdd(&out->node_visits, v[3]);
        atomicAdd(&out->fetch_bytes, v[4]);
        for (int k = 0; k < 4; ++k) atomicAdd(&out->walks[k], v[5 + k]);
    }
}

// ---------------------------------------------------------------------------------------------
// Batch kernels
// ---------------------------------------------------------------------------------------------
#ifndef RT_LANE_WAVES
#define RT_LANE_WAVES 8
#endif
__global__ __launch_bounds__(RT_BLOCK, RT_LANE_WAVES) void k_trace_closest(SceneView sv, const rt_ray* __restrict__ rays,
                                                          uint64_t n, rt_hit* __restrict__ out) {
    __shared__ uint2 s_stack[RT_STACK_LDS * RT_BLOCK];
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint2 stack_rest[LaneStack::REST];
    LaneStack stk;
    stk.lds = &s_stack[threadIdx.x];
    stk.spill = stack_rest;
    LaneStats st;
    const rt_ray r = rays[i];
    const HitD h = closest_all<false>(sv, f3(r.o[0], r.o[1], r.o[2]), f3(r.d[0], r.d[1], r.d[2]), stk, st);
    rt_hit o;
    o.tri = h.tri;
    o.t = h.t;
    o.u = h.u;
    o.v = h.v;
    out[i] = o;
}

// ---------------------------------------------------------------------------------------------
// Persistent per-lane walk for rays that share nothing: a wave keeps running and hands a new ray
// to a lane as soon as its ray is finished. The lengths of incoherent rays differ widely (in a
// triangle soup the longest of 64 runs about five times the mean), so a wave that takes 64 rays and
// waits for the last one leaves most lanes idle most of the time.
// The walk of closest_all is kept as a per-lane state machine: `walking` lanes take one node or one
// leaf per turn; the others move to the next object whose box their ray meets (after the trailing
// triangles of the one they leave), or deliver their result and become idle. A wave takes ray
// indices RT_CHUNK at a time and hands them to its idle lanes RT_REFILL or more at a time. The
// chunks of the first RT_STATIC_PCT per cent of the rays are the wave's own by number (wave w owns
// chunks w, w + waves, ...), the rest are handed out through a global counter so that the waves end
// together. The waves of a launch do not run at one speed: a SIMD issues from its oldest waves first,
// and with eight waves on it the youngest gets about half the instructions of the oldest (measured
// with the clock inside the kernel: equal shares of a launch ended between 98 and 127 ms), so the
// shared part has to be large. It cannot be everything: an atomic on one address shared by the whole
// device takes about 12 ns, which would bound a launch of cheap rays: the diffuse-bounce frame
// measured 148 / 134 / 137 / 149 ms with 0 / 35 / 50 / 75 per cent given out by number (30 to 50 per
// cent are within 1 % of each other).
// The loop ends when the wave has had
// its own chunks, the counter has passed the last ray and no lane holds one. Every wave overshoots
// the counter by at most one chunk, so n_rays must stay RT_WALK_HEADROOM below 2^32.
// Source: bool init(j, o, d) gives ray j (false: there is none, and the source has stored whatever
// stands for it), void emit(j, hit, o, d) stores its result. A source with QUEUES set collects
// follow-up rays while it emits (per wave, in LDS) and is given the chance to write them out with
// the whole wave: flush(lane, false) after every turn in which a lane emitted, flush(lane, true) at
// the end. In shadow mode (ANY) the walk ends at
// the first occluder and hit.tri != RT_MISS says "occluded".
// ---------------------------------------------------------------------------------------------
#ifndef RT_WALK_LDS
#define RT_WALK_LDS 13
#endif
#ifndef RT_REFILL
#define RT_REFILL 16
#endif
#ifndef RT_WALK_WAVES
#define RT_WALK_WAVES 8
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 64
#endif
#ifndef RT_WALK_REPS
#define RT_WALK_REPS 4
#endif
#ifndef RT_VOTE_T
#define RT_VOTE_T 16
#endif
#ifndef RT_STATIC_PCT
#define RT_STATIC_PCT 50
#endif
// The persistent walk reads the compact nodes (node_step_q).
#define RT_WALK_MAX_GRID (1024u * RT_WALK_WAVES)                              // waves of a persistent launch
#define RT_WALK_HEADROOM (RT_WALK_MAX_GRID * RT_CHUNK + 64u)  // counter overshoot at the end of a launch
// Stack of the persistent walk: RT_WALK_LDS entries of 6 bytes in LDS (node reference, and the
// entry distance cut to its upper 16 bits: rounded towards zero, so an entry is never dropped that
// the full distance would keep), the rest in private memory. LDS is handed out in pieces of 1,280
// bytes and 32 waves a CU can have four each: 13 entries per lane are 4,992 bytes a wave, and nothing else of
// a walk kernel's may live in LDS. With 8-byte entries the same LDS held 8; every entry more that stays out
// of private memory is worth 2 to 3 % on incoherent rays.
struct WalkLds {
    uint32_t refs[RT_WALK_LDS * 64];
    uint16_t keys[RT_WALK_LDS * 64];
};
static_assert(sizeof(WalkLds) <= 4 * 1280, "a walk's LDS has to leave room for 32 waves a CU");
struct WalkStack {
    WalkLds* lds;    // the wave's block (wave-uniform); entry e of this lane at [e * 64 + threadIdx.x]
    uint2* spill;    // RT_STACK_MAX - RT_WALK_LDS entries of the caller's, in private memory. (An array inside
                     // this struct would drag the stack pointer into private memory with it: every push and
                     // pop then goes through scratch loads.)
    // The stack pointer as the byte offset of the next entry's reference in the wave's block: entries * 256
    // + 4 * lane. The reference's address is then the pointer itself and the key's half of it, where an
    // entry count costs a shift, an or and two more shifts per push and per pop.
    uint32_t spb;
    __device__ __forceinline__ void reset() { spb = 4u * threadIdx.x; }
    __device__ __forceinline__ bool empty() const { return spb < 256u; }
    __device__ __forceinline__ void push(uint32_t node, float tnear) {
        char* base = reinterpret_cast<char*>(lds);
        if (spb < RT_WALK_LDS * 256u) {
            *reinterpret_cast<uint32_t*>(base + spb) = node;
            *reinterpret_cast<uint16_t*>(base + sizeof(lds->refs) + (spb >> 1)) = (uint16_t)(__float_as_uint(tnear) >> 16);
        } else if (spb < RT_STACK_MAX * 256u) {
            spill[(spb >> 8) - RT_WALK_LDS] = make_uint2(node, __float_as_uint(tnear));
        }
        spb = spb < RT_STACK_MAX * 256u ? spb + 256u : spb;
    }
    __device__ __forceinline__ uint2 pop() {
        spb -= 256u;
        char* base = reinterpret_cast<char*>(lds);
        if (spb < RT_WALK_LDS * 256u)
            return make_uint2(*reinterpret_cast<const uint32_t*>(base + spb),
                              (uint32_t)*reinterpret_cast<const uint16_t*>(base + sizeof(lds->refs) + (spb >> 1)) << 16);
        return spill[(spb >> 8) - RT_WALK_LDS];
    }
};

template <bool ANY, bool STATS, class Source>
__device__ __forceinline__ void persistent_walk(const SceneView& sv, uint32_t n_rays, uint32_t* counter,
                                                const Source& src, WalkLds& s_stack, LaneStats& st) {
    const uint32_t lane = threadIdx.x & 63u;
    uint2 stack_rest[RT_STACK_MAX - RT_WALK_LDS];
    WalkStack stk;
    stk.spill = stack_rest;
    stk.lds = &s_stack;
    stk.reset();
    bool has = false, walking = false, in_obj = false;
    bool exhausted = false;  // wave-uniform: the counter has passed the last ray
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: ray indices taken and not yet handed out
    // Whole rounds of one chunk per wave inside the first RT_STATIC_PCT per cent of the rays.
    const uint32_t static_turns = (uint32_t)((uint64_t)n_rays * RT_STATIC_PCT / 100u) / (gridDim.x * (uint32_t)RT_CHUNK);
    uint32_t turn = 0;
    const uint32_t shared_from = static_turns * gridDim.x * (uint32_t)RT_CHUNK;
    const uint32_t shards = sv.walk_counter_shards == RT_COUNTER_SHARDS ? RT_COUNTER_SHARDS : 1u;   // (never zero)
    // rays per stretch of the shared part: whole chunks
    const uint32_t shard_rays = ((n_rays - shared_from + shards * (uint32_t)RT_CHUNK - 1u) / (shards * (uint32_t)RT_CHUNK)) * (uint32_t)RT_CHUNK;
    uint32_t shard = blockIdx.x % shards, shards_left = shards;
    uint32_t j = 0, oi = 0, cur = 0, local_best = 0;
    rt_f3 o = f3(0.0f, 0.0f, 0.0f), d = f3(0.0f, 0.0f, 1.0f), inv = d;
    float closest = RT_INF;
    HitD hit;
    hit.tri = RT_MISS;
    hit.obj = 0;
    hit.t = hit.u = hit.v = 0.0f;
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        if (!exhausted && n_idle >= RT_REFILL) {
            if (chunk_next == chunk_end) {
                // The wave's own chunks first (chunk numbers blockIdx.x, + gridDim.x, ...: no
                // traffic), then chunks of the shared remainder, which evens out the end.
                uint32_t base;
                if (turn < static_turns) {
                    base = (turn * gridDim.x + blockIdx.x) * (uint32_t)RT_CHUNK;
                    ++turn;
                } else {
                    // The shared rays are cut into `shards` stretches (1 or RT_COUNTER_SHARDS, the scene's
                    // choice: see SceneView), each with a counter in a cache line of its own; a wave starts
                    // at the stretch of its number and moves on to the next one when a stretch is used up
                    // (a stretch that is used up stays so).
                    const int first = __builtin_ctzll(idle);
                    base = n_rays;
                    while (shards_left != 0u) {
                        uint32_t got = 0;
                        if ((int)lane == first) got = atomicAdd(counter + 16u * shard, (uint32_t)RT_CHUNK);
                        got = (uint32_t)__builtin_amdgcn_readlane((int)got, first);
                        const uint32_t lo = shared_from + shard * shard_rays;
                        if (got < shard_rays && lo + got < n_rays) {
                            base = lo + got;
                            break;
                        }
                        shard = shard + 1u == shards ? 0u : shard + 1u;
                        --shards_left;
                    }
                }
                exhausted = base >= n_rays;
                chunk_next = exhausted ? 0u : base;
                chunk_end = exhausted ? 0u : (n_rays - base < (uint32_t)RT_CHUNK ? n_rays : base + (uint32_t)RT_CHUNK);
            }
            const uint32_t left = chunk_end - chunk_next, take = n_idle < left ? n_idle : left;
            const uint32_t rank = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
            const uint32_t first_idx = chunk_next;
            chunk_next += take;
            if (!has && rank < take) {
                j = first_idx + rank;
                if (src.init(j, o, d)) {
                    has = true;
                    walking = false;
                    in_obj = false;
                    oi = 0;
                    inv = g_recip(d);
                    closest = RT_INF;
                    hit.tri = RT_MISS;
                    hit.obj = 0;
                    hit.t = hit.u = hit.v = 0.0f;
                    if (STATS) st.rays += 1;
                }
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted) break;
            continue;
        }
        bool emitted = false;
        // The object step runs once RT_VOTE_T lanes wait for it (it is what frees lanes for new rays) or
        // nothing else can; in every other turn all walking lanes take their step, node or triangle, on
        // one set of loads (see below).
        const int n_t = __builtin_popcountll(__ballot(has && !walking));
        const bool run_t = n_t >= RT_VOTE_T || __ballot(has && walking) == 0ull;
        if (run_t && has && !walking) {
            // Leave the current object (its trailing n % 8 triangles, src/main.rs:157-169: scalar
            // arithmetic, `closest` is not lowered, the last qualifying one wins), then find the
            // next object whose box the ray meets.
            if (in_obj) {
                const DevObject& ob = sv.objs[oi];
                // The object's tree is done. A hit of this object's (its trailing triangles come below, so
                // hit.obj == oi says so) whose answer may depend on the reference's order ends the ray here
                // with RT_REDO for a result: the kernel that follows the walk decides it.
                bool redo = false;
                if (!ANY && hit.tri != RT_MISS && hit.obj == oi) redo = closest_needs_resolve(sv, oi, ob.tri_base, ob.leaf_base, o, d, inv, hit);
                if (redo) hit.tri = RT_REDO, oi = sv.n_objs;
                for (uint32_t r = 0; !redo && r < ob.n_rest; ++r) {
                    const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
                    float t, u, v;
                    if (STATS) st.tri += 1;
                    if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, closest, &t, &u, &v)) {
                        if (ANY) {
                            // src/main.rs:226-231: any trailing triangle that qualifies occludes.
                            hit.tri = 0;
                            oi = sv.n_objs;
                            break;
                        }
                        if (closest > t) {
                            hit.tri = ob.tri_base + ob.n_tris - ob.n_rest + r;
                            hit.obj = oi;
                            hit.t = t;
                            hit.u = u;
                            hit.v = v;
                        }
                    }
                }
                in_obj = false;
                if (oi < sv.n_objs) ++oi;
            }
            while (oi < sv.n_objs) {
                const DevObject& ob = sv.objs[oi];
                if (STATS) st.box += 1;
                if (box_intersection_check(o, inv, ld3(ob.bbmin), ld3(ob.bbmax))) break;
                ++oi;
            }
            if (oi < sv.n_objs) {
                in_obj = true;
                const uint32_t root = sv.objs[oi].root;
                if (root != EMPTY) {
                    walking = true;
                    cur = root;
                    local_best = 0;
                    hit.near = 0xFFFFFFFFu;
                    stk.reset();
                }
            } else {
                hit.closest = closest;
                src.emit(j, hit, o, d);
                has = false;
                emitted = true;
            }
        }
        if constexpr (Source::QUEUES) {
            if (__ballot(emitted) != 0ull) src.flush(lane, false);
        }
        // RT_WALK_REPS such steps per turn of the refill and object logic above: the walk is bound by
        // instruction issue, and that logic costs every lane its ballots and branches whether it has
        // anything to do there or not.
#pragma unroll
        for (int rep = 0; rep < RT_WALK_REPS; ++rep)
        if (has && walking && !run_t) {
            // A lane at a node reads the node's 64 bytes, a lane at a leaf the 48 bytes of the leaf's next
            // triangle: the same three or four loads from an address of the lane's own, issued together,
            // so a turn waits for memory once whatever mix of lanes it holds (a wave that alternates node
            // turns and leaf turns waits twice, with the lanes of the other kind idle each time). A leaf
            // reference counts down in place: one record further, one triangle fewer.
            const bool at_leaf = (cur & LEAF_BIT) != 0u;
            const float4* ptr = at_leaf ? sv.recs + 3 * (size_t)((cur & 0x7FFFFFFFu) >> 3) : sv.qnodes + 4 * (size_t)cur;
            // (A leaf lane's fourth load reads into the next record; the array is padded for it. Asked for by
            // all lanes and pinned here, the four loads go out back to back before either kind computes.)
            float4 q0 = ptr[0], q1 = ptr[1], q2 = ptr[2], q3 = ptr[3];
            asm volatile("" :: "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w),
                              "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w), "v"(q3.x), "v"(q3.y), "v"(q3.z), "v"(q3.w));
            bool need_pop = true;
            if (!at_leaf) {
                uint32_t next;
                const float nox = -(o.x * inv.x), noy = -(o.y * inv.y), noz = -(o.z * inv.z);
                if (node_step_q<STATS>(q0, q1, q2, q3, inv, nox, noy, noz, closest, stk, st, &next)) {
                    cur = next;
                    need_pop = false;
                }
            } else {
                if (STATS) st.tri += 1, st.bytes += 48;
                if (record_test<ANY>(q0, q1, q2, oi, o, d, closest, local_best, hit)) {
                    hit.tri = 0;
                    stk.reset();
                    in_obj = false;
                    oi = sv.n_objs;
                } else if ((cur & 7u) != 0u) {
                    cur += 7u;
                    need_pop = false;
                }
            }
            if (need_pop) {
                walking = false;
                while (!stk.empty()) {
                    const uint2 e = stk.pop();
                    cur = e.x;   // (taken before the entry is looked at: the reference is then read beside its key, not after it)
                    if (!(__uint_as_float(e.y) > closest)) {
                        walking = true;
                        break;
                    }
                }
            }
        }
    }
    if constexpr (Source::QUEUES) src.flush(lane, true);
}

// Rays of a batch call.
struct BatchSource {
    static constexpr bool QUEUES = false;
    const rt_ray* rays;
    rt_hit* out;
    __device__ __forceinline__ bool init(uint32_t j, rt_f3& o, rt_f3& d) const {
        const rt_ray r = rays[j];
        o = f3(r.o[0], r.o[1], r.o[2]);
        d = f3(r.d[0], r.d[1], r.d[2]);
        return true;
    }
    __device__ __forceinline__ void emit(uint32_t j, const HitD& h, rt_f3, rt_f3) const {
        rt_hit r;
        r.tri = h.tri;   // (RT_REDO as it is: redo_marked finds it there)
        r.t = h.t;
        r.u = h.u;
        r.v = h.v;
        out[j] = r;
    }
    __device__ __forceinline__ bool marked(uint32_t j) const { return out[j].tri == RT_REDO; }
    __device__ __forceinline__ void flush(uint32_t, bool) const {}
};

// The rays a walk has left undecided (RT_REDO): the per-lane search with closest_resolve for each, through
// the same source. Whole waves call this; n is wave-uniform.
template <class Source>
__device__ __forceinline__ void redo_marked(const SceneView& sv, uint32_t n, const Source& src) {
    using Stack = LaneStackT<RT_STACK_LDS, 64>;
    __shared__ uint2 s_stack[RT_STACK_LDS * 64];
    uint2 stack_rest[Stack::REST];
    Stack stk;
    stk.lds = &s_stack[threadIdx.x];
    stk.spill = stack_rest;
    LaneStats st;
    const uint32_t lane = threadIdx.x;
    for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {
        const uint32_t idx = base + lane;
        const bool m = idx < n && src.marked(idx);
        if (__ballot(m) == 0ull) continue;
        if (m) {
            rt_f3 o, d;
            if (src.init(idx, o, d)) {
                const HitD h = closest_all<false>(sv, o, d, stk, st);
                src.emit(idx, h, o, d);
            }
        }
        if constexpr (Source::QUEUES) src.flush(lane, false);
    }
    if constexpr (Source::QUEUES) src.flush(lane, true);
}
__global__ __launch_bounds__(64) void k_trace_closest_redo(SceneView sv, const rt_ray* __restrict__ rays, uint32_t n,
                                                           rt_hit* __restrict__ out) {
    const BatchSource src{rays, out};
    redo_marked(sv, n, src);
}

struct BatchAnySource {
    static constexpr bool QUEUES = false;
    const rt_ray* rays;
    uint8_t* out;
    __device__ __forceinline__ bool init(uint32_t j, rt_f3& o, rt_f3& d) const {
        const rt_ray r = rays[j];
        o = f3(r.o[0], r.o[1], r.o[2]);
        d = f3(r.d[0], r.d[1], r.d[2]);
        return true;
    }
    __device__ __forceinline__ void emit(uint32_t j, const HitD& h, rt_f3, rt_f3) const {
        out[j] = h.tri != RT_MISS ? 1 : 0;
    }
};

__global__ __launch_bounds__(64, RT_WALK_WAVES) void k_trace_any_persistent(SceneView sv, const rt_ray* __restrict__ rays,
                                                                           uint32_t n, uint32_t* __restrict__ counter,
                                                                           uint8_t* __restrict__ out) {
    __shared__ WalkLds s_stack;
    LaneStats st;
    const BatchAnySource src{rays, out};
    persistent_walk<true, false>(sv, n, counter, src, s_stack, st);
}

__global__ __launch_bounds__(64, RT_WALK_WAVES) void k_trace_closest_persistent(SceneView sv, const rt_ray* __restrict__ rays,
                                                                   uint32_t n, uint32_t* __restrict__ counter,
                                                                   rt_hit* __restrict__ out) {
    __shared__ WalkLds s_stack;
    LaneStats st;
    const BatchSource src{rays, out};
    persistent_walk<false, false>(sv, n, counter, src, s_stack, st);
}

// The same search with the wave-coherent walks (beam where the wave's rays form one, mask walk
// otherwise): for batches whose consecutive rays are neighbours.
__global__ __launch_bounds__(64, 8) void k_trace_closest_wave(SceneView sv, const rt_ray* __restrict__ rays, uint64_t n,
                                                             uint32_t beam_levels, rt_hit* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint2* s_beam = reinterpret_cast<uint2*>(s_dyn);
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const bool act = i < n;
    LaneStats st;
    rt_ray r;
    if (act) r = rays[i];
    else r = rt_ray{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
    const HitD h = closest_all_packet<false, true>(sv, f3(r.o[0], r.o[1], r.o[2]), f3(r.d[0], r.d[1], r.d[2]), act, s_beam, st);
    if (act) {
        rt_hit o;
        o.tri = h.tri;
        o.t = h.t;
        o.u = h.u;
        o.v = h.v;
        out[i] = o;
    }
}

__global__ __launch_bounds__(RT_BLOCK, RT_LANE_WAVES) void k_trace_any(SceneView sv, const rt_ray* __restrict__ rays,
                                                      uint64_t n, uint8_t* __restrict__ occluded) {
    __shared__ uint2 s_stack[RT_STACK_LDS * RT_BLOCK];
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint2 stack_rest[LaneStack::REST];
    LaneStack stk;
    stk.lds = &s_stack[threadIdx.x];
    stk.spill = stack_rest;
    LaneStats st;
    const rt_ray r = rays[i];
    occluded[i] = any_all<false>(sv, f3(r.o[0], r.o[1], r.o[2]), f3(r.d[0], r.d[1], r.d[2]), stk, st) ? 1 : 0;
}

// The reference's own visiting order, src/main.rs:100-171: object box, then every 8-wide packet
// of leaf boxes with the `closest` of the moment, then the leaves that passed, then the leftover
// leaves and the trailing triangles. All lanes read the same boxes and triangles.
__global__ __launch_bounds__(RT_BLOCK) void k_trace_closest_linear(SceneView sv, const rt_ray* __restrict__ rays,
                                                                 uint64_t n, rt_hit* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const rt_ray rr = rays[i];
    const rt_f3 o = f3(rr.o[0], rr.o[1], rr.o[2]), d = f3(rr.d[0], rr.d[1], rr.d[2]);
    const rt_f3 inv = g_recip(d);
    rt_hit h;
    h.tri = RT_MISS;
    h.t = h.u = h.v = 0.0f;
    float closest = RT_INF;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        const DevObject& ob = sv.objs[oi];
        if (!box_intersection_check(o, inv, ld3(ob.bbmin), ld3(ob.bbmax))) continue;
        const uint32_t n_leaves = ob.n_tris / 8;
        const uint32_t n_packets = n_leaves / 8;
        auto scan_leaf = [&](uint32_t leaf) {
            // PackedTriangles::intersect with ray_length = closest at entry, then the j loop.
            const float ray_length = closest;
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t ti = leaf * 8 + j;
                const float* p = sv.tri9 + 9 * (size_t)(ob.tri_base + ti);
                const rt_tri_eval ev = tri_intersect(ld3(p), ld3(p + 3), ld3(p + 6), o, d);
                const bool failed = ev.failed || (ev.t > ray_length);
                if (!failed && ev.t > 0.0f && closest > ev.t) {
                    closest = ev.t;
                    h.tri = ob.tri_base + ti;
                    h.t = ev.t;
                    h.u = ev.u;
                    h.v = ev.v;
                }
            }
        };
        for (uint32_t pk = 0; pk < n_packets; ++pk) {
            const float ray_length = closest;  // closest_splat when the packet is tested
            uint32_t mask = 0;
            for (uint32_t b = 0; b < 8; ++b) {
                const float* lb = sv.leaf_bounds + 6 * (size_t)(ob.leaf_base + pk * 8 + b);
                if (packed_box_hit(ld3(lb), ld3(lb + 3), o, inv, ray_length)) mask |= 1u << b;
            }
            for (uint32_t b = 0; b < 8; ++b)
                if (mask & (1u << b)) scan_leaf(pk * 8 + b);
        }
        for (uint32_t l = n_packets * 8; l < n_leaves; ++l) {
            const float* lb = sv.leaf_bounds + 6 * (size_t)(ob.leaf_base + l);
            if (box_intersection_check(o, inv, ld3(lb), ld3(lb + 3))) scan_leaf(l);
        }
        for (uint32_t r = 0; r < ob.n_rest; ++r) {
            const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
            float t, u, v;
            if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, closest, &t, &u, &v)) {
                if (closest > t) {
                    h.tri = ob.tri_base + ob.n_tris - ob.n_rest + r;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                }
            }
        }
    }
    out[i] = h;
}

// ---------------------------------------------------------------------------------------------
// Scene preparation
// ---------------------------------------------------------------------------------------------
// pack_triangles, src/simd_accel.rs:8-66: e1 = b - a, e2 = c - a, v0 = a, one subtraction each.
__global__ void k_pack_tris(const float* __restrict__ verts, const uint32_t* __restrict__ tris, uint32_t n,
                            float* __restrict__ tri9) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rt_f3 a = ld3(verts + 3 * (size_t)tris[3 * (size_t)i]);
    const rt_f3 b = ld3(verts + 3 * (size_t)tris[3 * (size_t)i + 1]);
    const rt_f3 c = ld3(verts + 3 * (size_t)tris[3 * (size_t)i + 2]);
    const rt_f3 e1 = b - a, e2 = c - a;
    float* o = tri9 + 9 * (size_t)i;
    o[0] = e1.x; o[1] = e1.y; o[2] = e1.z;
    o[3] = e2.x; o[4] = e2.y; o[5] = e2.z;
    o[6] = a.x;  o[7] = a.y;  o[8] = a.z;
}

// Leaf boxes of pack_model, src/opt.rs:46-60: min / max over the 24 vertices of 8 triangles.
// leaf_first_tri[l] is the (global, sorted) index of the leaf's first triangle.
__global__ void k_leaf_bounds(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                              const uint32_t* __restrict__ leaf_first_tri, uint32_t n_leaves,
                              float* __restrict__ leaf_bounds, float4* __restrict__ leaf_gate) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_leaves) return;
    rt_f3 mn = f3(RT_INF, RT_INF, RT_INF), mx = f3(-RT_INF, -RT_INF, -RT_INF);
    // bit 31: one of its object's last n_leaves % 8 leaves, which the reference keeps out of its packets of eight
    const uint32_t t0 = leaf_first_tri[l] & 0x7FFFFFFFu;
    const bool leftover = (leaf_first_tri[l] >> 31) != 0u;
    for (uint32_t j = 0; j < 8; ++j) {
        const uint32_t* t = tris + 3 * (size_t)(t0 + j);
        const rt_f3 a = ld3(verts + 3 * (size_t)t[0]), b = ld3(verts + 3 * (size_t)t[1]),
                    c = ld3(verts + 3 * (size_t)t[2]);
        rt_f3 tmn = f3(g_min(g_min(a.x, b.x), c.x), g_min(g_min(a.y, b.y), c.y), g_min(g_min(a.z, b.z), c.z));
        rt_f3 tmx = f3(g_max(g_max(a.x, b.x), c.x), g_max(g_max(a.y, b.y), c.y), g_max(g_max(a.z, b.z), c.z));
        mn = f3(g_min(mn.x, tmn.x), g_min(mn.y, tmn.y), g_min(mn.z, tmn.z));
        mx = f3(g_max(mx.x, tmx.x), g_max(mx.y, tmx.y), g_max(mx.z, tmx.z));
    }
    float* o = leaf_bounds + 6 * (size_t)l;
    o[0] = mn.x; o[1] = mn.y; o[2] = mn.z;
    o[3] = mx.x; o[4] = mx.y; o[5] = mx.z;
    leaf_gate[2 * (size_t)l] = make_float4(mn.x, mn.y, mn.z, leftover ? 0.0f : 1.0f);
    leaf_gate[2 * (size_t)l + 1] = make_float4(mx.x, mx.y, mx.z, 0.0f);
}

// The compact form of the 4-wide nodes that node_step_q reads. One thread per node. The node's box
// (over its non-empty children) is widened by `pad` on every side, which covers what node_step_q
// rounds differently from node_step (the boxes themselves already carry the builders' margin for
// the slab products): the product scale * inv, the sum origin * inv - o * inv and the final fused
// multiply-add are each within 2^-24 of magnitudes no larger than |coordinate * inv|, and pad is 2^-17
// of the largest coordinate. Every child corner is then rounded outwards on the grid of 255 steps,
// and checked in the arithmetic the walk's conversion implies (origin + q * scale, one fma).
__global__ void k_quantize_nodes(const rtbvh::Node4* __restrict__ nodes, uint32_t n, float4* __restrict__ qnodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rtbvh::Node4 nd = nodes[i];
    float lo[3] = {RT_INF, RT_INF, RT_INF}, hi[3] = {-RT_INF, -RT_INF, -RT_INF};
    bool any = false;
    for (int k = 0; k < 4; ++k) {
        if (nd.c[k].ref == EMPTY) continue;
        bool finite = true;
        for (int a = 0; a < 3; ++a) finite = finite && __builtin_fabsf(nd.c[k].lo[a]) < RT_INF && __builtin_fabsf(nd.c[k].hi[a]) < RT_INF;
        if (!finite) continue;
        any = true;
        for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], nd.c[k].lo[a]), hi[a] = fmaxf(hi[a], nd.c[k].hi[a]);
    }
    uint32_t ref[4];
    unsigned char q[24];
    float origin[3] = {0.0f, 0.0f, 0.0f}, scale[3] = {0.0f, 0.0f, 0.0f};
    float pad[3] = {0.0f, 0.0f, 0.0f};
    if (any) {
        for (int a = 0; a < 3; ++a) {
            pad[a] = 7.62939453125e-6f * fmaxf(__builtin_fabsf(lo[a]), __builtin_fabsf(hi[a])) + 1e-20f;
            origin[a] = lo[a] - 2.0f * pad[a];
            scale[a] = ((hi[a] + 2.0f * pad[a]) - origin[a]) * (1.0f / 254.0f);   // 254 steps reach the far side: one to spare
        }
    }
    for (int k = 0; k < 4; ++k) {
        bool use = any && nd.c[k].ref != EMPTY;
        for (int a = 0; a < 3 && use; ++a) use = __builtin_fabsf(nd.c[k].lo[a]) < RT_INF && __builtin_fabsf(nd.c[k].hi[a]) < RT_INF;
        ref[k] = use ? nd.c[k].ref : EMPTY;
        for (int a = 0; a < 3; ++a) {
            int ql = 255, qh = 0;
            if (use) {
                const float l = nd.c[k].lo[a] - pad[a], h = nd.c[k].hi[a] + pad[a];
                ql = (int)floorf((l - origin[a]) / scale[a]);
                ql = min(max(ql, 0), 255);
                while (ql > 0 && __builtin_fmaf((float)ql, scale[a], origin[a]) > l) --ql;
                qh = (int)ceilf((h - origin[a]) / scale[a]);
                qh = min(max(qh, 0), 255);
                while (qh < 255 && __builtin_fmaf((float)qh, scale[a], origin[a]) < h) ++qh;
                // a corner the grid cannot reach (should not happen: 254 steps span the padded box)
                if (__builtin_fmaf((float)ql, scale[a], origin[a]) > l || __builtin_fmaf((float)qh, scale[a], origin[a]) < h) ql = 0, qh = 255;
            }
            q[8 * a + k] = (unsigned char)ql;       // word 2 a: the low planes of axis a, word 2 a + 1: the high ones
            q[8 * a + 4 + k] = (unsigned char)qh;
        }
    }
    uint32_t w[6];
    for (int j = 0; j < 6; ++j) w[j] = (uint32_t)q[4 * j] | ((uint32_t)q[4 * j + 1] << 8) | ((uint32_t)q[4 * j + 2] << 16) | ((uint32_t)q[4 * j + 3] << 24);
    float4* out = qnodes + 4 * (size_t)i;
    out[0] = make_float4(origin[0], origin[1], origin[2], scale[0]);
    out[1] = make_float4(scale[1], scale[2], __uint_as_float(ref[0]), __uint_as_float(ref[1]));
    out[2] = make_float4(__uint_as_float(ref[2]), __uint_as_float(ref[3]), __uint_as_float(w[0]), __uint_as_float(w[1]));
    out[3] = make_float4(__uint_as_float(w[2]), __uint_as_float(w[3]), __uint_as_float(w[4]), __uint_as_float(w[5]));
}

// Triangle records in tree-leaf order: record r holds triangle order[r] (a global id).
__global__ void k_gather_records(const float* __restrict__ tri9, const uint32_t* __restrict__ order, uint32_t n,
                                 float4* __restrict__ recs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t id = order[r];
    const float* p = tri9 + 9 * (size_t)id;
    recs[3 * (size_t)r + 0] = make_float4(p[6], p[7], p[8], p[0]);
    recs[3 * (size_t)r + 1] = make_float4(p[1], p[2], p[3], p[4]);
    recs[3 * (size_t)r + 2] = make_float4(p[5], __uint_as_float(id), 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// Frame kernel: draw + trace_ray, src/main.rs:88-379.
// ---------------------------------------------------------------------------------------------
struct Col {
    float r, g, b;
};
__device__ __forceinline__ Col cmul(Col c, float v) { return Col{c.r * v, c.g * v, c.b * v}; }
__device__ __forceinline__ Col cmul_col(Col c, Col d) { return Col{c.r * d.r, c.g * d.g, c.b * d.b}; }
__device__ __forceinline__ Col cadd(Col c, Col d) { return Col{c.r + d.r, c.g + d.g, c.b + d.b}; }
__device__ __forceinline__ Col cdiv(Col c, float v) { return Col{c.r / v, c.g / v, c.b / v}; }
// Color::from_u8(0x87, 0xce, 0xeb), src/main.rs:303.
// Color::gamma of the reference (powf(c, 1 / 2.2) per channel, src/objects.rs:151-156) as
// exp2(log2(c) / 2.2) on the hardware's exp2 / log2 (about 1e-6 relative, far inside the 1-LSB
// colour tolerance that libm differences already need): the full-precision powf was most of the
// resolve kernel's instructions. 0 stays 0, negative stays NaN.
#ifndef RT_FAST_GAMMA
#define RT_FAST_GAMMA 1
#endif
__device__ __forceinline__ float gamma22_1(float c) {
#if RT_FAST_GAMMA
    return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 2.2f));
#else
    return powf(c, 1.0f / 2.2f);
#endif
}
__device__ __forceinline__ Col sky() { return Col{135.0f / 255.0f, 206.0f / 255.0f, 235.0f / 255.0f}; }
__device__ __forceinline__ Col gamma22(Col c) { return Col{gamma22_1(c.r), gamma22_1(c.g), gamma22_1(c.b)}; }

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
// Counter-based stand-in for rand::random::<f32>() (src/main.rs:262,264): 24 random bits to
// [0,1), keyed on (seed, sample index, scatter index, which of the two draws).
__device__ __forceinline__ float counter_rand(uint32_t seed, uint64_t sample_index, uint32_t scatter, uint32_t which) {
    uint32_t h = mix32(seed ^ 0x9E3779B9u);
    h = mix32(h ^ (uint32_t)sample_index);
    h = mix32(h ^ (uint32_t)(sample_index >> 32));
    h = mix32(h ^ (scatter * 2u + which));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// ScreenColor::from, src/objects.rs:90-97: (c * 255).min(255) as u8 (saturating, NaN -> 255).
__device__ __forceinline__ uint32_t to_u8(float c) {
    float v = g_min(c * 255.0f, 255.0f);
    if (!(v > 0.0f)) return 0u;
    return (uint32_t)v;
}

struct Surface {
    rt_f3 pos;     // hit position offset along the stored normal, src/main.rs:121-124
    rt_f3 normal;
    Col color;
    float albedo, reflectivity, roughness;
};

__device__ __forceinline__ Surface surface_of(const SceneView& sv, const HitD& h, rt_f3 o, rt_f3 d) {
    const float* td = sv.tri_data + 6 * (size_t)h.tri;
    Surface s;
    s.normal = ld3(td);
    s.color = Col{td[3], td[4], td[5]};
    s.pos = (o + d * h.t) + (s.normal * 0.00001f);
    const DevObject& ob = sv.objs[h.obj];
    s.albedo = ob.albedo;
    s.reflectivity = ob.reflectivity;
    s.roughness = ob.roughness;
    return s;
}

// Direct sun light at a surface, src/main.rs:176-241.
template <bool STATS>
__device__ __forceinline__ Col direct_light(const SceneView& sv, const Surface& s, rt_f3 sun, LaneStack& stk,
                                            LaneStats& st) {
    const rt_f3 sun_dir = g_normalize(sun - s.pos);
    Col c{0.0f, 0.0f, 0.0f};
    if (g_dot(s.normal, sun_dir) > 0.0f) {
        if (!any_all<STATS>(sv, s.pos, sun_dir, stk, st)) {
            const float light_power = g_dot(s.normal, sun_dir);
            c = cmul(s.color, light_power * s.albedo);
        }
    }
    return c;
}

// trace_ray for a ray at `depth` when no diffuse scattering happens at this depth or below
// (depth >= DIFFUSE_DEPTH): the recursion is a chain of mirror bounces, src/main.rs:243-257,
// folded back in the reference's order: color = direct + col * reflectivity.
template <bool STATS>
__device__ __forceinline__ Col mirror_chain(const SceneView& sv, const DevRenderParams& p, rt_f3 o, rt_f3 d,
                                            uint32_t depth, LaneStack& stk, LaneStats& st) {
    Col direct[RT_MAX_REFLECT];
    float refl[RT_MAX_REFLECT];
    int n = 0;
    Col tail;
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    for (;;) {
        const HitD h = closest_all<STATS>(sv, o, d, stk, st);
        if (h.tri == RT_MISS) {
            tail = sky();
            break;
        }
        const Surface s = surface_of(sv, h, o, d);
        const Col c = direct_light<STATS>(sv, s, sun, stk, st);
        if (depth < p.reflect_depth && s.reflectivity > 0.0f && n < RT_MAX_REFLECT) {
            direct[n] = c;
            refl[n] = s.reflectivity;
            ++n;
            const rt_f3 rd = d - 2.0f * s.normal * g_dot(d, s.normal);
            o = s.pos;
            d = rd;
            ++depth;
            continue;
        }
        tail = c;
        break;
    }
    for (int i = n - 1; i >= 0; --i) tail = cadd(direct[i], cmul(tail, refl[i]));
    return tail;
}

// get_tangent_vectors, src/main.rs:74-86.
__device__ __forceinline__ void tangent_vectors(rt_f3 n, rt_f3* t1, rt_f3* t2) {
    const rt_f3 arbitrary = (__builtin_fabsf(n.z) < 1e-6f) ? f3(0.0f, 0.0f, 1.0f) : f3(1.0f, 0.0f, 0.0f);
    *t1 = g_normalize(g_cross(n, arbitrary));
    *t2 = g_normalize(g_cross(n, *t1));
}

// Sub-sample offsets: grid 2 is the reference's fixed order, src/main.rs:322-364.
__device__ __forceinline__ void sample_offset(uint32_t grid, uint32_t s, float* ox, float* oy) {
    if (grid == 2) {
        *ox = (s & 1u) ? 0.75f : 0.25f;
        *oy = (s == 1u || s == 2u) ? 0.75f : 0.25f;
    } else {
        *ox = ((float)(s % grid) + 0.5f) / (float)grid;
        *oy = ((float)(s / grid) + 0.5f) / (float)grid;
    }
}

// One lane per sub-sample. A workgroup covers 256 / spp pixels of one tile (a square block in
// Morton order, so a wave's rays stay close together); the samples of a pixel sit in
// consecutive lanes and are averaged through LDS in the reference's order.
// Camera rays and their sun shadow rays are traced wave-coherently (packet_search); mirror and
// diffuse bounces, which scatter, are traced per lane.
// MODE 0: camera + shadow rays only (no reflective object or reflect_depth <= 1, diffuse_depth <= 1)
// MODE 1: + mirror chains     MODE 2: + diffuse scattering at depth 1
// Scatter direction k of a sample: cosine-weighted about the normal, src/main.rs:259-279, with the
// counter generator in place of rand::random.
__device__ __forceinline__ rt_f3 diffuse_dir(uint32_t seed, uint64_t sample_index, uint32_t k, rt_f3 normal, rt_f3 tang1,
                                             rt_f3 tang2) {
    const float sin_theta = __builtin_sqrtf(counter_rand(seed, sample_index, k, 0));
    const float cos_theta = __builtin_sqrtf(1.0f - sin_theta * sin_theta);
    const float psi = counter_rand(seed, sample_index, k, 1) * 2.0f * 3.14159265358979323846f;
    float sin_psi, cos_psi;
    rt_sincos(psi, &sin_psi, &cos_psi);
    const float a = sin_theta * cos_psi;
    const float b = sin_theta * sin_psi;
    const float c = cos_theta;
    return g_normalize(f3(a * tang1.x + b * tang2.x + c * normal.x, a * tang1.y + b * tang2.y + c * normal.y,
                          a * tang1.z + b * tang2.z + c * normal.z));
}

template <int MODE, bool STATS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_render(SceneView sv, DevRenderParams p,
                                                  const uint32_t* __restrict__ tile_list,
                                                  uint8_t* __restrict__ rgba8, uint32_t* __restrict__ hit_ids,
                                                  DevStats* __restrict__ stats) {
    static_assert(MODE == 0 || BLOCK == RT_BLOCK, "per-lane stacks are laid out for RT_BLOCK threads");
    // LDS: [per-lane stacks (MODE > 0)] [sample colours] [beam levels: one slab per wave]
    extern __shared__ __align__(16) unsigned char s_dyn[];
    constexpr int OFF_COL = MODE > 0 ? RT_STACK_LDS * RT_BLOCK * 8 : 0;
    constexpr int OFF_BEAM = OFF_COL + 3 * BLOCK * 4;
    uint2* s_stack = reinterpret_cast<uint2*>(s_dyn);
    float* s_col = reinterpret_cast<float*>(s_dyn + OFF_COL);
    uint2* s_beam = reinterpret_cast<uint2*>(s_dyn + OFF_BEAM) + (threadIdx.x >> 6) * p.beam_levels * 64;
    const uint32_t spp = p.spp;
    const uint32_t ppb = BLOCK / spp;                            // pixels per workgroup
    const uint32_t blocks_per_tile = (RT_TILE * RT_TILE) / ppb;
    const uint32_t local_tile = blockIdx.x / blocks_per_tile;
    const uint32_t block_in_tile = blockIdx.x % blocks_per_tile;
    const uint32_t tile = tile_list ? tile_list[local_tile] : local_tile;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const uint32_t s = threadIdx.x % spp;
    const uint32_t pix_in_tile = block_in_tile * ppb + threadIdx.x / spp;  // Morton index in the tile
    uint32_t mx = 0, my = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        mx |= ((pix_in_tile >> (2 * b)) & 1u) << b;
        my |= ((pix_in_tile >> (2 * b + 1)) & 1u) << b;
    }
    const uint32_t x = tx * RT_TILE + mx, y = ty * RT_TILE + my;
    const bool inside = x < p.width && y < p.height;

    uint2 stack_rest[LaneStack::REST];
    LaneStack stk;
    stk.lds = &s_stack[MODE > 0 ? threadIdx.x : 0];
    stk.spill = stack_rest;
    LaneStats st;
    float ox, oy;
    sample_offset(p.grid, s, &ox, &oy);
    // src/main.rs:314,322-330. aspect_ratio is an integer division in the reference.
    const float aspect = (float)(p.width / p.height);
    const float xx = (2.0f * ((float)x + ox) / (float)p.width - 1.0f) * aspect * 1.0f;
    const float yy = (1.0f - 2.0f * ((float)y + oy) / (float)p.height) * 1.0f;
    const rt_f3 o = f3(0.0f, 0.0f, 0.0f), d = f3(xx, yy, 1.0f);
    const uint64_t pix = (uint64_t)y * p.width + x;
    const uint64_t sample_index = pix * spp + s;
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);

    // depth 1: nearest hit and direct light, src/main.rs:93-241.
    const HitD h = closest_all_packet<STATS>(sv, o, d, inside, s_beam, st);
    const bool hit_any = inside && h.tri != RT_MISS;
    HitD hs = h;
    if (!hit_any) hs.tri = 0, hs.obj = 0;  // keep the loads below in range for idle lanes
    const Surface sf = surface_of(sv, hs, o, d);
    const rt_f3 sun_dir = g_normalize(sun - sf.pos);
    const bool lit = hit_any && g_dot(sf.normal, sun_dir) > 0.0f;
    const bool occluded = any_all_packet<STATS>(sv, sf.pos, sun_dir, lit, s_beam, st);
    Col color{0.0f, 0.0f, 0.0f};
    if (lit && !occluded) color = cmul(sf.color, g_dot(sf.normal, sun_dir) * sf.albedo);

    if (MODE >= 1 && hit_any) {
        Col reflect_color{0.0f, 0.0f, 0.0f};
        if (1u < p.reflect_depth && sf.reflectivity > 0.0f) {
            const rt_f3 rd = d - 2.0f * sf.normal * g_dot(d, sf.normal);
            reflect_color = cmul(mirror_chain<STATS>(sv, p, sf.pos, rd, 2, stk, st), sf.reflectivity);
        }
        Col diffuse_color{0.0f, 0.0f, 0.0f};
        if (MODE >= 2 && 1u < p.diffuse_depth) {
            rt_f3 tang1, tang2;
            tangent_vectors(sf.normal, &tang1, &tang2);
            for (uint32_t k = 0; k < p.diffuse_scatt; ++k) {
                const rt_f3 vel = diffuse_dir(p.seed, sample_index, k, sf.normal, tang1, tang2);
                const float light_power = g_dot(sf.normal, vel);
                const Col col = mirror_chain<STATS>(sv, p, sf.pos, vel, 2, stk, st);
                diffuse_color = cadd(diffuse_color, cmul(cmul_col(col, sf.color), light_power * sf.roughness));
            }
            diffuse_color = cdiv(diffuse_color, (float)p.diffuse_scatt);
        }
        color = cadd(color, reflect_color);
        color = cadd(color, diffuse_color);
    }
    // Gamma only for samples that hit something; the sky is returned as is, src/main.rs:302-310.
    if (hit_any) color = gamma22(color);
    else color = inside ? sky() : Col{0.0f, 0.0f, 0.0f};
    if (hit_ids && inside) hit_ids[sample_index] = h.tri;
    if (STATS) flush_stats(st, stats);

    // Average the pixel's samples in order (c1 + c2 + c3 + c4) / n, src/main.rs:365.
    s_col[3 * threadIdx.x + 0] = color.r;
    s_col[3 * threadIdx.x + 1] = color.g;
    s_col[3 * threadIdx.x + 2] = color.b;
    __syncthreads();
    if (s == 0 && inside) {
        Col acc = color;
        for (uint32_t k = 1; k < spp; ++k) {
            const float* c = &s_col[3 * (threadIdx.x + k)];
            acc = cadd(acc, Col{c[0], c[1], c[2]});
        }
        acc = cdiv(acc, (float)spp);
        const uint32_t px = to_u8(acc.r) | (to_u8(acc.g) << 8) | (to_u8(acc.b) << 16) | 0xff000000u;
        size_t off;
        if (p.compact) off = ((size_t)local_tile * RT_TILE * RT_TILE + (size_t)my * RT_TILE + mx) * 4;
        else off = ((size_t)y * p.width + x) * 4;
        *reinterpret_cast<uint32_t*>(rgba8 + off) = px;
    }
}

// ---------------------------------------------------------------------------------------------
// Camera + shadow frames as three kernels with a sort in between (used when no ray bounces):
//   k_wf_primary  camera rays (beam walk); per sample: hit point, hit id, state; a light-space
//                 sort key for every sample that needs a shadow ray
//   (radix sort of (key, sample slot) pairs)
//   k_wf_shadow   shadow rays in light-space order: the 64 rays of a wave run along (nearly) the
//                 same line towards the sun, so the beam walk stays thin even though their
//                 origins lie at very different depths
//   k_wf_resolve  shading, gamma, sub-sample average, RGBA8
// Shading repeats the megakernel's arithmetic exactly (same operations on the same stored values).
// ---------------------------------------------------------------------------------------------
#define RT_ST_HIT 1u
#define RT_ST_LIT 2u
#define RT_ST_OCC 4u
#define RT_ST_REDO 8u   // the whole word: the sample's nearest hit is left to the kernel that follows (RT_REDO)

// Workgroups are handed to the eight XCDs in turn, and each XCD has its own L2. The groups are
// renumbered so that runs of RT_XCD_GROUP consecutive work items (neighbouring rays, which walk the
// same part of the tree) go to one XCD. The launch is padded to a multiple of 8 * RT_XCD_GROUP
// groups; renumbered groups at or beyond the real count have nothing to do.
#ifndef RT_XCD_GROUP
#define RT_XCD_GROUP 64
#endif
__device__ __forceinline__ uint32_t xcd_block(uint32_t b) {
#if RT_XCD_GROUP > 0
    const uint32_t x = b & 7u, j = b >> 3;
    return ((j / RT_XCD_GROUP) * 8u + x) * RT_XCD_GROUP + j % RT_XCD_GROUP;
#else
    return b;
#endif
}
static inline uint32_t xcd_grid(uint32_t blocks) {
#if RT_XCD_GROUP > 0
    const uint32_t unit = 8u * RT_XCD_GROUP;
    return (blocks + unit - 1) / unit * unit;
#else
    return blocks;
#endif
}

template <int BLOCK>
struct PixelMap {
    uint32_t x, y, s, mx, my, local_tile, launch_tile, block;
    bool inside;
    __device__ __forceinline__ PixelMap(const DevRenderParams& p, const uint32_t* tile_list, bool remap = false) {
        const uint32_t spp = p.spp;
        const uint32_t ppb = BLOCK / spp;
        const uint32_t blocks_per_tile = (RT_TILE * RT_TILE) / ppb;
        block = remap ? xcd_block(blockIdx.x) : blockIdx.x;
        launch_tile = block / blocks_per_tile;
        local_tile = p.tile_first + launch_tile;
        const uint32_t block_in_tile = block % blocks_per_tile;
        const uint32_t tile = (tile_list && launch_tile < p.n_tiles) ? tile_list[local_tile] : local_tile;
        const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
        s = threadIdx.x % spp;
        const uint32_t pix_in_tile = block_in_tile * ppb + threadIdx.x / spp;
        mx = 0;
        my = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            mx |= ((pix_in_tile >> (2 * b)) & 1u) << b;
            my |= ((pix_in_tile >> (2 * b + 1)) & 1u) << b;
        }
        x = tx * RT_TILE + mx;
        y = ty * RT_TILE + my;
        inside = x < p.width && y < p.height;
    }
};

__device__ __forceinline__ rt_f3 camera_dir(const DevRenderParams& p, uint32_t x, uint32_t y, uint32_t s) {
    float ox, oy;
    sample_offset(p.grid, s, &ox, &oy);
    // src/main.rs:314,322-330. aspect_ratio is an integer division in the reference.
    const float aspect = (float)(p.width / p.height);
    const float xx = (2.0f * ((float)x + ox) / (float)p.width - 1.0f) * aspect * 1.0f;
    const float yy = (1.0f - 2.0f * ((float)y + oy) / (float)p.height) * 1.0f;
    return f3(xx, yy, 1.0f);
}

__device__ __forceinline__ uint32_t spread16(uint32_t v) {
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Light-space coordinates of a point, in the key's units (65536 across the scene's bounds), not yet
// clamped: the point projected from the sun onto the plane across the light's dominant axis; the
// points of one line to the sun get one pair of coordinates.
__device__ __forceinline__ void light_uv(const LightProj& lp, rt_f3 pos, float& qu, float& qv) {
    const float pw = pick3(pos, lp.w);
    const float rw = 1.0f / (pw - lp.sun_w);
    const float dw = pw - lp.c_w;
    const float u = ((pick3(pos, lp.a1) - lp.c_a1) - lp.uc * dw) * rw;
    const float v = ((pick3(pos, lp.a2) - lp.c_a2) - lp.vc * dw) * rw;
    qu = (u - lp.u0) * lp.uscale;
    qv = (v - lp.v0) * lp.vscale;
}

// Light-space sort key of a shadow ray's origin: its light-space coordinates interleaved. The key
// only says which cell's rays (and which cell's triangle list) a ray goes with, so the hardware
// reciprocal will do: it moves a coordinate by a few millionths of its 65,536 units, and a ray that
// lands in the neighbouring cell for it finds the same triangles there, since every triangle's patch
// is grown by half a unit for roundings of this kind (k_lcell_bin).
__device__ __forceinline__ uint32_t light_key(const LightProj& lp, rt_f3 pos) {
    const float pw = pick3(pos, lp.w);
    const float rw = __builtin_amdgcn_rcpf(pw - lp.sun_w);
    const float dw = pw - lp.c_w;
    const float fu = (((pick3(pos, lp.a1) - lp.c_a1) - lp.uc * dw) * rw - lp.u0) * lp.uscale;
    const float fv = (((pick3(pos, lp.a2) - lp.c_a2) - lp.vc * dw) * rw - lp.v0) * lp.vscale;
    const float qu = fminf(fmaxf(fu, 0.0f), 65535.0f);
    const float qv = fminf(fmaxf(fv, 0.0f), 65535.0f);
    uint32_t key = spread16((uint32_t)qu) | (spread16((uint32_t)qv) << 1);
    if (key == 0xFFFFFFFFu) key = 0xFFFFFFFEu;
    return key >> (32u - lp.key_bits);  // below 32 bits no key can equal the all-ones "no shadow ray" value
}

// Minimum waves per SIMD the camera- and shadow-ray kernels are compiled for: 8 keeps them at
// <= 64 VGPRs, which measured 10 % faster than the 81 VGPRs the compiler picks unconstrained.
#ifndef RT_WF_WAVES
#define RT_WF_WAVES 8
#endif
// Workgroup size of the camera- and shadow-ray kernels. Their waves do not cooperate, so a small
// group only changes how soon a finished wave's slot is handed to the next group.
#ifndef RT_WF_BLOCK
#define RT_WF_BLOCK 64
#endif
// The record of a camera ray's sample: hit point, triangle, state and the light-space key of its shadow ray.
__device__ __forceinline__ void primary_record(const SceneView& sv, const LightProj& lp, rt_f3 sun, bool inside, const HitD& h,
                                               rt_f3 o, rt_f3 d, uint32_t slot, float4* __restrict__ samp,
                                               uint32_t* __restrict__ state, uint32_t* __restrict__ keys) {
    const bool hit_any = inside && h.tri != RT_MISS;
    HitD hs = h;
    if (!hit_any) hs.tri = 0, hs.obj = 0;
    const Surface sf = surface_of(sv, hs, o, d);
    const rt_f3 sun_dir = g_normalize(sun - sf.pos);
    const bool lit = hit_any && g_dot(sf.normal, sun_dir) > 0.0f;
    samp[slot] = make_float4(sf.pos.x, sf.pos.y, sf.pos.z, __uint_as_float(h.tri));
    state[slot] = (hit_any ? RT_ST_HIT : 0u) | (lit ? RT_ST_LIT : 0u) | (hs.obj << 8);
    // Light-space key: the hit point projected from the sun onto the plane across the light's
    // dominant axis; points on one line to the sun get (nearly) one key.
    keys[slot] = lit ? light_key(lp, sf.pos) : 0xFFFFFFFFu;
}

template <bool STATS, int BLOCK>
__global__ __launch_bounds__(BLOCK, RT_WF_WAVES) void k_wf_primary(SceneView sv, DevRenderParams p, LightProj lp,
                                                      const uint32_t* __restrict__ tile_list,
                                                      float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                      uint32_t* __restrict__ keys, DevStats* __restrict__ stats) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint2* s_beam = reinterpret_cast<uint2*>(s_dyn) + (threadIdx.x >> 6) * p.beam_levels * 64;
    const PixelMap<BLOCK> pm(p, tile_list, true);
    if (pm.launch_tile >= p.n_tiles) return;  // padding of the renumbered launch
    LaneStats st;
    const rt_f3 o = f3(0.0f, 0.0f, 0.0f), d = camera_dir(p, pm.x, pm.y, pm.s);
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    const HitD h = closest_all_packet<STATS, true>(sv, o, d, pm.inside, s_beam, st);
    const uint32_t slot = pm.block * BLOCK + threadIdx.x;
    if (pm.inside && h.tri == RT_REDO) {
        // left to k_wf_primary_redo
        state[slot] = RT_ST_REDO;
        keys[slot] = 0xFFFFFFFFu;
    } else {
        primary_record(sv, lp, sun, pm.inside, h, o, d, slot, samp, state, keys);
    }
    if (STATS) flush_stats(st, stats);
}

// ---------------------------------------------------------------------------------------------
// Camera rays by projection. All camera rays leave one point, and the sub-samples form a regular
// grid on the image plane, so the rays that can meet a triangle are the samples inside the
// triangle's projected bounding box. Triangle by triangle, the samples of that box are put through
// the reference's own ray-triangle arithmetic (the same camera_dir, the same tri_intersect, the
// object's box test of src/main.rs:101), and the nearest hit of a sample is kept as the minimum of
// (t, triangle id) in one 64-bit word: positive floats order like their bit patterns, and the lower
// id wins a tie on t, which is the reference's rule (it scans ids upward with a strict compare).
// Only for scenes whose objects have no trailing n % 8 triangles (those do not lower `closest`,
// which makes the result depend on the object order) and for triangles in front of the camera.
// One wave takes RT_RASTER_TRIS consecutive triangles: one lane each loads a triangle and works out
// its box of samples, then the wave steps over each box as blocks of 8 x 8 samples with the triangle's
// data broadcast from its lane. (Blocks of 16 x 4 ... 2 x 32, chosen per triangle, need 9 % fewer steps
// and were 4 % slower.)
// ---------------------------------------------------------------------------------------------
#ifndef RT_RASTER_TRIS
#define RT_RASTER_TRIS 32
#endif
#define RT_ZKEY_NONE 0xFFFFFFFFFFFFFFFFull

// Sub-sample number of position (subx, suby) on the n x n grid inside a pixel: the inverse of
// sample_offset.
__device__ __forceinline__ uint32_t sample_of(uint32_t grid, uint32_t subx, uint32_t suby) {
    if (grid == 2) return subx == suby ? subx : (subx ? 3u : 2u);
    return suby * grid + subx;
}

// Image-plane coordinates of the sample columns and rows: d = (xx, yy, 1) of camera_dir depends on
// the sample's column for xx and on its row for yy only. With the origin at zero the object's box
// test (box_intersection_check) splits the same way: the slab interval along x depends on the column,
// the one along y on the row, the one along z on nothing. All of it is tabulated once per frame with
// the very operations of camera_dir and box_intersection_check, which leaves the triangle's own
// division as the only one in the inner loop.
// Layout, L = gw + gh lines (gw = width * grid columns first, then gh = height * grid rows):
//   tab[line]                         xx of a column / yy of a row
//   tab[L * (1 + 2 * obj) + line]     near end of the object's slab interval along that axis
//   tab[L * (2 + 2 * obj) + line]     far end
__global__ void k_camera_table(SceneView sv, DevRenderParams p, float* __restrict__ tab) {
    const uint32_t g = p.grid, gw = p.width * g, gh = p.height * g, L = gw + gh;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const bool col = i < gw;
    const uint32_t r = col ? i : i - gw;
    const float c = col ? camera_dir(p, r / g, 0u, sample_of(g, r % g, 0u)).x
                        : camera_dir(p, 0u, r / g, sample_of(g, 0u, r % g)).y;
    tab[i] = c;
    const float inv = 1.0f / c;
    for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
        const DevObject& ob = sv.objs[oi];
        const float lo = col ? ob.bbmin[0] : ob.bbmin[1], hi = col ? ob.bbmax[0] : ob.bbmax[1];
        const float t1 = (lo - 0.0f) * inv, t2 = (hi - 0.0f) * inv;
        tab[(size_t)L * (1 + 2 * oi) + i] = g_min(t1, t2);
        tab[(size_t)L * (2 + 2 * oi) + i] = g_max(t1, t2);
    }
}

#ifndef RT_RASTER_SLOP
#define RT_RASTER_SLOP 0.125f   // samples: rounding of the projected vertices (hardware reciprocal, float products)
#endif
#define RT_RASTER_QUEUE 128     // candidate samples a wave holds before it tests 64 of them
#ifdef RT_DEBUG_RASTER
// Debug build only: triangle id, sample column, sample row to report on.
__device__ uint32_t g_dbg[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
#endif
// True when the shard owns one of the tiles that the sample rectangle [u0, u1] x [w0, w1] touches
// (tile (tx, ty) belongs to shard (tx + ty) % count). gs = log2(samples per pixel side).
__device__ __forceinline__ bool shard_touches(const DevRenderParams& p, uint32_t gs, int u0, int u1, int w0, int w1) {
    if (p.shard_count <= 1) return true;
    const uint32_t tx0 = ((uint32_t)u0 >> gs) / RT_TILE, tx1 = ((uint32_t)u1 >> gs) / RT_TILE;
    const uint32_t ty0 = ((uint32_t)w0 >> gs) / RT_TILE, ty1 = ((uint32_t)w1 >> gs) / RT_TILE;
    // A run of `count` consecutive diagonals holds every shard.
    if (tx1 - tx0 + ty1 - ty0 + 1 >= p.shard_count) return true;
    // dsum % count through the reciprocal the host prepared (tile sums stay far below 2^16).
    for (uint32_t dsum = tx0 + ty0; dsum <= tx1 + ty1; ++dsum) {
        const uint32_t q = (uint32_t)(((uint64_t)dsum * p.shard_magic) >> 32);
        if (dsum - q * p.shard_count == p.shard_index) return true;
    }
    return false;
}

__device__ __forceinline__ float bcast(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// |a| x |b| with every product taken positive: bounds the magnitude of the terms of a cross product.
__device__ __forceinline__ rt_f3 abs_cross(rt_f3 a, rt_f3 b) {
    const float ax = __builtin_fabsf(a.x), ay = __builtin_fabsf(a.y), az = __builtin_fabsf(a.z);
    const float bx = __builtin_fabsf(b.x), by = __builtin_fabsf(b.y), bz = __builtin_fabsf(b.z);
    return f3(ay * bz + az * by, az * bx + ax * bz, ax * by + ay * bx);
}

// What a camera ray's exact triangle test (tri_intersect with the origin at zero) can accept, bounded
// from outside by three half-planes of the image plane. For d = (dx, dy, 1) the test's quantities
// are linear in d:  a = d . (e2 x e1)... precisely, with s = -v0:
//     U(d) = s . (d x e2) = d . (e2 x s)      (u = U / a)
//     V(d) = d . (s x e1)                     (v = V / a)
//     A(d) = e1 . (d x e2) = d . (e2 x e1)    (a)
//     T    = e2 . (s x e1)                    (t = T / a; no d in it: one value per triangle)
// A sample is accepted only if t > 0, so sign(a) = sign(T) =: sg, and then u >= 0, v >= 0,
// u + v <= 1 say  sg U >= 0, sg V >= 0, sg (A - U - V) >= 0  up to rounding. Every computed
// quantity differs from its exact value by at most 8 u M (u = 2^-24; M = the same expression with
// every product taken positive, over |dx| <= Dx, |dy| <= Dy): cross products are a fused
// multiply-subtract of a rounded product (3 u per component), dot products three more roundings.
// The half-plane functions below are evaluated in float as fma(dx, gx, fma(dy, gy, gz)) from
// float cross products: 5 u M more. A threshold of 16 u M (32 u M for the third, which sums three
// quantities and a rounded u + v) plus a term that covers products underflowing to zero is
// folded into gz, so that  F >= 0  holds for every sample the exact test can accept.
// In barycentric coordinates of the projected triangle the three functions are lambda_b h_b,
// lambda_c h_c, lambda_a h_a with h_k = |T| / z_k, so the accepted region is the triangle grown by
// eps_k = threshold_k / h_k in each coordinate, and its bounding box is the vertices' box grown by
// at most (eps_a + eps_b + eps_c) of its own extent: `grow`. A triangle whose |T| is not safely
// above its own rounding error (seen edge-on from the camera) gets grow = infinity.
struct RasterSetup {
    rt_f3 gu, gv, gw;   // F_k(d) = fma(dx, g.x, fma(dy, g.y, g.z)), thresholds folded in
    float grow;         // relative growth of the projected vertices' box that holds every accepted sample
    bool visible;       // false: T is zero or NaN, no camera ray can be accepted
};
__device__ __forceinline__ RasterSetup raster_setup(rt_f3 e1, rt_f3 e2, rt_f3 v0, float Dx, float Dy, float zmax) {
    RasterSetup rs;
    const rt_f3 s = f3(0.0f, 0.0f, 0.0f) - v0;
    const rt_f3 r = s_cross(s, e1);
    const float T = s_dot(e2, r);             // exactly the test's e2 . r
    rs.visible = (T > 0.0f) || (T < 0.0f);
    const float sg = T < 0.0f ? -1.0f : 1.0f;
    const rt_f3 gu = s_cross(e2, s), gv = r, ga = s_cross(e2, e1);
    // magnitudes
    const rt_f3 qh = f3(Dy * __builtin_fabsf(e2.z) + __builtin_fabsf(e2.y), __builtin_fabsf(e2.x) + Dx * __builtin_fabsf(e2.z),
                        Dx * __builtin_fabsf(e2.y) + Dy * __builtin_fabsf(e2.x));
    const float Mu = __builtin_fabsf(s.x) * qh.x + __builtin_fabsf(s.y) * qh.y + __builtin_fabsf(s.z) * qh.z;
    const float Ma = __builtin_fabsf(e1.x) * qh.x + __builtin_fabsf(e1.y) * qh.y + __builtin_fabsf(e1.z) * qh.z;
    const rt_f3 rh = abs_cross(s, e1);
    const float Mv = Dx * rh.x + Dy * rh.y + rh.z;
    const float Mt = __builtin_fabsf(e2.x) * rh.x + __builtin_fabsf(e2.y) * rh.y + __builtin_fabsf(e2.z) * rh.z;
    const float U16 = 9.5367431640625e-7f;    // 16 u = 2^-20
    const float tiny = 1e-37f + 1e-30f * Ma;  // a product f * U that underflows to zero passes the sign tests
    const float thr_u = U16 * Mu + tiny, thr_v = U16 * Mv + tiny, thr_w = 2.0f * U16 * (Ma + Mu + Mv) + tiny;
    rs.gu = f3(sg * gu.x, sg * gu.y, sg * gu.z + thr_u);
    rs.gv = f3(sg * gv.x, sg * gv.y, sg * gv.z + thr_v);
    rs.gw = f3(sg * (ga.x - gu.x - gv.x), sg * (ga.y - gu.y - gv.y), sg * (ga.z - gu.z - gv.z) + thr_w);
    // |T| from below, and the growth of the box. z_k <= zmax.
    const float Tlow = __builtin_fabsf(T) - U16 * Mt;
    rs.grow = Tlow > 0.0f ? (thr_u + thr_v + thr_w) * zmax / Tlow * 1.0001f : RT_INF;
    if (!(rs.grow >= 0.0f)) rs.grow = RT_INF;   // NaN from overflowing magnitudes
    return rs;
}

// One triangle made ready for projection: its data, its object, the box of samples that holds every
// sample the exact test can accept, and the half-planes.
struct RasterTri {
    rt_f3 e1, e2, v0;
    uint32_t objword;          // object index; bit 31: the object's box test is true for every camera ray
    int u0, u1, w0, w1;        // empty (u0 > u1) when nothing can be accepted
    RasterSetup rs;
};
__device__ __forceinline__ RasterTri raster_prepare(const SceneView& sv, const DevRenderParams& p, uint32_t id, int gw,
                                                    int gh, float aspect, uint32_t& st_bytes) {
    RasterTri rt;
    uint32_t moi = 0;
    while (id >= sv.objs[moi].tri_base + sv.objs[moi].n_tris) ++moi;
    const float* t9 = sv.tri9 + 9 * (size_t)id;
    const rt_f3 me1 = ld3(t9), me2 = ld3(t9 + 3), mv0 = ld3(t9 + 6);
    st_bytes += 36;
    rt.e1 = me1, rt.e2 = me2, rt.v0 = mv0;
    const rt_f3 vb = mv0 + me1, vc = mv0 + me2;
    int mu0 = 0, mu1 = gw - 1, mw0 = 0, mw1 = gh - 1;
    const float zmin = fminf(mv0.z, fminf(vb.z, vc.z)), zmax = fmaxf(mv0.z, fmaxf(vb.z, vc.z));
    // |dx| <= aspect and |dy| <= 1 on the whole sample grid (src/main.rs:322-324).
    rt.rs = raster_setup(me1, me2, mv0, aspect, 1.0f, zmax * 1.000001f);
    const DevObject& ob = sv.objs[moi];
    const float ax = (float)gw * 0.5f / aspect, ay = (float)gh * 0.5f;
    const float lim = 1.0e9f;
    if (zmin > 1e-3f && rt.rs.grow < RT_INF) {
        // Sample column of image-plane x: xx = (2 (x + o) / W - 1) * aspect, sample index
        // (x + o) * g - 0.5. The hardware reciprocal will do: RT_RASTER_SLOP covers its error and
        // the rounding of the products many times over.
        const float ia = __builtin_amdgcn_rcpf(mv0.z), ib = __builtin_amdgcn_rcpf(vb.z), ic = __builtin_amdgcn_rcpf(vc.z);
        const float xa = mv0.x * ia, xb = vb.x * ib, xc = vc.x * ic;
        const float ya = mv0.y * ia, yb = vb.y * ib, yc = vc.y * ic;
        const float fu0 = (fminf(xa, fminf(xb, xc)) + aspect) * ax - 0.5f;
        const float fu1 = (fmaxf(xa, fmaxf(xb, xc)) + aspect) * ax - 0.5f;
        const float fw0 = (1.0f - fmaxf(ya, fmaxf(yb, yc))) * ay - 0.5f;
        const float fw1 = (1.0f - fminf(ya, fminf(yb, yc))) * ay - 0.5f;
        // The box that holds every sample the exact test can accept: the vertices' box grown by
        // `grow` of its extent, the slop, and a relative part for sample coordinates in the
        // tens of thousands.
        const float mx = (fu1 - fu0) * rt.rs.grow + RT_RASTER_SLOP + 1e-6f * fmaxf(__builtin_fabsf(fu0), __builtin_fabsf(fu1));
        const float my = (fw1 - fw0) * rt.rs.grow + RT_RASTER_SLOP + 1e-6f * fmaxf(__builtin_fabsf(fw0), __builtin_fabsf(fw1));
        mu0 = max(mu0, (int)floorf(fminf(fmaxf(fu0 - mx, -lim), lim)));
        mu1 = min(mu1, (int)ceilf(fminf(fmaxf(fu1 + mx, -lim), lim)));
        mw0 = max(mw0, (int)floorf(fminf(fmaxf(fw0 - my, -lim), lim)));
        mw1 = min(mw1, (int)ceilf(fminf(fmaxf(fw1 + my, -lim), lim)));
    }
    // The reference tests a triangle only for rays that pass the slab test of its leaf: the box of
    // the eight consecutive triangles it is packed with (src/main.rs:103-111, src/simd_accel.rs:342-377).
    // With the origin at zero that test is  max_k(min_k * inv_k) <= min_k(max_k * inv_k),  products of
    // rounded reciprocals: it can only pass for directions inside the projection of the leaf's box
    // (all of it in front of the camera) widened by a few units of rounding, far inside the slop.
    // That bounds the samples of a triangle whatever its own shape, edge-on ones included. (The
    // leftover leaves at the end of an object go through the looser scalar test, src/main.rs:134-136,
    // and are not bounded this way.)
    const uint32_t leaf = (id - ob.tri_base) >> 3, n_leaves = ob.n_tris >> 3;
    if (leaf < (n_leaves & ~7u)) {
        const float* lb = sv.leaf_bounds + 6 * (size_t)(ob.leaf_base + leaf);
        const rt_f3 lo = ld3(lb), hi = ld3(lb + 3);
        st_bytes += 24;
        if (lo.z > 1e-3f) {
            const float in = __builtin_amdgcn_rcpf(lo.z), ifar = __builtin_amdgcn_rcpf(hi.z);
            const float bx0 = lo.x * (lo.x < 0.0f ? in : ifar), bx1 = hi.x * (hi.x > 0.0f ? in : ifar);
            const float by0 = lo.y * (lo.y < 0.0f ? in : ifar), by1 = hi.y * (hi.y > 0.0f ? in : ifar);
            const float fu0 = (bx0 + aspect) * ax - 0.5f, fu1 = (bx1 + aspect) * ax - 0.5f;
            const float fw0 = (1.0f - by1) * ay - 0.5f, fw1 = (1.0f - by0) * ay - 0.5f;
            const float mx = RT_RASTER_SLOP + 1e-6f * fmaxf(__builtin_fabsf(fu0), __builtin_fabsf(fu1));
            const float my = RT_RASTER_SLOP + 1e-6f * fmaxf(__builtin_fabsf(fw0), __builtin_fabsf(fw1));
            mu0 = max(mu0, (int)floorf(fminf(fmaxf(fu0 - mx, -lim), lim)));
            mu1 = min(mu1, (int)ceilf(fminf(fmaxf(fu1 + mx, -lim), lim)));
            mw0 = max(mw0, (int)floorf(fminf(fmaxf(fw0 - my, -lim), lim)));
            mw1 = min(mw1, (int)ceilf(fminf(fmaxf(fw1 + my, -lim), lim)));
        }
    }
    if (!rt.rs.visible) mu1 = mu0 - 1;
    // an object's trailing n % 8 triangles are not projected: the tile kernel tests them sample by sample
    if (id - ob.tri_base >= ob.n_tris - ob.n_rest) mu1 = mu0 - 1;
    rt.u0 = mu0, rt.u1 = mu1, rt.w0 = mw0, rt.w1 = mw1;
    // The object's box test (src/main.rs:101, src/objects.rs:322-333) with the origin at zero is
    // g_min(t_near, 0) <= t_far. When the box holds the view axis on both lateral axes and reaches
    // in front of the camera, one end of every lateral slab interval is >= 0 whatever the
    // direction, and so is the far end along z: t_far >= 0, the test is true for every sample.
    const bool always = ob.bbmin[0] < 0.0f && ob.bbmax[0] > 0.0f && ob.bbmin[1] < 0.0f && ob.bbmax[1] > 0.0f &&
                        ob.bbmax[2] > 0.0f && ob.bbmin[2] > -RT_INF;
    rt.objword = moi | (always ? 0x80000000u : 0u);
    return rt;
}

// box_intersection_check (src/objects.rs:322-333) for the camera ray of sample column su, row sw
// against object oi, from the tabulated slab intervals.
__device__ __forceinline__ bool raster_object_box(const SceneView& sv, const float* __restrict__ tab, uint32_t L, int gw,
                                                  uint32_t oi, uint32_t su, uint32_t sw) {
    const DevObject& ob = sv.objs[oi];
    const float tz1 = (ob.bbmin[2] - 0.0f) * (1.0f / 1.0f), tz2 = (ob.bbmax[2] - 0.0f) * (1.0f / 1.0f);
    const float nz = g_min(tz1, tz2), fz = g_max(tz1, tz2);
    const float* near_tab = tab + (size_t)L * (1 + 2 * oi);
    const float* far_tab = tab + (size_t)L * (2 + 2 * oi);
    const float nx = near_tab[su], fx = far_tab[su], ny = near_tab[gw + sw], fy = far_tab[gw + sw];
    const float t_near = g_max(g_max(nx, ny), nz);
    const float t_far = g_min(g_min(fx, fy), fz);
    return g_min(t_near, 0.0f) <= t_far;
}

// Counters of an instrumented run: tri = samples put through the triangle test, box = object box
// tests, node = blocks of 8 x 8 samples stepped over, bytes = triangle data, table lines and
// nearest-hit words asked for.
// One wave takes RT_RASTER_TRIS consecutive triangles, one lane each for the set-up (data, box of
// samples, half-planes). It then steps over each triangle's box in blocks of 8 x 8 samples; a sample
// that passes the three half-planes is a candidate and goes into a queue in LDS (its ray direction,
// its coordinates, its triangle's slot); whenever 64 candidates wait, the wave puts them through the
// exact test together, every lane with its own triangle's data from LDS. The exact test, with its
// division, thus runs on full waves of samples that are (nearly all) inside their triangles, instead
// of on every sample of every box.
struct RasterLds {
    float tri[RT_RASTER_TRIS][12];            // e1 xyz, e2 xyz, v0 xyz, id, object word, -
    float4 queue[RT_RASTER_QUEUE];            // dx, dy, (su | sw << 16), slot
};

template <bool STATS>
__global__ __launch_bounds__(64, 8) void k_raster(SceneView sv, DevRenderParams p, uint32_t n_tris_total,
                                                  const float* __restrict__ tab,
                                                  unsigned long long* __restrict__ zbuf,
                                                  DevStats* __restrict__ stats) {
    __shared__ RasterLds lds;
    LaneStats st;
    const uint32_t lane = threadIdx.x;
    const uint32_t g = p.grid, gs = 31u - (uint32_t)__builtin_clz(g);  // grid is a power of two
    const float aspect = (float)(p.width / p.height);
    const int gw = (int)(p.width * g), gh = (int)(p.height * g);
    const uint32_t L = (uint32_t)(gw + gh);
    // Lane l < RT_RASTER_TRIS prepares triangle l of the wave.
    const uint32_t my_id = blockIdx.x * RT_RASTER_TRIS + lane;
    const bool mine = lane < RT_RASTER_TRIS && my_id < n_tris_total;
    int mu0 = 0, mu1 = -1, mw0 = 0, mw1 = -1;
    RasterSetup rs;
    rs.gu = rs.gv = rs.gw = f3(0.0f, 0.0f, 0.0f);
    rs.grow = 0.0f;
    rs.visible = false;
    if (mine) {
        const RasterTri rt = raster_prepare(sv, p, my_id, gw, gh, aspect, st.bytes);
        rs = rt.rs;
        mu0 = rt.u0, mu1 = rt.u1, mw0 = rt.w0, mw1 = rt.w1;
        float* td = lds.tri[lane];
        td[0] = rt.e1.x, td[1] = rt.e1.y, td[2] = rt.e1.z;
        td[3] = rt.e2.x, td[4] = rt.e2.y, td[5] = rt.e2.z;
        td[6] = rt.v0.x, td[7] = rt.v0.y, td[8] = rt.v0.z;
        td[9] = __uint_as_float(my_id);
        td[10] = __uint_as_float(rt.objword);
        td[11] = 0.0f;
    }
    __syncthreads();
    unsigned long long todo =
        __ballot(mine && rs.visible && mu0 <= mu1 && mw0 <= mw1 && shard_touches(p, gs, mu0, mu1, mw0, mw1));
    const rt_f3 o = f3(0.0f, 0.0f, 0.0f);
    uint32_t q_head = 0, q_tail = 0;   // wave-uniform: candidates lie at queue[(q_head .. q_tail) % RT_RASTER_QUEUE]

    // Puts up to 64 waiting candidates through the exact test: the reference's arithmetic on the
    // reference's ray, acceptance rule !failed && t > 0 (src/main.rs:117), then the object's box test.
    auto test_candidates = [&](uint32_t n) {
        const bool have = lane < n;
        const float4 e = lds.queue[(q_head + (have ? lane : 0u)) % RT_RASTER_QUEUE];
        q_head += n;
        if (!have) return;
        const float* td = lds.tri[__float_as_uint(e.w)];
        const rt_f3 e1 = f3(td[0], td[1], td[2]), e2 = f3(td[3], td[4], td[5]), v0 = f3(td[6], td[7], td[8]);
        const uint32_t id = __float_as_uint(td[9]), ow = __float_as_uint(td[10]);
        const rt_f3 d = f3(e.x, e.y, 1.0f);
        if (STATS) st.tri += 1;
        const rt_tri_eval ev = tri_intersect(e1, e2, v0, o, d);
        if (ev.failed || !(ev.t > 0.0f) || !(ev.t < RT_INF)) return;
        const uint32_t su = __float_as_uint(e.z) & 0xFFFFu, sw = __float_as_uint(e.z) >> 16;
        if (!(ow & 0x80000000u)) {
            if (STATS) st.box += 1, st.bytes += 16;
            if (!raster_object_box(sv, tab, L, gw, ow, su, sw)) return;
        }
        // the reference tests the triangle only if the ray passes the box of its leaf
        if (!ref_leaf_gate(sv, sv.objs[ow & 0x7FFFFFFFu], id, o, g_recip(d))) return;
        if (STATS) st.bytes += 8;
        const uint32_t x = su >> gs, y = sw >> gs;
        const uint32_t smp = sample_of(g, su & (g - 1u), sw & (g - 1u));
        const unsigned long long key = ((unsigned long long)__float_as_uint(ev.t) << 32) | id;
        atomicMin(&zbuf[((size_t)y * p.width + x) * p.spp + smp], key);
    };

    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const rt_f3 gu = f3(bcast(rs.gu.x, l), bcast(rs.gu.y, l), bcast(rs.gu.z, l));
        const rt_f3 gv = f3(bcast(rs.gv.x, l), bcast(rs.gv.y, l), bcast(rs.gv.z, l));
        const rt_f3 gwv = f3(bcast(rs.gw.x, l), bcast(rs.gw.y, l), bcast(rs.gw.z, l));
        const int u0 = __builtin_amdgcn_readlane(mu0, l), u1 = __builtin_amdgcn_readlane(mu1, l);
        const int w0 = __builtin_amdgcn_readlane(mw0, l), w1 = __builtin_amdgcn_readlane(mw1, l);
        // Blocks of 8 x 8 samples, 64 at a time with one lane per block: a block all of whose samples
        // lie outside one of the half-planes is dropped before the wave looks at its samples. The
        // table's columns increase with the sample index and its rows decrease, fma is monotone in
        // each argument and so is its rounding, so a function evaluated at the block's best corner
        // bounds what the samples of the block will evaluate to.
        const uint32_t nbx = (uint32_t)(u1 - u0) / 8u + 1u, nby = (uint32_t)(w1 - w0) / 8u + 1u, nblocks = nbx * nby;
        for (uint32_t base = 0; base < nblocks; base += 64u) {
            const uint32_t b = base + lane;
            const uint32_t brow = b / nbx, bcol = b - brow * nbx;
            const int mbx = u0 + 8 * (int)bcol, mby = w0 + 8 * (int)brow;
            bool keep = b < nblocks;
            if (keep) {
                const int ex = min(mbx + 7, u1), ey = min(mby + 7, w1);
                keep = shard_touches(p, gs, mbx, ex, mby, ey);
                const float dx_lo = tab[mbx], dx_hi = tab[ex], dy_hi = tab[gw + mby], dy_lo = tab[gw + ey];
                if (STATS) st.bytes += 16;
                const float fu = __builtin_fmaf(gu.x < 0.0f ? dx_lo : dx_hi, gu.x, __builtin_fmaf(gu.y < 0.0f ? dy_lo : dy_hi, gu.y, gu.z));
                const float fv = __builtin_fmaf(gv.x < 0.0f ? dx_lo : dx_hi, gv.x, __builtin_fmaf(gv.y < 0.0f ? dy_lo : dy_hi, gv.y, gv.z));
                const float fw = __builtin_fmaf(gwv.x < 0.0f ? dx_lo : dx_hi, gwv.x, __builtin_fmaf(gwv.y < 0.0f ? dy_lo : dy_hi, gwv.y, gwv.z));
                keep = keep && !(fu < 0.0f) && !(fv < 0.0f) && !(fw < 0.0f);
            }
            unsigned long long bm = __ballot(keep);
            while (bm) {
                const int j = __builtin_ctzll(bm);
                bm &= bm - 1ull;
                const int bx = __builtin_amdgcn_readlane(mbx, j), by = __builtin_amdgcn_readlane(mby, j);
                const int su = bx + (int)(lane & 7u), sw = by + (int)(lane >> 3);
                if (STATS && lane == 0) st.node += 1;
                // A shard keeps the samples of its own tiles only.
                const bool in = su <= u1 && sw <= w1 && shard_touches(p, gs, su, su, sw, sw);
                const float dx = tab[in ? su : u1], dy = tab[gw + (in ? sw : w1)];
                if (STATS) st.bytes += in ? 8 : 0;
                // !(F < 0) rather than F >= 0: a function that overflowed to NaN must not reject.
                const bool cand = in && !(__builtin_fmaf(dx, gu.x, __builtin_fmaf(dy, gu.y, gu.z)) < 0.0f) &&
                                  !(__builtin_fmaf(dx, gv.x, __builtin_fmaf(dy, gv.y, gv.z)) < 0.0f) &&
                                  !(__builtin_fmaf(dx, gwv.x, __builtin_fmaf(dy, gwv.y, gwv.z)) < 0.0f);
                const unsigned long long cm = __ballot(cand);
                if (cm == 0ull) continue;
                if (cand) {
                    const uint32_t pos = q_tail + (uint32_t)__builtin_popcountll(cm & ((1ull << lane) - 1ull));
                    lds.queue[pos % RT_RASTER_QUEUE] =
                        make_float4(dx, dy, __uint_as_float((uint32_t)su | ((uint32_t)sw << 16)), __uint_as_float((uint32_t)l));
                }
                q_tail += (uint32_t)__builtin_popcountll(cm);
                if (q_tail - q_head >= 64u) {
                    __syncthreads();
                    test_candidates(64u);
                    __syncthreads();
                }
            }
        }
    }
    if (q_tail != q_head) {
        __syncthreads();
        test_candidates(q_tail - q_head);
    }
    if (STATS) flush_stats(st, stats);
}

// ---------------------------------------------------------------------------------------------
// Projection by screen tile. The kernel above keeps every sample's nearest hit with an atomic
// minimum in device memory: five layers of triangles make 170 M of them on the benchmark frame, and
// the words do not stay in L2 between layers. Here the triangles are first put into lists per tile of
// RT_RTILE x RT_RTILE samples (count, prefix sum, fill: the list of a tile holds every triangle whose
// box of samples and whose half-planes meet it), and then one wave owns one tile: the nearest-hit
// words of its 1024 samples live in LDS, the wave goes through the tile's list 64 triangles at a time
// (one lane sets one triangle up, then the wave steps over the blocks of the triangle's box inside the
// tile, queues the samples that pass the half-planes and puts 64 of them through the exact test at a
// time), and at the end writes the samples' records itself: hit point, id, state, light-space key,
// exactly what k_wf_primary stores. No atomics on device memory, no nearest-hit buffer, no record pass.
// A tile's 1024 samples are 1024 consecutive record slots (the pixels of a tile are numbered along a
// Morton curve and the samples of a pixel are consecutive).
// ---------------------------------------------------------------------------------------------
#define RT_RTILE 32u
#ifndef RT_RTILE_TRIS
#define RT_RTILE_TRIS 64   // triangles of a tile's list a wave sets up at a time
#endif

// A triangle made ready once per frame by the counting pass of the binning and read back by the
// list-writing pass and by the tile kernel: 96 bytes, six float4.
//   0: gu.xyz gv.x   1: gv.yz gw.xy   2: gw.z, u0 | u1 << 16, w0 | w1 << 16, object word
//   3: e1.xyz e2.x   4: e2.yz v0.xy   5: v0.z - - -
// The box is stored clipped to the sample grid (fewer than 65,536 columns and rows); an empty box as
// u0 = 1, u1 = 0.
#define RT_RREC 6
#define RT_BIN_HASH_BITS 9
#define RT_BIN_HASH (1u << RT_BIN_HASH_BITS)   // slots of a workgroup's table of tiles
__device__ __forceinline__ void raster_rec_store(float4* __restrict__ r, const RasterTri& rt) {
    const bool any = rt.u0 <= rt.u1 && rt.w0 <= rt.w1;
    const uint32_t bu = any ? ((uint32_t)rt.u0 | ((uint32_t)rt.u1 << 16)) : 1u;
    const uint32_t bw = any ? ((uint32_t)rt.w0 | ((uint32_t)rt.w1 << 16)) : 1u;
    r[0] = make_float4(rt.rs.gu.x, rt.rs.gu.y, rt.rs.gu.z, rt.rs.gv.x);
    r[1] = make_float4(rt.rs.gv.y, rt.rs.gv.z, rt.rs.gw.x, rt.rs.gw.y);
    r[2] = make_float4(rt.rs.gw.z, __uint_as_float(bu), __uint_as_float(bw), __uint_as_float(rt.objword));
    r[3] = make_float4(rt.e1.x, rt.e1.y, rt.e1.z, rt.e2.x);
    r[4] = make_float4(rt.e2.y, rt.e2.z, rt.v0.x, rt.v0.y);
    r[5] = make_float4(rt.v0.z, 0.0f, 0.0f, 0.0f);
}
// The half-planes and the box of a stored record (what the binning needs).
__device__ __forceinline__ void raster_rec_head(const float4* __restrict__ r, RasterSetup& rs, int& u0, int& u1, int& w0,
                                                int& w1, uint32_t& objword) {
    const float4 a = r[0], b = r[1], c = r[2];
    rs.gu = f3(a.x, a.y, a.z), rs.gv = f3(a.w, b.x, b.y), rs.gw = f3(b.z, b.w, c.x);
    rs.grow = 0.0f;
    rs.visible = true;
    const uint32_t bu = __float_as_uint(c.y), bw = __float_as_uint(c.z);
    u0 = (int)(bu & 0xFFFFu), u1 = (int)(bu >> 16), w0 = (int)(bw & 0xFFFFu), w1 = (int)(bw >> 16);
    objword = __float_as_uint(c.w);
}

// Raster tiles of the frame: ceil(gw / 32) x ceil(gh / 32), numbered row by row.
__device__ __forceinline__ bool raster_tile_owned(const DevRenderParams& p, uint32_t gs, uint32_t rtx, uint32_t rty) {
    if (p.shard_count <= 1) return true;
    const uint32_t ptx = ((rtx * RT_RTILE) >> gs) / RT_TILE, pty = ((rty * RT_RTILE) >> gs) / RT_TILE;
    return (ptx + pty) % p.shard_count == p.shard_index;
}

// True when no sample of the rectangle [x0, x1] x [y0, y1] can pass the three half-planes: each
// function is linear in (dx, dy), so its largest value over the rectangle is taken at a corner, and
// the rectangle is outside as soon as one function is negative there. The corners are taken 1/16
// sample beyond the rectangle, from the closed form of the table (dx = (column + 0.5) / ax - aspect,
// dy = 1 - (row + 0.5) / ay): the tabulated values differ from it by a few units of their last place,
// under 0.02 sample for the widest frames the projection takes, so the corner values bound the table's.
struct RasterGrid {
    float inv_ax, inv_ay, aspect;
};
__device__ __forceinline__ bool raster_rect_outside(const RasterSetup& rs, const RasterGrid& gr, int x0, int x1, int y0,
                                                    int y1) {
    const float dx_lo = ((float)x0 + 0.4375f) * gr.inv_ax - gr.aspect, dx_hi = ((float)x1 + 0.5625f) * gr.inv_ax - gr.aspect;
    const float dy_hi = 1.0f - ((float)y0 + 0.4375f) * gr.inv_ay, dy_lo = 1.0f - ((float)y1 + 0.5625f) * gr.inv_ay;
    const float fu = __builtin_fmaf(rs.gu.x < 0.0f ? dx_lo : dx_hi, rs.gu.x, __builtin_fmaf(rs.gu.y < 0.0f ? dy_lo : dy_hi, rs.gu.y, rs.gu.z));
    const float fv = __builtin_fmaf(rs.gv.x < 0.0f ? dx_lo : dx_hi, rs.gv.x, __builtin_fmaf(rs.gv.y < 0.0f ? dy_lo : dy_hi, rs.gv.y, rs.gv.z));
    const float fw = __builtin_fmaf(rs.gw.x < 0.0f ? dx_lo : dx_hi, rs.gw.x, __builtin_fmaf(rs.gw.y < 0.0f ? dy_lo : dy_hi, rs.gw.y, rs.gw.z));
    return (fu < 0.0f) || (fv < 0.0f) || (fw < 0.0f);
}

// One lane per triangle, workgroups of 256. FILL == false: counts[t] += 1 for every tile t the triangle goes to.
// FILL == true: list[offsets[t] + counts[t]++] = triangle id (counts zeroed in between; `capacity`
// entries at most are written).
template <bool FILL>
__global__ void k_raster_bin(SceneView sv, DevRenderParams p, uint32_t n_tris_total, const float* __restrict__ tab,
                             uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                             uint32_t* __restrict__ list, uint32_t capacity, float4* __restrict__ recs) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = id < n_tris_total;
    const uint32_t g = p.grid, gs = 31u - (uint32_t)__builtin_clz(g);
    const int gw = (int)(p.width * g);
    RasterGrid gr;
    gr.aspect = (float)(p.width / p.height);
    gr.inv_ax = 1.0f / ((float)gw * 0.5f / gr.aspect);
    gr.inv_ay = 1.0f / ((float)(p.height * g) * 0.5f);
    RasterTri rt;
    if (FILL) {
        // the counting pass has left the triangle's half-planes and box
        raster_rec_head(recs + (size_t)RT_RREC * (valid ? id : 0u), rt.rs, rt.u0, rt.u1, rt.w0, rt.w1, rt.objword);
    } else {
        uint32_t unused = 0;
        rt = raster_prepare(sv, p, valid ? id : 0u, gw, (int)(p.height * g), gr.aspect, unused);
        if (valid) raster_rec_store(recs + (size_t)RT_RREC * id, rt);
    }
    const uint32_t ntx = ((uint32_t)gw + RT_RTILE - 1u) / RT_RTILE;
    const bool any = valid && rt.u0 <= rt.u1 && rt.w0 <= rt.w1;
    const uint32_t tx0 = any ? (uint32_t)rt.u0 / RT_RTILE : 1u, tx1 = any ? (uint32_t)rt.u1 / RT_RTILE : 0u;
    const uint32_t ty0 = any ? (uint32_t)rt.w0 / RT_RTILE : 1u, ty1 = any ? (uint32_t)rt.w1 / RT_RTILE : 0u;
    const uint32_t lane = threadIdx.x & 63u;
    // A triangle whose box spans many tiles (a large one, or one whose box is only bounded by its
    // leaf) is walked by the whole wave, one lane per tile, so that no lane holds the others up.
    const bool big = any && (tx1 - tx0 + 1u) * (ty1 - ty0 + 1u) > 8u;
#ifdef RT_DEBUG_RASTER
    if (valid && id == g_dbg[0])
        printf("bin<%d> tri %u: box u %d..%d w %d..%d any %d big %d objword %x\n  gu %a %a %a\n  gv %a %a %a\n  gw %a %a %a grow %g vis %d\n",
               (int)FILL, id, rt.u0, rt.u1, rt.w0, rt.w1, (int)any, (int)big, rt.objword, rt.rs.gu.x, rt.rs.gu.y, rt.rs.gu.z,
               rt.rs.gv.x, rt.rs.gv.y, rt.rs.gv.z, rt.rs.gw.x, rt.rs.gw.y, rt.rs.gw.z, rt.rs.grow, (int)rt.rs.visible);
#endif
    unsigned long long bigs = __ballot(big);
    while (bigs != 0ull) {
        const int l = __builtin_ctzll(bigs);
        bigs &= bigs - 1ull;
        RasterSetup rs;
        rs.gu = f3(bcast(rt.rs.gu.x, l), bcast(rt.rs.gu.y, l), bcast(rt.rs.gu.z, l));
        rs.gv = f3(bcast(rt.rs.gv.x, l), bcast(rt.rs.gv.y, l), bcast(rt.rs.gv.z, l));
        rs.gw = f3(bcast(rt.rs.gw.x, l), bcast(rt.rs.gw.y, l), bcast(rt.rs.gw.z, l));
        const int u0 = __builtin_amdgcn_readlane(rt.u0, l), u1 = __builtin_amdgcn_readlane(rt.u1, l);
        const int w0 = __builtin_amdgcn_readlane(rt.w0, l), w1 = __builtin_amdgcn_readlane(rt.w1, l);
        const uint32_t bid = (uint32_t)__builtin_amdgcn_readlane((int)id, l);
        const uint32_t bx0 = (uint32_t)u0 / RT_RTILE, nx = (uint32_t)u1 / RT_RTILE - bx0 + 1u;
        const uint32_t by0 = (uint32_t)w0 / RT_RTILE, ny = (uint32_t)w1 / RT_RTILE - by0 + 1u;
        for (uint32_t k = lane; k < nx * ny; k += 64u) {
            const uint32_t tyk = by0 + k / nx, txk = bx0 + k % nx;
            if (!raster_tile_owned(p, gs, txk, tyk)) continue;
            const int x0 = max(u0, (int)(txk * RT_RTILE)), x1 = min(u1, (int)(txk * RT_RTILE + RT_RTILE - 1u));
            const int y0 = max(w0, (int)(tyk * RT_RTILE)), y1 = min(w1, (int)(tyk * RT_RTILE + RT_RTILE - 1u));
#ifdef RT_DEBUG_RASTER
            if (bid == g_dbg[0] && txk == g_dbg[1] / RT_RTILE && tyk == g_dbg[2] / RT_RTILE)
                printf("bin<%d> big walk tri %u tile (%u,%u) rect %d..%d x %d..%d outside %d\n", (int)FILL, bid, txk, tyk, x0, x1, y0, y1,
                       (int)raster_rect_outside(rs, gr, x0, x1, y0, y1));
#endif
            if (raster_rect_outside(rs, gr, x0, x1, y0, y1)) continue;
            const uint32_t t = tyk * ntx + txk;
            const uint32_t pos = atomicAdd(&counts[t], 1u);
#ifdef RT_DEBUG_RASTER
            if (bid == g_dbg[0] && txk == g_dbg[1] / RT_RTILE && tyk == g_dbg[2] / RT_RTILE)
                printf("bin<%d> big walk tri %u tile %u pos %u offset %u capacity %u\n", (int)FILL, bid, t, pos, FILL ? offsets[t] : 0u, capacity);
#endif
            if (FILL && offsets[t] + pos < capacity) list[offsets[t] + pos] = bid;
        }
    }
    // The other lanes walk their own triangles' tiles. The 256 consecutive triangles of a workgroup go
    // to a few dozen tiles between them, so the group first gathers its entries per tile in a small hash
    // table in LDS and then touches each tile's global counter once: a tenth of the atomics, which
    // were what the pass spent most of its time on. An entry that finds no room in the table (a group
    // whose triangles are scattered over the frame) goes to the global counter directly.
    __shared__ uint32_t h_key[RT_BIN_HASH], h_cnt[RT_BIN_HASH], h_base[RT_BIN_HASH];
    for (uint32_t i = threadIdx.x; i < RT_BIN_HASH; i += blockDim.x) h_key[i] = 0xFFFFFFFFu, h_cnt[i] = 0u;
    __syncthreads();
    // Walks the lane's tiles; emit(tile) is called for every tile the triangle goes to.
    auto walk = [&](auto&& emit) {
        uint32_t tx = tx0, ty = ty0;
        for (bool walking = any && !big; walking;) {
            if (raster_tile_owned(p, gs, tx, ty)) {
                const int x0 = max(rt.u0, (int)(tx * RT_RTILE)), x1 = min(rt.u1, (int)(tx * RT_RTILE + RT_RTILE - 1u));
                const int y0 = max(rt.w0, (int)(ty * RT_RTILE)), y1 = min(rt.w1, (int)(ty * RT_RTILE + RT_RTILE - 1u));
                if (!raster_rect_outside(rt.rs, gr, x0, x1, y0, y1)) emit(ty * ntx + tx);
            }
            if (++tx > tx1) {
                tx = tx0;
                walking = ++ty <= ty1;
            }
        }
    };
    // The table slot of tile t, claimed if `insert`; RT_BIN_HASH when there is none.
    auto slot_of = [&](uint32_t t, bool insert) -> uint32_t {
        uint32_t h = (t * 0x9E3779B1u) >> (32 - RT_BIN_HASH_BITS);
        for (int probe = 0; probe < 16; ++probe) {
            uint32_t k = h_key[h];
            if (k == 0xFFFFFFFFu && insert) k = atomicCAS(&h_key[h], 0xFFFFFFFFu, t), k = k == 0xFFFFFFFFu ? t : k;
            if (k == t) return h;
            if (k == 0xFFFFFFFFu) return RT_BIN_HASH;   // not inserting, and the chain ends here
            h = (h + 1u) & (RT_BIN_HASH - 1u);
        }
        return RT_BIN_HASH;
    };
    // First walk: entries per tile of this group; entries without a slot are placed at once.
    walk([&](uint32_t t) {
        const uint32_t sl = slot_of(t, true);
        if (sl < RT_BIN_HASH) {
            atomicAdd(&h_cnt[sl], 1u);
        } else {
            const uint32_t pos = atomicAdd(&counts[t], 1u);
            if (FILL && offsets[t] + pos < capacity) list[offsets[t] + pos] = id;
        }
    });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < RT_BIN_HASH; i += blockDim.x) {
        const uint32_t t = h_key[i];
        if (t == 0xFFFFFFFFu) continue;
        const uint32_t before = atomicAdd(&counts[t], h_cnt[i]);
        if (FILL) h_base[i] = offsets[t] + before, h_cnt[i] = 0u;
    }
    if (!FILL) return;
    __syncthreads();
    // Second walk: every entry takes its place in the stretch of the list the group has reserved.
    walk([&](uint32_t t) {
        const uint32_t sl = slot_of(t, false);
        if (sl == RT_BIN_HASH) return;   // placed by the first walk
        const uint32_t pos = h_base[sl] + atomicAdd(&h_cnt[sl], 1u);
        if (pos < capacity) list[pos] = id;
    });
}

// offsets[i] = counts[0] + ... + counts[i - 1] for i = 0 .. n (offsets[n] = the total); counts are
// set to zero. One workgroup of 1024 threads, 32 consecutive counters a thread and stretch (the tiles
// of a 3840 x 2160 x 4 frame are one stretch: the kernel's time is the latency of a stretch's loads,
// two barriers and stores, times the stretches).
#define RT_SCAN_PER 32u
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ offsets) {
    __shared__ uint32_t s_wave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t carry = 0;   // sum of everything before the current stretch
    for (uint32_t first = 0; first < n; first += 1024u * RT_SCAN_PER) {
        const uint32_t i = first + RT_SCAN_PER * t;
        uint32_t c[RT_SCAN_PER];
#pragma unroll
        for (uint32_t q = 0; q < RT_SCAN_PER; q += 4u) {
            if (i + q + 3u < n) {
                uint4* p4 = reinterpret_cast<uint4*>(counts + i + q);   // i + q is a multiple of four
                const uint4 v = *p4;
                c[q] = v.x, c[q + 1u] = v.y, c[q + 2u] = v.z, c[q + 3u] = v.w;
                *p4 = make_uint4(0u, 0u, 0u, 0u);
            } else {
#pragma unroll
                for (uint32_t k = q; k < q + 4u; ++k) {
                    c[k] = i + k < n ? counts[i + k] : 0u;
                    if (i + k < n) counts[i + k] = 0u;
                }
            }
        }
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < RT_SCAN_PER; ++k) mine += c[k];
        uint32_t incl = mine;   // inclusive sum over the wave's lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = carry, total = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t v = s_wave[w];
            if (w < wave) before += v;
            total += v;
        }
        uint32_t run = before + incl - mine;
#pragma unroll
        for (uint32_t q = 0; q < RT_SCAN_PER; q += 4u) {
            uint32_t r[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) r[k] = run, run += c[q + k];
            if (i + q + 3u < n) {
                *reinterpret_cast<uint4*>(offsets + i + q) = make_uint4(r[0], r[1], r[2], r[3]);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k)
                    if (i + q + k < n) offsets[i + q + k] = r[k];
            }
        }
        carry += total;
        __syncthreads();
    }
    if (t == 0) offsets[n] = carry;
}

// Waves that share a tile (each takes every other stretch of the tile's list). Two waves a tile
// raise the occupancy from 3 to 5 waves a SIMD and measured the same frame time; one is simpler.
#ifndef RT_RTILE_GROUP
#define RT_RTILE_GROUP 1
#endif
#ifndef RT_SEG
#define RT_SEG 4u    // rows of a triangle's box that one lane goes down
#endif
#ifndef RT_RUN
#define RT_RUN 8u   // longest run of samples queued as one entry (a longer span of a row is queued in pieces; 16 is
                    // 0.02 ms faster by itself and costs the LDS that an eleventh group per CU needs: 0.05 ms)
#endif
struct RasterTileLds {
    unsigned long long z[RT_RTILE * RT_RTILE];   // nearest (t, id) per sample of the tile, row by row
    float tri[RT_RTILE_GROUP][RT_RTILE_TRIS][12];    // per wave: e1 xyz, e2 xyz, v0 xyz, id, object word, a word of tie bits
    uint32_t queue[RT_RTILE_GROUP][RT_RASTER_QUEUE]; // per wave: runs of candidate samples: first sample | length << 10 | slot << 16
    uint16_t samples[RT_RTILE_GROUP][64 * RT_RUN];   // per wave: the samples of up to 64 runs, packed: sample | slot << 10
    uint16_t segs[RT_RTILE_GROUP][64 * (RT_RTILE / RT_SEG)];   // per wave: row segments of the current 64 triangles: lane | segment << 6
    float dx[RT_RTILE], dy[RT_RTILE];            // image-plane coordinates of the tile's columns and rows
    // (Their reciprocals, as Ray.inv_dir holds them (src/main.rs:329), are wanted by the record loop only: they
    // go where the sample codes were, see recip_x. With that and runs of eight the structure is 14,080 bytes,
    // eleven of the 1,280-byte pieces LDS is handed out in, and a CU holds eleven groups instead of ten.)
    __device__ __forceinline__ float* recip_x() { return reinterpret_cast<float*>(&samples[0][0]); }
    __device__ __forceinline__ float* recip_y() { return recip_x() + RT_RTILE; }
    // Tie bits, one per sample: two accepted triangles within RT_TIE_ULPS of each other in t. They live in
    // the twelfth word of the triangle slots (an array of their own took the structure past a multiple
    // of the 1,280 bytes LDS is handed out in, and the kernel from ten groups a CU to nine: 0.07 ms a frame).
    __device__ __forceinline__ uint32_t* tie_word(uint32_t w) {
        return reinterpret_cast<uint32_t*>(&tri[w / RT_RTILE_TRIS][w % RT_RTILE_TRIS][11]);
    }
};
static_assert(RT_RTILE * RT_RTILE / 32 <= RT_RTILE_GROUP * RT_RTILE_TRIS, "a tie word per triangle slot covers the tile's samples");
// How near (in units of the last place of t) two accepted triangles of one sample have to lie for the
// sample to be marked: only then can the order of the reference's search decide between them (see the
// record loop of k_raster_tile).
#define RT_TIE_ULPS 1024u
// Orders a wave's own LDS writes before its later LDS reads (the hardware serves one wave's LDS
// instructions in order; this keeps the compiler from moving them).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Position (subx, suby) of sub-sample s on the n x n grid inside a pixel, n = 1 << shift: the inverse
// of sample_of.
__device__ __forceinline__ void sample_pos(uint32_t grid, uint32_t shift, uint32_t s, uint32_t* subx, uint32_t* suby) {
    *subx = s & (grid - 1u);
    *suby = grid == 2u ? ((s == 1u || s == 2u) ? 1u : 0u) : s >> shift;
}

// One half-plane along one row of the tile, F(column) = fma(dx[column], gx, r): the columns that
// pass (!(F < 0)) are those at or beyond the boundary x = (aspect - r / gx) * ax - 0.5 for gx > 0, at
// or before it for gx < 0. fma is correctly rounded, so the sign of F is the sign of the exact
// dx * gx + r; the table's dx differs from (column + 0.5) / ax - aspect by a few units of its last
// place (under 0.008 column for the widest frames the projection takes, 65535 sample columns), and
// `x`, computed with a rounded reciprocal, from the exact boundary by as much again unless it
// lies far outside the tile. With RT_CLIP_SLACK = 1/16 column on either side the cut keeps every
// passing column. An x that is NaN (gx == 0 with r == 0) cuts nothing.
#ifndef RT_CLIP_SLACK
#define RT_CLIP_SLACK 0.0625f
#endif
// The run [lo, hi] of columns is kept in floats and cut without a branch: a half-plane with gx > 0
// raises lo to ceil(x - slack), one with gx < 0 lowers hi to floor(x + slack). `raise` is 0 for gx > 0
// and -inf otherwise, `lower` 0 for gx < 0 and +inf otherwise (set once per triangle: see row_pens),
// so the side that does not apply becomes -inf / +inf, or NaN, and max / min pass it over.
__device__ __forceinline__ void row_clip(float x, float raise, float lower, float& lo, float& hi) {
    lo = fmaxf(lo, ceilf(x - RT_CLIP_SLACK) + raise);
    hi = fminf(hi, floorf(x + RT_CLIP_SLACK) + lower);
}
__device__ __forceinline__ void row_pens(float gx, float& raise, float& lower) {
    raise = gx > 0.0f ? 0.0f : -RT_INF;
    lower = gx < 0.0f ? 0.0f : RT_INF;
}

#ifndef RT_RECORD_ROUND
#define RT_RECORD_ROUND 4   // samples a lane writes the records of at a time
#endif
#ifndef RT_RTILE_WAVES
#define RT_RTILE_WAVES 3
#endif
template <bool STATS>
__global__ __launch_bounds__(64 * RT_RTILE_GROUP, RT_RTILE_WAVES) void k_raster_tile(SceneView sv, DevRenderParams p, LightProj lp,
                                                       const uint32_t* __restrict__ tile_list,
                                                       const float* __restrict__ tab,
                                                       const uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ list,
                                                       const float4* __restrict__ recs,
                                                       float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                       uint32_t* __restrict__ keys, DevStats* __restrict__ stats) {
    __shared__ RasterTileLds lds;
    LaneStats st;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t g = p.grid, gs = 31u - (uint32_t)__builtin_clz(g), spp = p.spp;
    const float aspect = (float)(p.width / p.height);
    const int gw = (int)(p.width * g), gh = (int)(p.height * g);
    const uint32_t L = (uint32_t)(gw + gh);
    // Which raster tile: sub-tile (a, b) of g x g in pixel tile `launch_tile` of this launch.
    const uint32_t launch_tile = blockIdx.x / spp, sub = blockIdx.x % spp;
    const uint32_t a = sub % g, b = sub / g;
    const uint32_t local_tile = p.tile_first + launch_tile;
    const uint32_t tile = tile_list ? tile_list[local_tile] : local_tile;
    const uint32_t ptx = tile % p.tiles_x, pty = tile / p.tiles_x;
    const uint32_t rtx = ptx * g + a, rty = pty * g + b;
    const int su0 = (int)(rtx * RT_RTILE), sw0 = (int)(rty * RT_RTILE);
    const uint32_t P = RT_RTILE >> gs;                       // pixels a side of this raster tile
    // Morton number of the raster tile's first pixel inside its pixel tile.
    uint32_t m0 = 0;
    {
        const uint32_t px0 = a * P, py0 = b * P;
#pragma unroll
        for (int k = 0; k < 5; ++k) m0 |= (((px0 >> k) & 1u) << (2 * k)) | (((py0 >> k) & 1u) << (2 * k + 1));
    }
    const uint32_t slot_base = launch_tile * (RT_TILE * RT_TILE * spp) + m0 * spp;
    const uint32_t ntx = ((uint32_t)gw + RT_RTILE - 1u) / RT_RTILE;
    const bool on_frame = su0 < gw && sw0 < gh;

    // The tile's list, and the first stretch of it: the loads are issued before the tile's words in LDS
    // are set up, and every later stretch is fetched while the one before it is worked on.
    const uint32_t rt_index = rty * ntx + rtx;
    const uint32_t first = on_frame ? offsets[rt_index] : 0u, n_list = on_frame ? offsets[rt_index + 1] - first : 0u;
    struct Stretch {
        bool valid;
        uint32_t id;
        float4 r0, r1, r2, r3, r4, r5;
    };
    auto fetch = [&](uint32_t base) {
        Stretch c;
        c.valid = lane < RT_RTILE_TRIS && base + lane < n_list;
        c.id = 0;
        c.r0 = c.r1 = c.r2 = c.r3 = c.r4 = c.r5 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c.valid) {
            c.id = list[first + base + lane];
            const float4* r = recs + (size_t)RT_RREC * c.id;
            c.r0 = r[0], c.r1 = r[1], c.r2 = r[2], c.r3 = r[3], c.r4 = r[4], c.r5 = r[5];
        }
        return c;
    };
    Stretch cur = fetch(wv * RT_RTILE_TRIS);
#ifdef RT_DEBUG_RASTER
    const bool dbg_tile = g_dbg[0] != 0xFFFFFFFFu && rtx == g_dbg[1] / RT_RTILE && rty == g_dbg[2] / RT_RTILE;
    const uint32_t dbg_zi = (g_dbg[2] % RT_RTILE) * RT_RTILE + g_dbg[1] % RT_RTILE;
    if (dbg_tile && lane == 0) printf("tile (%u,%u) index %u: first %u n_list %u su0 %d sw0 %d\n", rtx, rty, rt_index, first, n_list, su0, sw0);
#endif

    for (uint32_t i = threadIdx.x; i < RT_RTILE * RT_RTILE; i += 64u * RT_RTILE_GROUP) lds.z[i] = RT_ZKEY_NONE;
    for (uint32_t i = threadIdx.x; i < RT_RTILE * RT_RTILE / 32u; i += 64u * RT_RTILE_GROUP) *lds.tie_word(i) = 0u;
    if (threadIdx.x < RT_RTILE) {
        const float c = tab[min(su0 + (int)threadIdx.x, gw - 1)];
        lds.dx[threadIdx.x] = c;
    } else if (threadIdx.x < 2u * RT_RTILE) {
        const float c = tab[gw + min(sw0 + (int)(threadIdx.x - RT_RTILE), gh - 1)];
        lds.dy[threadIdx.x - RT_RTILE] = c;
    }
    __syncthreads();

    uint32_t q_head = 0, q_tail = 0;   // wave-uniform: runs wait at queue[(q_head .. q_tail) % RT_RASTER_QUEUE]
    const rt_f3 o = f3(0.0f, 0.0f, 0.0f);
    // Puts the samples of the first n (<= 64) waiting runs through the exact test: the reference's
    // arithmetic on the reference's ray, acceptance rule !failed && t > 0 (src/main.rs:117), then the
    // object's box test; the nearest (t, id) of a sample is kept with an atomic minimum in LDS. The
    // runs are short (a triangle's row span averages under four samples), so they are first laid out
    // sample by sample (a prefix sum of the runs' lengths over the lanes says where each run's samples
    // go) and the test then takes 64 samples at a time whatever runs they came from: full waves in the
    // division-carrying part instead of lanes reserved for the longest possible run.
    auto test_runs = [&](uint32_t n) {
        uint32_t e = 0, len = 0;
        if (lane < n) {
            e = lds.queue[wv][(q_head + lane) % RT_RASTER_QUEUE];
            len = (e >> 10) & 63u;
        }
        q_head += n;
        uint32_t incl = len;   // inclusive sum of the lengths over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t at = incl - len, code0 = (e & 1023u) | ((e >> 16) << 10);
#pragma unroll
        for (uint32_t j = 0; j < RT_RUN; ++j) {
            if (j >= 4u && __ballot(j < len) == 0ull) break;   // most runs are short
            if (j < len) lds.samples[wv][at + j] = (uint16_t)(code0 + j);
        }
        wave_lds_sync();
        for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
            const uint32_t g = g0 + lane;
            if (g >= total) continue;
            const uint32_t code = lds.samples[wv][g];
            const uint32_t zi = code & 1023u;
            const float* td = lds.tri[wv][code >> 10];
            const rt_f3 e1 = f3(td[0], td[1], td[2]), e2 = f3(td[3], td[4], td[5]), v0 = f3(td[6], td[7], td[8]);
            const uint32_t id = __float_as_uint(td[9]), ow = __float_as_uint(td[10]);
            const rt_f3 d = f3(lds.dx[zi & (RT_RTILE - 1u)], lds.dy[zi / RT_RTILE], 1.0f);
            if (STATS) st.tri += 1;
            const rt_tri_eval ev = tri_intersect(e1, e2, v0, o, d);
#ifdef RT_DEBUG_RASTER
            if (dbg_tile && id == g_dbg[0] && zi == dbg_zi)
                printf("tile test tri %u zi %u slot %u: t %a u %a v %a failed %d d %a %a ow %x\n", id, zi, code >> 10, ev.t, ev.u, ev.v, (int)ev.failed, d.x, d.y, ow);
#endif
            if (ev.failed || !(ev.t > 0.0f) || !(ev.t < RT_INF)) continue;
            if (!(ow & 0x80000000u)) {
                if (STATS) st.box += 1, st.bytes += 16;
                if (!raster_object_box(sv, tab, L, gw, ow, (uint32_t)su0 + (zi & (RT_RTILE - 1u)), (uint32_t)sw0 + zi / RT_RTILE)) continue;
            }
            // The minimum so far comes back from the exchange: whenever the new t and that one lie within
            // RT_TIE_ULPS of each other the sample is marked. The minima of a sample only fall, so every accepted
            // triangle that near the final minimum meets it, or a value between the two, in one of these
            // exchanges ("none" has the largest bits of all and is near nothing; a triangle that meets its own
            // earlier entry, from a run tested twice, is no tie).
            const uint32_t tb = __float_as_uint(ev.t);
            const unsigned long long key = ((unsigned long long)tb << 32) | id;
            const unsigned long long was = atomicMin(&lds.z[zi], key);
            uint32_t apart;   // |was.t - t| in units of the last place
            asm("v_sad_u32 %0, %1, %2, 0" : "=v"(apart) : "v"((uint32_t)(was >> 32)), "v"(tb));
            if (apart <= RT_TIE_ULPS && was != key) atomicOr(lds.tie_word(zi >> 5), 1u << (zi & 31u));
        }
        wave_lds_sync();   // the next call lays its samples out over these
    };

    // Column of image-plane x: dx = (column + 0.5) / ax - aspect up to rounding, so a half-plane's
    // boundary along a row, fma(dx, gx, r) = 0, lies near column (aspect - r / gx) * ax - 0.5.
    const float ax = (float)gw * 0.5f / aspect;
    for (uint32_t base = wv * RT_RTILE_TRIS; base < n_list; base += RT_RTILE_TRIS * RT_RTILE_GROUP) {
        // One lane sets one triangle of the list up; its box is cut to the tile (tile-local from here on).
        int c0 = 0, c1 = -1, r0 = 0, r1 = -1;
        RasterSetup rs;
        rs.gu = rs.gv = rs.gw = f3(0.0f, 0.0f, 0.0f);
        if (cur.valid) {
            if (STATS) st.bytes += 4 + 16 * RT_RREC;
            const float4 a = cur.r0, b = cur.r1, c = cur.r2;
            rs.gu = f3(a.x, a.y, a.z), rs.gv = f3(a.w, b.x, b.y), rs.gw = f3(b.z, b.w, c.x);
            const uint32_t bu = __float_as_uint(c.y), bw = __float_as_uint(c.z);
            const int u0 = (int)(bu & 0xFFFFu), u1 = (int)(bu >> 16), w0 = (int)(bw & 0xFFFFu), w1 = (int)(bw >> 16);
            c0 = max(u0 - su0, 0), c1 = min(u1 - su0, (int)RT_RTILE - 1);
            r0 = max(w0 - sw0, 0), r1 = min(w1 - sw0, (int)RT_RTILE - 1);
            if (c0 > c1) r1 = r0 - 1;
            float* td = lds.tri[wv][lane];
            td[0] = cur.r3.x, td[1] = cur.r3.y, td[2] = cur.r3.z;
            td[3] = cur.r3.w, td[4] = cur.r4.x, td[5] = cur.r4.y;
            td[6] = cur.r4.z, td[7] = cur.r4.w, td[8] = cur.r5.x;
            td[9] = __uint_as_float(cur.id);
            td[10] = c.w;
#ifdef RT_DEBUG_RASTER
            if (dbg_tile && cur.id == g_dbg[0])
                printf("tile setup tri %u base %u lane %u: box u %d..%d w %d..%d -> c %d..%d r %d..%d\n", cur.id, base, lane, u0, u1, w0, w1, c0, c1, r0, r1);
#endif
        }
        if (base + RT_RTILE_TRIS * RT_RTILE_GROUP < n_list) cur = fetch(base + RT_RTILE_TRIS * RT_RTILE_GROUP);
        // The triangles' boxes differ in height (ten rows on average, up to the tile's 32), so a wave in
        // which every lane went down its own triangle's rows would run as long as its tallest triangle
        // with a third of its lanes at work. The rows are dealt out in segments of RT_SEG instead: a
        // prefix sum of the triangles' segment counts says which lane takes which segment, and the lane
        // fetches that triangle's half-planes and box from the lane that set it up.
        const uint32_t h = r1 >= r0 ? (uint32_t)(r1 - r0 + 1) : 0u;
        const uint32_t nseg = (h + RT_SEG - 1u) / RT_SEG;
        uint32_t seg_incl = nseg;
#pragma unroll
        for (int off2 = 1; off2 < 64; off2 <<= 1) {
            const uint32_t v = __shfl_up(seg_incl, off2, 64);
            if ((int)lane >= off2) seg_incl += v;
        }
        const uint32_t n_segs = (uint32_t)__builtin_amdgcn_readlane((int)seg_incl, 63);
        {
            const uint32_t at = seg_incl - nseg;
#pragma unroll
            for (uint32_t k = 0; k < RT_RTILE / RT_SEG; ++k)
                if (k < nseg) lds.segs[wv][at + k] = (uint16_t)(lane | (k << 6));
        }
        const uint32_t box_word = (uint32_t)c0 | ((uint32_t)max(c1, 0) << 5) | ((uint32_t)r0 << 10) | ((uint32_t)max(r1, 0) << 15);
        wave_lds_sync();
        // slope and offset of the boundary column as a function of the row's r = fma(dy, gy, gz)
        const float off = (aspect * ax - 0.5f) - (float)su0;
        for (uint32_t g0 = 0; g0 < n_segs; g0 += 64u) {
            const bool has_seg = g0 + lane < n_segs;
            const uint32_t code = has_seg ? lds.segs[wv][g0 + lane] : 0u;
            const int t = (int)(code & 63u);          // the lane that holds the triangle, and its slot in lds.tri
            const uint32_t sg = code >> 6;
            RasterSetup q;
            q.gu = f3(__shfl(rs.gu.x, t, 64), __shfl(rs.gu.y, t, 64), __shfl(rs.gu.z, t, 64));
            q.gv = f3(__shfl(rs.gv.x, t, 64), __shfl(rs.gv.y, t, 64), __shfl(rs.gv.z, t, 64));
            q.gw = f3(__shfl(rs.gw.x, t, 64), __shfl(rs.gw.y, t, 64), __shfl(rs.gw.z, t, 64));
            const uint32_t bw2 = (uint32_t)__shfl((int)box_word, t, 64);
            const int sc0 = (int)(bw2 & 31u), sc1 = (int)((bw2 >> 5) & 31u);
            const int sr0 = (int)((bw2 >> 10) & 31u) + (int)(sg * RT_SEG);
            const int sr1 = has_seg ? min((int)((bw2 >> 15) & 31u), sr0 + (int)RT_SEG - 1) : sr0 - 1;
            const float ku = -ax * __builtin_amdgcn_rcpf(q.gu.x), kv = -ax * __builtin_amdgcn_rcpf(q.gv.x),
                        kw = -ax * __builtin_amdgcn_rcpf(q.gw.x);
            float raise_u, lower_u, raise_v, lower_v, raise_w, lower_w;
            row_pens(q.gu.x, raise_u, lower_u), row_pens(q.gv.x, raise_v, lower_v), row_pens(q.gw.x, raise_w, lower_w);
            // Every lane goes down the rows of its segment: per row, the run of columns that pass all
            // three half-planes (an exact run: see row_clip), queued in pieces of RT_RUN samples.
            for (int row = sr0; __ballot(row <= sr1) != 0ull; ++row) {
                int lo = sc0, hi = row <= sr1 ? sc1 : sc0 - 1;
                if (lo <= hi) {
                    if (STATS) st.node += 1;
                    const float dy = lds.dy[row];
                    const float ru = __builtin_fmaf(dy, q.gu.y, q.gu.z), rv = __builtin_fmaf(dy, q.gv.y, q.gv.z),
                                rw = __builtin_fmaf(dy, q.gw.y, q.gw.z);
                    float flo = (float)lo, fhi = (float)hi;
                    row_clip(__builtin_fmaf(ru, ku, off), raise_u, lower_u, flo, fhi);
                    row_clip(__builtin_fmaf(rv, kv, off), raise_v, lower_v, flo, fhi);
                    row_clip(__builtin_fmaf(rw, kw, off), raise_w, lower_w, flo, fhi);
#ifdef RT_DEBUG_RASTER
                    if (dbg_tile && __float_as_uint(lds.tri[wv][t][9]) == g_dbg[0] && (uint32_t)row == dbg_zi / RT_RTILE)
                        printf("tile row %d tri slot %d seg %u (sr %d..%d sc %d..%d): xu %g xv %g xw %g -> flo %g fhi %g; ru %a rv %a rw %a gx %a %a %a\n", row, t, sg, sr0, sr1,
                               sc0, sc1, __builtin_fmaf(ru, ku, off), __builtin_fmaf(rv, kv, off), __builtin_fmaf(rw, kw, off), flo, fhi, ru, rv, rw, q.gu.x, q.gv.x, q.gw.x);
#endif
                    // a half-plane that does not depend on the column passes the whole row or none of it
                    const bool none = (q.gu.x == 0.0f && ru < 0.0f) || (q.gv.x == 0.0f && rv < 0.0f) || (q.gw.x == 0.0f && rw < 0.0f);
                    // lo and hi stay within the box's columns whenever the run is not empty
                    if (flo <= fhi && !none) lo = (int)flo, hi = (int)fhi;
                    else hi = lo - 1;
                }
                while (true) {
                    const bool has = lo <= hi;
                    const unsigned long long pm = __ballot(has);
                    if (pm == 0ull) break;
                    if (has) {
                        const uint32_t len = min((uint32_t)(hi - lo + 1), RT_RUN);
                        const uint32_t pos = q_tail + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull));
                        lds.queue[wv][pos % RT_RASTER_QUEUE] = ((uint32_t)row * RT_RTILE + (uint32_t)lo) | (len << 10) | ((uint32_t)t << 16);
#ifdef RT_DEBUG_RASTER
                        if (dbg_tile && __float_as_uint(lds.tri[wv][t][9]) == g_dbg[0] && (uint32_t)row == dbg_zi / RT_RTILE)
                            printf("tile queue row %d lo %d len %u slot %d pos %u (head %u tail %u)\n", row, lo, len, t, pos, q_head, q_tail);
#endif
                        lo += (int)RT_RUN;
                    }
                    q_tail += (uint32_t)__builtin_popcountll(pm);
                    if (q_tail - q_head >= 64u) {
                        wave_lds_sync();
                        while (q_tail - q_head >= 64u) test_runs(64u);
                        wave_lds_sync();
                    }
                }
            }
        }
        // The triangle slots are reused by the next 64 triangles: test what still waits.
        wave_lds_sync();
        while (q_tail != q_head) test_runs(min(q_tail - q_head, 64u));
        wave_lds_sync();
    }

    __syncthreads();   // both waves' nearest hits are in
    static_assert(sizeof(lds.samples) >= 2 * RT_RTILE * sizeof(float), "the reciprocals take the sample codes' place");
    if (threadIdx.x < RT_RTILE) lds.recip_x()[threadIdx.x] = 1.0f / lds.dx[threadIdx.x];
    else if (threadIdx.x < 2u * RT_RTILE) lds.recip_y()[threadIdx.x - RT_RTILE] = 1.0f / lds.dy[threadIdx.x - RT_RTILE];
    __syncthreads();
#ifdef RT_DEBUG_RASTER
    if (dbg_tile && lane == 0) printf("tile final z[%u] = %llx\n", dbg_zi, lds.z[dbg_zi]);
#endif
    // The records of the tile's samples, in slot order: what k_wf_primary stores.
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    // Four samples a lane at a time: their nearest hits are read and their triangles' normals asked for
    // before the first record is worked out, so that a lane waits for device memory once in four samples
    // (the normal's address comes out of the nearest-hit word: two dependent reads per sample).
    constexpr uint32_t U = RT_RECORD_ROUND;
    static_assert(RT_RTILE * RT_RTILE % (64u * RT_RTILE_GROUP * U) == 0, "the tile's samples are dealt out in rounds of U per lane");
    // A sample's nearest hit over again, the way the reference finds it (src/main.rs:93-153): the triangles
    // that the exact test accepts are gone through in the order of their ids, the nearest so far in
    // `closest`; a leaf's triangles count only if its box passes the test with the ray cut at the `closest`
    // of the moment its packet of eight leaves began (the last n_leaves % 8 leaves of an object: the
    // object-box form of the test, which knows no length), and a triangle replaces the nearest only if
    // strictly nearer. For the rare sample whose nearest hit does not pass its leaf's box with the ray cut at
    // the hit itself: then, and only then, the outcome can depend on that order (when the box does pass,
    // it passes for every earlier `closest` too, all of which are at least as large, and the nearest hit
    // is the reference's). The tile's list holds every triangle the exact test can accept at any of the
    // tile's samples, in no particular order, so the next id is looked for anew each time: a few thousand
    // triangle tests for one sample in many millions.
    auto rescan = [&](uint32_t sl, uint32_t wl) -> unsigned long long {
        const rt_f3 d = f3(lds.dx[sl], lds.dy[wl], 1.0f), inv = f3(lds.recip_x()[sl], lds.recip_y()[wl], 1.0f);
        unsigned long long best = RT_ZKEY_NONE;
        float closest = RT_INF, packet_closest = RT_INF;
        uint32_t last = 0, packet = 0xFFFFFFFFu, packet_obj = 0xFFFFFFFFu;
        bool any_before = false;
        for (;;) {
            uint32_t nid = 0xFFFFFFFFu, now = 0;
            float nt = 0.0f;
            for (uint32_t i = 0; i < n_list; ++i) {
                const uint32_t id = list[first + i];
                if ((any_before && id <= last) || id >= nid) continue;
                const float4* r = recs + (size_t)RT_RREC * id;
                const float4 r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
                const uint32_t ow = __float_as_uint(r2.w);
                const rt_tri_eval ev = tri_intersect(f3(r3.x, r3.y, r3.z), f3(r3.w, r4.x, r4.y), f3(r4.z, r4.w, r5.x), o, d);
                if (ev.failed || !(ev.t > 0.0f) || !(ev.t < RT_INF)) continue;
                if (!(ow & 0x80000000u) && !raster_object_box(sv, tab, L, gw, ow, (uint32_t)su0 + sl, (uint32_t)sw0 + wl)) continue;
                nid = id, nt = ev.t, now = ow & 0x7FFFFFFFu;
            }
            if (nid == 0xFFFFFFFFu) break;
            any_before = true, last = nid;
            const DevObject& ob = sv.objs[now];
            const uint32_t leaf = ob.leaf_base + ((nid - ob.tri_base) >> 3);
            // a packet is eight consecutive leaves of one object, counted from the object's first
            const uint32_t pk = (nid - ob.tri_base) >> 6;
            if (pk != packet || now != packet_obj) packet = pk, packet_obj = now, packet_closest = closest;
            const float4* lg = sv.leaf_gate + 2 * (size_t)leaf;
            if (!ref_leaf_gate(lg[0], lg[1], o, inv, packet_closest)) continue;
            if (closest > nt) {
                closest = nt;
                best = ((unsigned long long)__float_as_uint(nt) << 32) | nid;
            }
        }
        return best;
    };
    for (uint32_t k0 = threadIdx.x; k0 < RT_RTILE * RT_RTILE; k0 += 64u * RT_RTILE_GROUP * U) {
        uint32_t sul[U], swl[U];
        bool inside[U], hit_any[U];
        HitD hs[U];
        rt_f3 normal[U];
        float4 leaf_lo[U], leaf_hi[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t k = k0 + u * 64u * RT_RTILE_GROUP;
            const uint32_t pl = k >> (2u * gs), s = k & (spp - 1u);   // pixel of the raster tile (Morton number), sample: spp = g * g, g a power of two
            uint32_t pxl = 0, pyl = 0;
#pragma unroll
            for (int bit = 0; bit < 5; ++bit) {
                pxl |= ((pl >> (2 * bit)) & 1u) << bit;
                pyl |= ((pl >> (2 * bit + 1)) & 1u) << bit;
            }
            uint32_t subx, suby;
            sample_pos(g, gs, s, &subx, &suby);
            sul[u] = (pxl << gs) + subx, swl[u] = (pyl << gs) + suby;   // sample inside the tile
            const uint32_t x = ((uint32_t)su0 + sul[u]) >> gs, y = ((uint32_t)sw0 + swl[u]) >> gs;
            inside[u] = x < p.width && y < p.height;
            const unsigned long long key = inside[u] ? lds.z[swl[u] * RT_RTILE + sul[u]] : RT_ZKEY_NONE;
            hit_any[u] = key != RT_ZKEY_NONE;
            hs[u].tri = hit_any[u] ? (uint32_t)key : 0u;
            hs[u].t = hit_any[u] ? __uint_as_float((uint32_t)(key >> 32)) : 0.0f;
            hs[u].u = hs[u].v = 0.0f;
            hs[u].obj = 0;
            while (hs[u].obj + 1 < sv.n_objs && hs[u].tri >= sv.objs[hs[u].obj].tri_base + sv.objs[hs[u].obj].n_tris) ++hs[u].obj;
            normal[u] = ld3(sv.tri_data + 6 * (size_t)hs[u].tri);
            // no object of a projected scene has trailing triangles: leaf l of the scene holds triangles 8 l .. 8 l + 7
            const float4* lg = sv.leaf_gate + 2 * (size_t)(hs[u].tri >> 3);
            leaf_lo[u] = lg[0], leaf_hi[u] = lg[1];
        }
        // The reference tests a triangle only for rays that pass the box of its leaf (ref_leaf_gate): the
        // nearest hit is checked against it, with the ray cut at the hit, and the few samples whose nearest
        // hit fails are worked out again in the reference's order.
        // (A projected scene's eight-wide triangles fill whole leaves from triangle 0 on: objects with
        // trailing triangles come last, see rt_scene_build.)
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const rt_f3 inv = f3(lds.recip_x()[sul[u]], lds.recip_y()[swl[u]], 1.0f);
            if (hit_any[u] && !ref_leaf_gate(leaf_lo[u], leaf_hi[u], o, inv, hs[u].t)) {
                // A hit in the very plane of its leaf's box (an axis-aligned wall) fails this by rounding as
                // often as not. The order matters only if another accepted triangle lies between the hit
                // and the box's near plane: none does when the box passes with the ray cut RT_TIE_ULPS
                // behind the hit and the sample is not marked.
                const uint32_t zi = swl[u] * RT_RTILE + sul[u];
                const float t_far = __uint_as_float(__float_as_uint(hs[u].t) + RT_TIE_ULPS);
                if (!((*lds.tie_word(zi >> 5) >> (zi & 31u)) & 1u) && ref_leaf_gate(leaf_lo[u], leaf_hi[u], o, inv, t_far)) continue;
                const unsigned long long key = rescan(sul[u], swl[u]);
                hit_any[u] = key != RT_ZKEY_NONE;
                hs[u].tri = hit_any[u] ? (uint32_t)key : 0u;
                hs[u].t = hit_any[u] ? __uint_as_float((uint32_t)(key >> 32)) : 0.0f;
                hs[u].obj = 0;
                while (hs[u].obj + 1 < sv.n_objs && hs[u].tri >= sv.objs[hs[u].obj].tri_base + sv.objs[hs[u].obj].n_tris) ++hs[u].obj;
                normal[u] = ld3(sv.tri_data + 6 * (size_t)hs[u].tri);
            }
        }
        // The objects' trailing n % 8 triangles (src/main.rs:157-169), which the reference tests one by one
        // after an object's eight-wide ones and which never lower `closest`: the last of them nearer
        // than the nearest eight-wide hit wins. Projection takes scenes in which no eight-wide triangle
        // follows a trailing one in object order, so `closest` is the nearest eight-wide hit of the whole
        // scene for every one of them, and "the last" is found by going through them backwards.
        if (sv.n_rest_total != 0u) {
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                if (!inside[u]) continue;
                const rt_f3 d = f3(lds.dx[sul[u]], lds.dy[swl[u]], 1.0f);
                const float closest = hit_any[u] ? hs[u].t : RT_INF;
                bool found = false;
                for (uint32_t oi = sv.n_objs; oi-- > 0u && !found;) {
                    const DevObject& ob = sv.objs[oi];
                    if (ob.n_rest == 0u) continue;
                    if (!raster_object_box(sv, tab, L, gw, oi, (uint32_t)su0 + sul[u], (uint32_t)sw0 + swl[u])) continue;
                    for (uint32_t r = ob.n_rest; r-- > 0u && !found;) {
                        const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
                        float t, bu, bv;
                        if (STATS) st.tri += 1;
                        if (triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, closest, &t, &bu, &bv)) {
                            found = true;
                            hit_any[u] = true;
                            hs[u].tri = ob.tri_base + ob.n_tris - ob.n_rest + r;
                            hs[u].obj = oi;
                            hs[u].t = t;
                            normal[u] = ld3(sv.tri_data + 6 * (size_t)hs[u].tri);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t k = k0 + u * 64u * RT_RTILE_GROUP;
            const rt_f3 d = f3(lds.dx[sul[u]], lds.dy[swl[u]], 1.0f);
            const rt_f3 pos = (o + d * hs[u].t) + (normal[u] * 0.00001f);   // as surface_of has it
            const rt_f3 sun_dir = g_normalize(sun - pos);
            const bool lit = hit_any[u] && g_dot(normal[u], sun_dir) > 0.0f;
            const uint32_t slot = slot_base + k;
            samp[slot] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(hit_any[u] ? hs[u].tri : RT_MISS));
            state[slot] = (hit_any[u] ? RT_ST_HIT : 0u) | (lit ? RT_ST_LIT : 0u) | (hs[u].obj << 8);
            keys[slot] = lit ? light_key(lp, pos) : 0xFFFFFFFFu;
            if (STATS) st.rays += inside[u] ? 1u : 0u;
        }
    }
    if (STATS) flush_stats(st, stats);
}

// The records of the camera rays from the nearest-hit words: what k_wf_primary stores. Every word
// read is set back to "none", so the buffer is ready for the next frame without a pass of its own.
template <bool STATS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_raster_records(SceneView sv, DevRenderParams p, LightProj lp,
                                                          const uint32_t* __restrict__ tile_list,
                                                          unsigned long long* __restrict__ zbuf,
                                                          float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                          uint32_t* __restrict__ keys, DevStats* __restrict__ stats) {
    const PixelMap<BLOCK> pm(p, tile_list, false);
    const rt_f3 o = f3(0.0f, 0.0f, 0.0f), d = camera_dir(p, pm.x, pm.y, pm.s);
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    unsigned long long key = RT_ZKEY_NONE;
    if (pm.inside) {
        unsigned long long* zp = &zbuf[((size_t)pm.y * p.width + pm.x) * p.spp + pm.s];
        key = *zp;
        *zp = RT_ZKEY_NONE;
    }
    const bool hit_any = key != RT_ZKEY_NONE;
    HitD hs;
    hs.tri = hit_any ? (uint32_t)key : 0u;
    hs.t = hit_any ? __uint_as_float((uint32_t)(key >> 32)) : 0.0f;
    hs.u = hs.v = 0.0f;
    hs.obj = 0;
    while (hs.obj + 1 < sv.n_objs && hs.tri >= sv.objs[hs.obj].tri_base + sv.objs[hs.obj].n_tris) ++hs.obj;
    const Surface sf = surface_of(sv, hs, o, d);
    const rt_f3 sun_dir = g_normalize(sun - sf.pos);
    const bool lit = hit_any && g_dot(sf.normal, sun_dir) > 0.0f;
    const uint32_t slot = pm.block * BLOCK + threadIdx.x;
    samp[slot] = make_float4(sf.pos.x, sf.pos.y, sf.pos.z, __uint_as_float(hit_any ? hs.tri : RT_MISS));
    state[slot] = (hit_any ? RT_ST_HIT : 0u) | (lit ? RT_ST_LIT : 0u) | (hs.obj << 8);
    keys[slot] = lit ? light_key(lp, sf.pos) : 0xFFFFFFFFu;
    // a nearest hit whose place in the reference's order could matter is left to k_wf_primary_redo
    hs.near = 0u;   // (nothing is known of the other triangles at this sample)
    if (hit_any && closest_needs_resolve(sv, hs.obj, o, d, g_recip(d), hs)) state[slot] = RT_ST_REDO, keys[slot] = 0xFFFFFFFFu;
    if (STATS) {
        LaneStats st;
        st.rays = pm.inside ? 1u : 0u;
        st.bytes = pm.inside ? 8u : 0u;
        flush_stats(st, stats);
    }
}

#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES RT_WF_WAVES
#endif
template <bool STATS>
__global__ __launch_bounds__(RT_WF_BLOCK, RT_SHADOW_WAVES) void k_wf_shadow(SceneView sv, DevRenderParams p, uint32_t n,
                                                        uint32_t cell_shift,
                                                        const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals,
                                                        const float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                        DevStats* __restrict__ stats,
                                                        const uint32_t* __restrict__ n_valid,
                                                        const uint32_t* __restrict__ done_flag, uint32_t n_blocks) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    uint2* s_beam = reinterpret_cast<uint2*>(s_dyn) + (threadIdx.x >> 6) * p.beam_levels * 64;
    // done_flag: set when k_shadow_cells has taken this launch's rays. Such a launch is the stand-by
    // and comes with a small grid, whose groups take the blocks of rays in turn.
    if (done_flag && *done_flag != 0u) return;
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t i = (uint64_t)xcd_block(blk) * RT_WF_BLOCK + threadIdx.x;
        // n_valid (the library's own sort): the first *n_valid keys carry a shadow ray, what follows is
        // stale. Otherwise all n keys are sorted and those without a ray, all ones, come last.
        const bool act = n_valid ? i < *n_valid : (i < n && keys[i] != 0xFFFFFFFFu);
        if (__ballot(act) == 0ull) continue;
        LaneStats st;
        const uint32_t slot = act ? vals[i] : 0u;
        const float4 a = samp[slot];
        const rt_f3 pos = f3(a.x, a.y, a.z);
        const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
        const rt_f3 sdir = g_normalize(sun - pos);
        const bool occ = any_all_packet<STATS>(sv, pos, sdir, act, s_beam, st, act ? keys[i] >> cell_shift : 0u);
        if (act && occ) state[slot] |= RT_ST_OCC;
        if (STATS) flush_stats(st, stats);
    }
}

// ---------------------------------------------------------------------------------------------
// Shadow rays by light-space cells. Every shadow ray ends at the sun, so seen from the sun a ray is
// one point (its light-space coordinates, light_uv) and a triangle a small patch: a ray can only be
// occluded by the triangles whose patch holds its point. The sort has already grouped the rays by
// cell of the light-space grid (the key's leading bits); here the triangles are listed per cell
// (k_lcell_bin; k_lcell_check says whether the lists came out usable), and k_shadow_cells puts each
// wave's 64 sorted rays against the list of their cell: a box test in light space per (ray,
// triangle), and the pairs that pass queue up in LDS until 64 of them take the exact triangle test
// together, one pair per lane. No tree, no stack; the exact test runs on full waves. The result is
// the tree walk's: whether any triangle passes the reference's test does not depend on the order
// in which they are tried.
//
// A triangle's patch must hold the point of every ray its exact test can accept. The test's
// quantities are signed volumes U, V, A - U - V against thresholds of a few units of rounding of
// M = |s| |e| (s = origin - v0, e an edge), and the distance of the ray from the edge's line, seen
// along the ray, is U / |e x d|: a ray further outside an edge than  c u |s| |e| / |e x d|  (u = 2^-24,
// c below 100 with everything counted, |s| at most the scene's diagonal) fails. The patch is the
// box of the three vertices' light-space coordinates grown by that distance (512 u diag amp with
// amp = max |e| / |e x D|, D the direction to the sun), projected (1 + |uc| + |u|) / |p.w - sun.w|
// per unit of length, plus half a unit of the key for the rounding of the coordinates themselves. A
// triangle with an edge within 1/256 of the direction to the sun (amp above 256; degenerate
// triangles too) gets the patch of its eight-triangle leaf's box instead: the reference tests a
// triangle only for rays that pass the slab test of that box (src/main.rs:103-111,
// src/simd_accel.rs:342-377), whatever the triangle's own shape.
// The trailing n % 8 triangles of an object take the reference's scalar test; k_shadow_cells tries
// them on every ray first, as the tree walk does.
//
// Depth. A triangle occludes a ray only from between the ray's origin and the sun, so an entry also
// carries the least depth of its vertices (depth = distance from the scene's centre along the light's
// dominant axis, growing away from the sun), lowered by a bound on what rounding can accept, and a
// ray nearer the sun than that skips the entry. The bound: the test's point X = o + t d = v0 + u e1 +
// v e2 (exact arithmetic) has depth (1-u-v) z0 + u z1 + v z2 and, for t >= 0, a depth no larger than
// the origin's. Rounding moves each of the test's signed volumes by less than 32 u |s| |e| (u =
// 2^-24), so an accepted ray has exact u, v, 1-u-v >= -eta and t >= -eta max|e| with eta = 32 u diag
// max|e| / |a|, and |a| = |N . d| = G / |S - X| with N = e1 x e2, G = |(S - v0) . N|, S the sun. With
// kappa = 512 u diag max|e| R / G and R = 2 (max |S - Pi| + 2 max|e|), kappa <= 1/4 keeps the computed
// sign of a right and X within R of the sun (a point further out is so far outside the triangle that
// one of the volumes is wrong by G / (8 max|e|), more than rounding gives), hence eta <= kappa / 16:
// the entry's depth is  min z - kappa (|z1-z0| + |z2-z0| + |z2-z1| + max|e|).  Triangles with kappa
// above 1/4 (nearly edge-on to the sun), and those that took their leaf's patch, get no depth: every
// ray of the patch tests them.
// ---------------------------------------------------------------------------------------------
#define RT_LCELL_AMP 256.0f
#define RT_LCELL_MAX_OBJS 32u       // the objects a ray's box tests are kept for, one bit each
#define RT_LCELL_ID_BITS 26u        // triangle ids in a list entry; the object's index above them

struct LightCells {
    uint32_t shift;       // light-space coordinate (16 bits) >> shift = cell coordinate
    uint32_t n_cells;
    float margin;         // 512 u x the scene's diagonal
    float slope_u, slope_v;   // 1 + |uc| + largest |u|, and the same for v
    float away;           // +1 when the light's dominant coordinate grows away from the sun, -1 otherwise
    float depth_span;     // the scene's extent along that axis
};

__device__ __forceinline__ uint32_t lcell_of(uint32_t cx, uint32_t cy) { return spread16(cx) | (spread16(cy) << 1); }

// One thread per triangle: for every cell its patch touches, counts[cell] += 1 and, while the cell's
// list (room for `1 << cap_shift` entries at cell << cap_shift) has room, an entry goes into it: six
// words, the patch (x0, y0, width, height), the id with the object's index above it, and the entry's
// depth. counts keeps counting past the room, so that k_lcell_check sees it.
__global__ __launch_bounds__(256) void k_lcell_bin(SceneView sv, DevRenderParams p, LightProj lp, LightCells lc,
                                                   uint32_t n_tris_total, uint32_t* __restrict__ counts,
                                                   float* __restrict__ lists, uint32_t cap_shift) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n_tris_total) return;
    uint32_t moi = 0;
    while (id >= sv.objs[moi].tri_base + sv.objs[moi].n_tris) ++moi;
    const DevObject& ob = sv.objs[moi];
    const uint32_t local = id - ob.tri_base;
    if (local >= (ob.n_tris & ~7u)) return;   // trailing triangle
    const float* t9 = sv.tri9 + 9 * (size_t)id;
    const rt_f3 e1 = ld3(t9), e2 = ld3(t9 + 3), v0 = ld3(t9 + 6);
    const rt_f3 vb = v0 + e1, vc = v0 + e2, e3 = e2 - e1;
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    const rt_f3 cen = f3((v0.x + vb.x + vc.x) * (1.0f / 3.0f), (v0.y + vb.y + vc.y) * (1.0f / 3.0f),
                         (v0.z + vb.z + vc.z) * (1.0f / 3.0f));
    const rt_f3 D = g_normalize(sun - cen);
    float amp = 1.0f;
    bool ill = false;
    {
        const rt_f3 es[3] = {e1, e2, e3};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const rt_f3 c = g_cross(es[k], D);
            const float cr = __builtin_sqrtf(g_dot(c, c)), len = __builtin_sqrtf(g_dot(es[k], es[k]));
            if (!(cr * RT_LCELL_AMP > len)) ill = true;   // NaN too
            else amp = fmaxf(amp, len / cr);
        }
    }
    float x0 = RT_INF, x1 = -RT_INF, y0 = RT_INF, y1 = -RT_INF, rwmax = 0.0f;
    auto take = [&](rt_f3 q) {
        float qu, qv;
        light_uv(lp, q, qu, qv);
        x0 = fminf(x0, qu), x1 = fmaxf(x1, qu), y0 = fminf(y0, qv), y1 = fmaxf(y1, qv);
        rwmax = fmaxf(rwmax, __builtin_fabsf(1.0f / (pick3(q, lp.w) - lp.sun_w)));
    };
    if (!ill) {
        take(v0), take(vb), take(vc);
    } else {
        const float* lb = sv.leaf_bounds + 6 * (size_t)(ob.leaf_base + (local >> 3));
        const rt_f3 lo = ld3(lb), hi = ld3(lb + 3);
        amp = 1.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) take(f3(c & 1 ? hi.x : lo.x, c & 2 ? hi.y : lo.y, c & 4 ? hi.z : lo.z));
    }
    // Least depth a ray must have for this triangle to lie between it and the sun (see above).
    float zcull = -RT_INF;
    if (!ill) {
        const float z0 = (pick3(v0, lp.w) - lp.c_w) * lc.away, z1 = (pick3(vb, lp.w) - lp.c_w) * lc.away,
                    z2 = (pick3(vc, lp.w) - lp.c_w) * lc.away;
        const rt_f3 N = g_cross(e1, e2);
        const rt_f3 s0 = sun - v0, s1 = sun - vb, s2 = sun - vc;
        const float G = __builtin_fabsf(g_dot(s0, N));
        const float r0 = __builtin_sqrtf(fmaxf(g_dot(s0, s0), fmaxf(g_dot(s1, s1), g_dot(s2, s2))));
        const float emax = __builtin_sqrtf(fmaxf(g_dot(e1, e1), fmaxf(g_dot(e2, e2), g_dot(e3, e3))));
        const float kappa = lc.margin * emax * (2.0f * (r0 + 2.0f * emax)) / G * 1.001f;
        if (kappa <= 0.25f) {   // false for NaN
            const float zmin = fminf(z0, fminf(z1, z2)), zabs = fmaxf(__builtin_fabsf(z0), fmaxf(__builtin_fabsf(z1), __builtin_fabsf(z2)));
            const float spread = __builtin_fabsf(z1 - z0) + __builtin_fabsf(z2 - z0) + __builtin_fabsf(z2 - z1) + emax;
            zcull = zmin - kappa * spread * 1.001f - 4e-6f * (zabs + lc.depth_span);
            if (!(zcull == zcull)) zcull = -RT_INF;
        }
    }
    const float grow = lc.margin * amp * rwmax * 1.0001f;
    const float padx = grow * lc.slope_u * lp.uscale + 0.5f + 1e-6f * fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1));
    const float pady = grow * lc.slope_v * lp.vscale + 0.5f + 1e-6f * fmaxf(__builtin_fabsf(y0), __builtin_fabsf(y1));
    x0 -= padx, x1 += padx, y0 -= pady, y1 += pady;
    if (!(x0 <= x1) || !(y0 <= y1)) x0 = y0 = -RT_INF, x1 = y1 = RT_INF;   // NaN: every cell (and the list overflows)
    // widths rounded up, so that q - x0 <= width holds in float whenever q <= x1 does
    const float bw = (x1 - x0) * 1.000001f + 1e-30f, bh = (y1 - y0) * 1.000001f + 1e-30f;
    const uint32_t cx0 = (uint32_t)fminf(fmaxf(x0, 0.0f), 65535.0f) >> lc.shift;
    const uint32_t cx1 = (uint32_t)fminf(fmaxf(x1, 0.0f), 65535.0f) >> lc.shift;
    const uint32_t cy0 = (uint32_t)fminf(fmaxf(y0, 0.0f), 65535.0f) >> lc.shift;
    const uint32_t cy1 = (uint32_t)fminf(fmaxf(y1, 0.0f), 65535.0f) >> lc.shift;
    for (uint32_t cy = cy0; cy <= cy1; ++cy)
        for (uint32_t cx = cx0; cx <= cx1; ++cx) {
            const uint32_t cell = lcell_of(cx, cy);
            const uint32_t k = atomicAdd(&counts[cell], 1u);
            if (k < (1u << cap_shift)) {
                // one record of 24 bytes (8-byte aligned): three stores into one or two 32-byte sectors
                float2* rec = reinterpret_cast<float2*>(lists + 6 * (((size_t)cell << cap_shift) + k));
                rec[0] = make_float2(x0, y0);
                rec[1] = make_float2(bw, bh);
                rec[2] = make_float2(__uint_as_float(id | (moi << RT_LCELL_ID_BITS)), zcull);
            }
        }
}

// flags[0] = 1 when no list is longer than its room and the lists of the cells that have one average
// at most max_mean entries: then the shadow rays go by cells, otherwise by tree walk. flags[1..3] =
// the lists' total, the longest, the cells with a list; flags[4] counts the workgroups that are done
// (the last one decides) and is zero again afterwards.
// What k_shadow_cells needs only at its very end, kept beside the flags (from word 16 on) and read from
// there: as kernel arguments these would sit in scalar registers through the kernel's loop, which has
// none to spare.
struct CellsTail {
    const float4* leaf_gate;
    uint32_t* redo;
    uint32_t redo_all;   // test switch: every ray the lists found occluded goes to k_shadow_redo
};
__global__ __launch_bounds__(256) void k_lcell_check(const uint32_t* __restrict__ counts, uint32_t n, uint32_t cap,
                                                     uint32_t max_mean, uint32_t* __restrict__ flags, CellsTail tail) {
    __shared__ uint32_t s_last;
    uint32_t total = 0, longest = 0, used = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t c = counts[i];
        total += c, longest = max(longest, c), used += c != 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        total += __shfl_xor(total, off, 64);
        longest = max(longest, (uint32_t)__shfl_xor(longest, off, 64));
        used += __shfl_xor(used, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&flags[1], total);
        atomicMax(&flags[2], longest);
        atomicAdd(&flags[3], used);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&flags[4], 1u) == gridDim.x - 1u;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        const uint32_t t = atomicAdd(&flags[1], 0u), m = atomicAdd(&flags[2], 0u), u = atomicAdd(&flags[3], 0u);
        const bool ok = m <= cap && (uint64_t)t <= (uint64_t)max_mean * max(u, 1u);
        flags[0] = ok ? 1u : 0u;
        flags[4] = 0u;
        *reinterpret_cast<CellsTail*>(flags + 16) = tail;
    }
}

struct ShadowCellsLds {
    float ray[6][64];       // origin and direction of the wave's rays
    alignas(512) uint32_t queue[128];   // ring of (triangle entry << 6 | ray's lane) waiting for the exact test
    uint32_t hit[2];        // rays the last batch of exact tests found occluded
    uint32_t occluder[64];  // per ray: a triangle the exact test accepted
};
static_assert(sizeof(uint32_t[128]) == 512, "ring_append wraps the ring's byte offset with a mask of 0x1fc");

// The lanes of mask m append (e << 6 | lane) to the ring of 128 words at LDS address `ring` (a
// multiple of 512), from position `tail` on, in lane order. Written out because the compiler turns
// the same thing in C++ (ballot of a combined predicate, then a branch on the lane's own bit) into
// twice the instructions: here the mask goes into exec as it is.
__device__ __forceinline__ void ring_append(unsigned long long m, uint32_t tail, uint32_t e, uint32_t lane, uint32_t ring) {
    uint32_t at, word;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b32 exec_lo, %[mlo]\n\t"
        "s_mov_b32 exec_hi, %[mhi]\n\t"
        "v_mbcnt_lo_u32_b32 %[at], %[mlo], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[at], %[mhi], %[at]\n\t"
        "v_add_lshl_u32 %[at], %[at], %[tail], 2\n\t"
        "v_and_or_b32 %[at], %[at], %[wrap], %[ring]\n\t"
        "v_lshl_or_b32 %[word], %[e], 6, %[lane]\n\t"
        "ds_write_b32 %[at], %[word]\n\t"
        "s_mov_b64 exec, %[saved]"
        : [saved] "=&s"(saved), [at] "=&v"(at), [word] "=&v"(word)
        : [mlo] "s"((uint32_t)m), [mhi] "s"((uint32_t)(m >> 32)), [tail] "s"(tail), [wrap] "s"(0x1fcu), [ring] "v"(ring),
          [e] "s"(e), [lane] "v"(lane)
        : "memory");
}

template <bool STATS, bool ONE_OBJ>
__global__ __launch_bounds__(64, 8) void k_shadow_cells(SceneView sv, DevRenderParams p, LightProj lp,
                                                        const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals,
                                                        const float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                        DevStats* __restrict__ stats,
                                                        const uint32_t* __restrict__ n_valid,
                                                        const uint32_t* __restrict__ counts,
                                                        const float* __restrict__ lists, uint32_t cap_shift,
                                                        uint32_t* __restrict__ flags) {
    __shared__ ShadowCellsLds lds;
    if (flags[0] == 0u) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t i = (uint64_t)xcd_block(blockIdx.x) * 64u + lane;
    const bool act = i < *n_valid;
    unsigned long long alive = __ballot(act);
    if (alive == 0ull) return;
    LaneStats st;
    const uint32_t slot = act ? vals[i] : 0u;
    const uint32_t key = act ? keys[i] : 0xFFFFFFFFu;
    const float4 a = samp[slot];
    const rt_f3 o = f3(a.x, a.y, a.z);
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    const rt_f3 d = g_normalize(sun - o);
    if (STATS && act) st.rays += 1;
    float qu, qv;
    light_uv(lp, o, qu, qv);
    // the origin's depth, as k_lcell_bin takes the vertices'
    const float depth = (pick3(o, lp.w) - lp.c_w) * (lp.c_w > lp.sun_w ? 1.0f : -1.0f);
    lds.ray[0][lane] = o.x, lds.ray[1][lane] = o.y, lds.ray[2][lane] = o.z;
    lds.ray[3][lane] = d.x, lds.ray[4][lane] = d.y, lds.ray[5][lane] = d.z;
    if (lane < 2u) lds.hit[lane] = 0u;
    lds.occluder[lane] = 0xFFFFFFFFu;   // none from the lists so far
    // The objects' boxes (src/main.rs:178-180), one bit per object, and their trailing triangles.
    uint32_t objmask = 0u;
    {
        const rt_f3 inv = g_recip(d);
        for (uint32_t oi = 0; oi < sv.n_objs; ++oi) {
            const char* op = reinterpret_cast<const char*>(sv.objs) + 64 * (size_t)oi;
            const f32x16_t ob = sload16(op);
            const rt_f3 bbmin = f3(ob[0], ob[1], ob[2]), bbmax = f3(ob[3], ob[4], ob[5]);
            const uint32_t rest_base = __float_as_uint(ob[10]), n_rest = __float_as_uint(ob[11]);
            if (STATS && act) st.box += 1;
            const bool in_obj = act && box_intersection_check(o, inv, bbmin, bbmax);
            objmask |= in_obj ? 1u << oi : 0u;
            for (uint32_t r = 0; r < n_rest; ++r) {
                const char* pv = reinterpret_cast<const char*>(sv.rest_verts) + 36 * (size_t)(rest_base + r);
                const f32x8_t va = sload8(pv);
                const float vc2 = __uint_as_float(sload1u(pv + 32));
                float t, u, v;
                if (STATS && in_obj) st.tri += 1;
                const bool h = in_obj && triangle_ray_hits(f3(va[0], va[1], va[2]), f3(va[3], va[4], va[5]),
                                                           f3(va[6], va[7], vc2), o, d, RT_INF, &t, &u, &v);
                alive &= ~__ballot(h);
            }
        }
    }
    // A ray that is occluded, or whose object's box it misses (one object), takes no further part:
    // its light-space coordinate becomes NaN, which is inside no box.
    const float nan = __uint_as_float(0x7FC00000u);
    if (!((alive >> lane) & 1ull) || (ONE_OBJ && !(objmask & 1u))) qu = nan;
    uint32_t head = 0, tail = 0;   // ring positions, wave-uniform
    // The exact test for the m oldest pairs of the ring, one per lane.
    auto flush = [&](uint32_t m) {
        wave_lds_sync();
        bool h = false;
        if (lane < m) {
            const uint32_t e = lds.queue[(head + lane) & 127u];
            const uint32_t rl = e & 63u;
            const float* t9 = sv.tri9 + 9 * (size_t)(e >> 6);
            const rt_f3 e1 = ld3(t9), e2 = ld3(t9 + 3), v0 = ld3(t9 + 6);
            const rt_f3 ro = f3(lds.ray[0][rl], lds.ray[1][rl], lds.ray[2][rl]);
            const rt_f3 rd = f3(lds.ray[3][rl], lds.ray[4][rl], lds.ray[5][rl]);
            const rt_tri_eval ev = tri_intersect(e1, e2, v0, ro, rd);
            // src/main.rs:185-193: a t with a clear sign bit occludes
            h = !ev.failed && !__builtin_signbit(ev.t) && !(ev.t != ev.t);
            if (STATS) st.tri += 1;
            // the ray's occluder is kept for the leaf gate at the end (any one of them, if several lanes
            // write at once)
            if (h) atomicOr(&lds.hit[rl >> 5], 1u << (rl & 31u)), lds.occluder[rl] = e >> 6;
        }
        head += m;
        if (__ballot(h) != 0ull) {
            wave_lds_sync();
            const uint32_t h0 = __builtin_amdgcn_readfirstlane(lds.hit[0]), h1 = __builtin_amdgcn_readfirstlane(lds.hit[1]);
            const unsigned long long hm = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
            alive &= ~hm;
            if ((hm >> lane) & 1ull) qu = nan;
            wave_lds_sync();
            if (lane < 2u) lds.hit[lane] = 0u;
        }
    };
    // One list entry against the rays of the current cell (quc: their coordinate, NaN for the others).
    float quc = nan;
    const uint32_t ring = (uint32_t)(uintptr_t)&lds.queue[0];   // an LDS address is the low half of the flat one
    auto entry = [&](float x0, float y0, float bw, float bh, uint32_t e, float zcull) {
        // 0 <= q - x0 <= x1 - x0 as one unsigned comparison of the floats' bits: a negative difference
        // has its sign bit set and compares above every positive width; so does a NaN. One vote per
        // comparison, the masks combined as scalars: each vote is then the comparison itself.
        unsigned long long m = __ballot(__float_as_uint(quc - x0) <= __float_as_uint(bw)) &
                               __ballot(__float_as_uint(qv - y0) <= __float_as_uint(bh)) & __ballot(depth >= zcull);
        if (!ONE_OBJ) m &= __ballot(((objmask >> (e >> RT_LCELL_ID_BITS)) & 1u) != 0u);
        if (m == 0ull) return;
        ring_append(m, tail, e, lane, ring);   // the object's bits fall off the top of e << 6
        tail += (uint32_t)__builtin_popcountll(m);
        if (tail - head >= 64u) {
            flush(64u);
            if (qu != qu) quc = nan;
        }
    };
    // four entries of six words: sixteen words and eight more
    auto four = [&](const f32x16_t& B, const f32x8_t& E, uint32_t cnt) {
        if (STATS && quc == quc) st.box += cnt;
        entry(B[0], B[1], B[2], B[3], __float_as_uint(B[4]), B[5]);
        if (cnt > 1u) entry(B[6], B[7], B[8], B[9], __float_as_uint(B[10]), B[11]);
        if (cnt > 2u) entry(B[12], B[13], B[14], B[15], __float_as_uint(E[0]), E[1]);
        if (cnt > 3u) entry(E[2], E[3], E[4], E[5], __float_as_uint(E[6]), E[7]);
    };
    unsigned long long todo = alive;
    while (todo != 0ull) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(todo));
        const bool mine = key == c;
        const unsigned long long grp = __ballot(mine);
        todo &= ~grp;
        quc = mine ? qu : nan;
        const uint32_t beg = c << cap_shift, end = beg + sload1u(counts + c);
        // The lists are read four entries at a time through the scalar cache (an entry is x0, y0, x1 - x0,
        // y1 - y0, id, depth), the next four while the present four are worked on: the address of the next read
        // is made to depend on the present data (a width's sign bit, always clear), which keeps the
        // compiler from issuing it ahead of the wait for that data and then waiting for both.
        if (beg == end) continue;
        f32x16_t B0 = sload16(lists + 6 * (size_t)beg), B1 = B0;
        f32x8_t E0 = sload8(lists + 6 * (size_t)beg + 16), E1 = E0;
        for (uint32_t j = beg;;) {
            if ((alive & grp) == 0ull) break;
            uint32_t nx = j + 4u + (__float_as_uint(B0[2]) >> 31);
            if (nx < end) B1 = sload16(lists + 6 * (size_t)nx), E1 = sload8(lists + 6 * (size_t)nx + 16);
            four(B0, E0, min(4u, end - j));
            j = nx;
            if (j >= end || (alive & grp) == 0ull) break;
            nx = j + 4u + (__float_as_uint(B1[2]) >> 31);
            if (nx < end) B0 = sload16(lists + 6 * (size_t)nx), E0 = sload8(lists + 6 * (size_t)nx + 16);
            four(B1, E1, min(4u, end - j));
            j = nx;
            if (j >= end) break;
        }
    }
    if (tail != head) flush(tail - head);
    wave_lds_sync();
    // The reference tests a triangle only for rays that pass the box of its leaf (ref_leaf_gate;
    // src/main.rs:192-200, 214-216). Every ray the lists found occluded puts the occluder it was
    // found with through that test, here at the end where the loop's registers are free. A ray whose
    // occluder fails (a needle-shaped triangle whose test rounds badly) has been taken out of the search
    // too early: it goes to the queue of k_shadow_redo, which runs the reference's own search for it.
    bool occ = act && !((alive >> lane) & 1ull);
    const uint32_t tri = lds.occluder[lane];
    if (occ && tri != 0xFFFFFFFFu) {
        const rt_f3 ro = f3(lds.ray[0][lane], lds.ray[1][lane], lds.ray[2][lane]);
        const rt_f3 rd = f3(lds.ray[3][lane], lds.ray[4][lane], lds.ray[5][lane]);
        const CellsTail* ct = reinterpret_cast<const CellsTail*>(flags + 16);
        if (ONE_OBJ) {
            // one object: leaf l holds the triangles 8 l .. 8 l + 7
            const float4* lg = ct->leaf_gate + 2 * (size_t)(tri >> 3);
            occ = ref_leaf_gate(lg[0], lg[1], ro, g_recip(rd));
        } else {
            occ = ref_leaf_gate(sv, tri, ro, g_recip(rd));
        }
        occ = occ && ct->redo_all == 0u;
        if (!occ) ct->redo[atomicAdd(&flags[5], 1u)] = slot;
    }
    if (occ) state[slot] |= RT_ST_OCC;
    if (STATS) {
        st.bytes = 0;
        flush_stats(st, stats);
    }
}

// The reference's shadow search (src/main.rs:185-233) for the few rays k_shadow_cells could not
// decide (flags[5] of them, their sample slots in `queue`): one wave per ray, the lanes share the
// object's leaves out among themselves; per leaf the reference's box test, then its eight triangles.
// Any occluder will do, so the order in which the leaves are taken does not matter.
__global__ __launch_bounds__(64) void k_shadow_redo(SceneView sv, DevRenderParams p, const uint32_t* __restrict__ queue,
                                                    const uint32_t* __restrict__ flags, const float4* __restrict__ samp,
                                                    uint32_t* __restrict__ state) {
    const uint32_t n = flags[5], lane = threadIdx.x;
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint32_t slot = queue[q];
        const float4 a = samp[slot];
        const rt_f3 o = f3(a.x, a.y, a.z), d = g_normalize(sun - o), inv = g_recip(d);
        bool found = false;
        for (uint32_t oi = 0; oi < sv.n_objs && !found; ++oi) {
            const DevObject& ob = sv.objs[oi];
            if (!box_intersection_check(o, inv, ld3(ob.bbmin), ld3(ob.bbmax))) continue;
            const uint32_t n_leaves = ob.n_tris >> 3;
            for (uint32_t l0 = 0; l0 < n_leaves && !found; l0 += 64u) {
                const uint32_t leaf = l0 + lane;
                bool h = false;
                if (leaf < n_leaves && ref_leaf_gate(sv, ob, ob.tri_base + 8u * leaf, o, inv)) {
                    for (uint32_t j = 0; j < 8u; ++j) {
                        const float* t9 = sv.tri9 + 9 * (size_t)(ob.tri_base + 8u * leaf + j);
                        const rt_tri_eval ev = tri_intersect(ld3(t9), ld3(t9 + 3), ld3(t9 + 6), o, d);
                        h = h || (!ev.failed && !__builtin_signbit(ev.t) && !(ev.t != ev.t));
                    }
                }
                found = __ballot(h) != 0ull;
            }
            for (uint32_t r0 = 0; r0 < ob.n_rest && !found; r0 += 64u) {
                const uint32_t r = r0 + lane;
                bool h = false;
                if (r < ob.n_rest) {
                    const float* pv = sv.rest_verts + 9 * (size_t)(ob.rest_base + r);
                    float t, u, v;
                    h = triangle_ray_hits(ld3(pv), ld3(pv + 3), ld3(pv + 6), o, d, RT_INF, &t, &u, &v);
                }
                found = __ballot(h) != 0ull;
            }
        }
        if (found && lane == 0u) state[slot] |= RT_ST_OCC;
    }
}

// Splits the bits of a pixel's Morton number inside its tile into column and row.
__device__ __forceinline__ void morton_xy(uint32_t m, uint32_t& x, uint32_t& y) {
    x = 0, y = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        x |= ((m >> (2 * b)) & 1u) << b;
        y |= ((m >> (2 * b + 1)) & 1u) << b;
    }
}

// One wave shades 64 * RT_RESOLVE_SLOTS consecutive sample slots, lane l the slots l, l + 64, ...: the
// loads of all of a lane's slots are on their way before the first is used (a kernel with one slot
// per lane is bound by the two dependent loads of each wave, record then normal and colour, not by
// bytes), each load is a contiguous kilobyte, and the samples of a pixel (spp consecutive slots, spp a
// power of two up to 64) meet in one wave. A slot's place in its tile is tile part + slot part + lane
// part without carries between them, so its Morton number splits into three that are or-ed together,
// the first two on the scalar side.
#define RT_RESOLVE_SLOTS 4
__global__ __launch_bounds__(64) void k_wf_resolve(SceneView sv, DevRenderParams p, const uint32_t* __restrict__ tile_list,
                                                   const float4* __restrict__ samp, const uint32_t* __restrict__ state,
                                                   uint8_t* __restrict__ rgba8, uint32_t* __restrict__ hit_ids) {
    constexpr uint32_t K = RT_RESOLVE_SLOTS, PER = 64u * K;
    __shared__ float s_col[3 * PER];
    const uint32_t lane = threadIdx.x, spp = p.spp, ss = 31u - (uint32_t)__builtin_clz(spp);
    // PER divides a tile's RT_TILE * RT_TILE * spp slots: (1024 / PER) << ss waves a tile
    const uint32_t wshift = ss + 2u;
    static_assert(RT_TILE * RT_TILE / PER == 4u, "waves per tile are taken as 4 * spp");
    const uint32_t launch_tile = blockIdx.x >> wshift, in_tile = blockIdx.x & ((1u << wshift) - 1u);
    const uint32_t local_tile = p.tile_first + launch_tile;
    const uint32_t tile = (tile_list && launch_tile < p.n_tiles) ? tile_list[local_tile] : local_tile;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const uint32_t slot0 = blockIdx.x * PER + lane;
    float4 a[K];
    uint32_t stt[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) a[k] = samp[slot0 + 64u * k], stt[k] = state[slot0 + 64u * k];
    rt_f3 normal[K];
    Col tcol[K];
    float albedo[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        normal[k] = f3(0.0f, 0.0f, 0.0f), tcol[k] = Col{0.0f, 0.0f, 0.0f}, albedo[k] = 0.0f;
        if (stt[k] & RT_ST_HIT) {
            const float* td = sv.tri_data + 6 * (size_t)__float_as_uint(a[k].w);
            normal[k] = ld3(td);
            tcol[k] = Col{td[3], td[4], td[5]};
            albedo[k] = sv.objs[stt[k] >> 8].albedo;
        }
    }
    uint32_t ax, ay, lx, ly;
    morton_xy((in_tile * PER) >> ss, ax, ay);
    morton_xy(lane >> ss, lx, ly);
    const uint32_t s = lane & (spp - 1u);
    const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
    uint32_t x[K], y[K];
    bool inside[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        uint32_t kx, ky;
        morton_xy((64u * k) >> ss, kx, ky);
        x[k] = tx * RT_TILE + (ax | kx | lx), y[k] = ty * RT_TILE + (ay | ky | ly);
        inside[k] = x[k] < p.width && y[k] < p.height;
        const uint32_t tri = __float_as_uint(a[k].w);
        Col color{0.0f, 0.0f, 0.0f};
        if (stt[k] & RT_ST_HIT) {
            // src/main.rs:176-241 on the stored hit point: the same operations as in k_render.
            const rt_f3 sun_dir = g_normalize(sun - f3(a[k].x, a[k].y, a[k].z));
            if ((stt[k] & RT_ST_LIT) && !(stt[k] & RT_ST_OCC)) color = cmul(tcol[k], g_dot(normal[k], sun_dir) * albedo[k]);
            color = gamma22(color);
        } else {
            color = inside[k] ? sky() : Col{0.0f, 0.0f, 0.0f};
        }
        if (hit_ids && inside[k]) hit_ids[((uint64_t)y[k] * p.width + x[k]) * spp + s] = tri;
        float* c = &s_col[3 * (64u * k + lane)];
        c[0] = color.r, c[1] = color.g, c[2] = color.b;
    }
    wave_lds_sync();
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        if (s != 0u || !inside[k]) continue;
        const float* c = &s_col[3 * (64u * k + lane)];
        Col acc = Col{c[0], c[1], c[2]};
        for (uint32_t j = 1; j < spp; ++j) acc = cadd(acc, Col{c[3 * j], c[3 * j + 1], c[3 * j + 2]});
        acc = cdiv(acc, (float)spp);
        const uint32_t px = to_u8(acc.r) | (to_u8(acc.g) << 8) | (to_u8(acc.b) << 16) | 0xff000000u;
        size_t off;
        if (p.compact) off = ((size_t)local_tile * RT_TILE * RT_TILE + (size_t)(y[k] - ty * RT_TILE) * RT_TILE + (x[k] - tx * RT_TILE)) * 4;
        else off = ((size_t)y[k] * p.width + x[k]) * 4;
        *reinterpret_cast<uint32_t*>(rgba8 + off) = px;
    }
}

// Scatter of gathered tile-compact slabs into a frame. Tile t belongs to shard (tx + ty) % count;
// its slot inside the shard's slab is its rank among that shard's tiles in ascending tile id,
// given by tile_slot[t].
// ---------------------------------------------------------------------------------------------
// Frames with diffuse scattering at the camera ray's hit (DIFFUSE_DEPTH = 2), as a pipeline:
//   k_wf_primary, sort, k_wf_shadow       camera rays and their shadow rays, as above
//   k_wf_bounce, once per level           level 0: the scatter rays of every hit sample and, off a
//                                         mirror, the mirror ray; level l: the mirror bounces off the
//                                         hits of level l - 1. Nearest hit by the per-lane walk (the
//                                         rays share origins, not directions); per ray: hit point,
//                                         id, state, light-space key
//   sort, k_wf_shadow                     the shadow rays of all levels, in light-space order
//   k_wf_resolve_tree                     shading of every level, gamma, average, RGBA8
// Small kernels instead of one: the per-lane walk needs occupancy more than anything (8 waves per
// SIMD here, 3 in the one-kernel frame), lanes whose sample has no ray at some level are refilled
// instead of idling, and half of all rays are shadow rays that the sorted beam walk handles an
// order of magnitude faster.
// ---------------------------------------------------------------------------------------------
// Pixel and sample of a launch-relative sample slot (the inverse of PixelMap's slot numbering).
struct SlotMap {
    uint32_t x, y, s;
    uint64_t sample_index;
    bool inside;
    __device__ __forceinline__ SlotMap(const DevRenderParams& p, const uint32_t* tile_list, uint32_t slot) {
        const uint32_t per_tile = RT_TILE * RT_TILE * p.spp;
        const uint32_t launch_tile = slot / per_tile, rest = slot % per_tile;
        const uint32_t local_tile = p.tile_first + launch_tile;
        const uint32_t tile = tile_list ? tile_list[local_tile] : local_tile;
        const uint32_t pix_in_tile = rest / p.spp;
        s = rest % p.spp;
        uint32_t mx = 0, my = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            mx |= ((pix_in_tile >> (2 * b)) & 1u) << b;
            my |= ((pix_in_tile >> (2 * b + 1)) & 1u) << b;
        }
        x = (tile % p.tiles_x) * RT_TILE + mx;
        y = (tile / p.tiles_x) * RT_TILE + my;
        inside = x < p.width && y < p.height;
        sample_index = ((uint64_t)y * p.width + x) * p.spp + s;
    }
};

// The samples k_wf_primary has left undecided (RT_ST_REDO), by the per-lane search with closest_resolve. A few
// thousand waves go over all the slots' state words; a marked slot's pixel and sample come from its number.
__global__ __launch_bounds__(64) void k_wf_primary_redo(SceneView sv, DevRenderParams p, LightProj lp,
                                                        const uint32_t* __restrict__ tile_list, uint32_t n_slots,
                                                        float4* __restrict__ samp, uint32_t* __restrict__ state,
                                                        uint32_t* __restrict__ keys) {
    using Stack = LaneStackT<RT_STACK_LDS, 64>;
    __shared__ uint2 s_stack[RT_STACK_LDS * 64];
    uint2 stack_rest[Stack::REST];
    Stack stk;
    stk.lds = &s_stack[threadIdx.x];
    stk.spill = stack_rest;
    LaneStats st;
    // four state words a lane and turn (n_slots is a multiple of 1,024: whole tiles)
    for (uint32_t s4 = (blockIdx.x * 64u + threadIdx.x) * 4u; s4 < n_slots; s4 += gridDim.x * 256u) {
        const uint4 w = *reinterpret_cast<const uint4*>(state + s4);
        if (w.x != RT_ST_REDO && w.y != RT_ST_REDO && w.z != RT_ST_REDO && w.w != RT_ST_REDO) continue;
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        for (uint32_t k = 0; k < 4u; ++k) {
            if (ws[k] != RT_ST_REDO) continue;
            const uint32_t slot = s4 + k;
            const SlotMap sm(p, tile_list, slot);
            const rt_f3 o = f3(0.0f, 0.0f, 0.0f), d = camera_dir(p, sm.x, sm.y, sm.s);
            const HitD h = closest_all<false>(sv, o, d, stk, st);
            primary_record(sv, lp, f3(p.sun[0], p.sun[1], p.sun[2]), true, h, o, d, slot, samp, state, keys);
        }
    }
}

// Mirror direction of src/main.rs:245.
__device__ __forceinline__ rt_f3 reflect_dir(rt_f3 d, rt_f3 n) { return d - 2.0f * n * g_dot(d, n); }

// Ray j of level `level` of the secondary-ray tree: branch j % branches of sample slot j / branches.
// Level 0 leaves the camera ray's hit (scatter direction k for branch k < scatt, the mirror direction
// for the branch after them) and runs over every j; level l > 0 is the mirror bounce off that
// branch's hit at level l - 1 and runs over the queue of rays the level before filled: those that
// hit a mirror. Directions are not stored: they follow from the stored triangle ids along the branch.
// Records of rays that do not exist: level 0 stores "no hit, no shadow ray" itself, the deeper
// levels are preset to that before the first launch.
// Follow-up rays a wave collects in LDS before it writes them to the next level's queue with one atomic (an
// atomic per finishing turn on the one counter all waves share doubled the time of a frame full of mirrors).
// With its count the list is 128 bytes: what thirteen stack entries per lane leave of a walk's 5,120.
#define RT_QUEUE_LDS 31
struct TreeSource {
    static constexpr bool QUEUES = true;
    const SceneView& sv;
    const DevRenderParams& p;
    const LightProj& lp;
    const RayTree& tr;
    uint32_t level;
    const uint32_t* tile_list;
    const float4* samp;
    const uint32_t* state;
    float4* dsamp;
    uint32_t* dstate;
    uint32_t* dkeys;
    const uint32_t* queue;  // this level's rays (level > 0)
    uint32_t* next_queue;   // the next level's rays and their count, null at the last level
    uint32_t* next_count;
    uint32_t* s_q;          // LDS, per wave: RT_QUEUE_LDS entries, then their count
    __device__ __forceinline__ bool none(uint32_t j) const {
        dstate[j] = 0u;
        dkeys[j] = 0xFFFFFFFFu;
        return false;
    }
    __device__ __forceinline__ bool init(uint32_t idx, rt_f3& o, rt_f3& d) const {
        const uint32_t j = level ? queue[idx] : idx;
        const uint32_t slot = j / tr.branches, k = j % tr.branches;
        const uint32_t ps = state[slot];
        const float4 a = samp[slot];   // read beside the state word, not after it: a refill waits for memory once less
        if (level == 0) {
            if (!(ps & RT_ST_HIT)) return none(j);
            if (k >= tr.scatt && !(sv.objs[ps >> 8].reflectivity > 0.0f)) return none(j);
        }
        const SlotMap sm(p, tile_list, slot);
        const rt_f3 normal = ld3(sv.tri_data + 6 * (size_t)__float_as_uint(a.w));
        o = f3(a.x, a.y, a.z);
        if (k < tr.scatt) {
            rt_f3 tang1, tang2;
            tangent_vectors(normal, &tang1, &tang2);
            d = diffuse_dir(p.seed, sm.sample_index, k, normal, tang1, tang2);
        } else {
            d = reflect_dir(camera_dir(p, sm.x, sm.y, sm.s), normal);
        }
        for (uint32_t l = 0; l < level; ++l) {
            const float4 r = dsamp[(size_t)l * tr.level_rays + j];
            d = reflect_dir(d, ld3(sv.tri_data + 6 * (size_t)__float_as_uint(r.w)));
            o = f3(r.x, r.y, r.z);
        }
        return true;
    }
    __device__ __forceinline__ bool marked(uint32_t idx) const {
        const uint32_t j = level ? queue[idx] : idx;
        return dstate[(size_t)level * tr.level_rays + j] == RT_ST_REDO;
    }
    __device__ __forceinline__ void emit(uint32_t idx, const HitD& h, rt_f3 o, rt_f3 d) const {
        const uint32_t j = level ? queue[idx] : idx;
        const size_t i = (size_t)level * tr.level_rays + j;
        bool spawn = false;
        if (h.tri == RT_MISS || h.tri == RT_REDO) {
            dstate[i] = h.tri == RT_REDO ? RT_ST_REDO : 0u;   // (RT_ST_REDO: k_wf_bounce_redo takes the ray up again)
            dkeys[i] = 0xFFFFFFFFu;
        } else {
            const Surface s2 = surface_of(sv, h, o, d);
            const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
            const rt_f3 sun_dir = g_normalize(sun - s2.pos);
            const bool lit = g_dot(s2.normal, sun_dir) > 0.0f;
            dsamp[i] = make_float4(s2.pos.x, s2.pos.y, s2.pos.z, __uint_as_float(h.tri));
            dstate[i] = RT_ST_HIT | (lit ? RT_ST_LIT : 0u) | (h.obj << 8);
            dkeys[i] = lit ? light_key(lp, s2.pos) : 0xFFFFFFFFu;
            spawn = next_queue != nullptr && s2.reflectivity > 0.0f;
        }
        // The lanes that emit in this turn and bounce on append to the wave's list together; a list that
        // cannot take them is written out first, by the lanes that are here (emit runs for the lanes that
        // finish in a turn, not for the wave), and more than the list holds go out by themselves. The
        // hardware serves a wave's LDS instructions in order; the fences keep the compiler from moving or
        // caching them.
        const unsigned long long mask = __ballot(spawn);
        if (mask != 0ull) {
            const uint32_t lane = threadIdx.x & 63u;
            const unsigned long long here = __ballot(true);
            const int first = __builtin_ctzll(here);
            const uint32_t add = (uint32_t)__builtin_popcountll(mask);
            const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            wave_lds_sync();
            uint32_t n = s_q[RT_QUEUE_LDS];
            if (add > RT_QUEUE_LDS) {
                uint32_t base = 0;
                if ((int)lane == first) base = atomicAdd(next_count, add);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (spawn) next_queue[base + rank] = j;
            } else {
                if (n + add > RT_QUEUE_LDS) {
                    const uint32_t me = (uint32_t)__builtin_popcountll(here & ((1ull << lane) - 1ull));
                    const uint32_t lanes = (uint32_t)__builtin_popcountll(here);
                    uint32_t base = 0;
                    if ((int)lane == first) base = atomicAdd(next_count, n);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                    for (uint32_t i = me; i < n; i += lanes) next_queue[base + i] = s_q[i];
                    wave_lds_sync();
                    n = 0;
                }
                if (spawn) s_q[n + rank] = j;
                if ((int)lane == first) s_q[RT_QUEUE_LDS] = n + add;
                wave_lds_sync();
            }
        }
    }
    // Whole wave, at the end of the walk: writes out what is left in the list.
    __device__ __forceinline__ void flush(uint32_t lane, bool all) const {
        if (!all || !next_queue) return;
        wave_lds_sync();
        const uint32_t n = s_q[RT_QUEUE_LDS];
        if (n == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(next_count, n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t i = lane; i < n; i += 64u) next_queue[base + i] = s_q[i];
        wave_lds_sync();
        if (lane == 0) s_q[RT_QUEUE_LDS] = 0u;
        wave_lds_sync();
    }
};

// counts: one word per level, the number of rays in that level's queue (level 0 has none: it runs
// over all level_rays). queues: level l's list at l * level_rays.
template <bool STATS>
__global__ __launch_bounds__(64, RT_WALK_WAVES) void k_wf_bounce(SceneView sv, DevRenderParams p, LightProj lp, RayTree tr,
                                                    uint32_t level, const uint32_t* __restrict__ tile_list,
                                                    uint32_t* __restrict__ counter, const float4* __restrict__ samp,
                                                    const uint32_t* __restrict__ state, float4* dsamp,
                                                    uint32_t* dstate, uint32_t* __restrict__ dkeys,
                                                    uint32_t* queues, uint32_t* counts,
                                                    DevStats* __restrict__ stats) {
    __shared__ WalkLds s_stack;
    __shared__ uint32_t s_q[RT_QUEUE_LDS + 1];
    static_assert(sizeof(WalkLds) + sizeof(s_q) <= 4 * 1280, "a walk's LDS has to leave room for 32 waves a CU");
    if (threadIdx.x == 0) s_q[RT_QUEUE_LDS] = 0u;
    LaneStats st;
    const bool last = level + 1 >= tr.levels;
    const uint32_t n = level ? counts[level] : tr.level_rays;
    const TreeSource src{sv, p, lp, tr, level, tile_list, samp, state, dsamp, dstate, dkeys,
                         queues + (size_t)level * tr.level_rays,
                         last ? nullptr : queues + (size_t)(level + 1) * tr.level_rays,
                         last ? nullptr : counts + level + 1, s_q};
#ifdef RT_DEBUG_CLOCK
    const long long c0 = clock64(), w0 = wall_clock64();
#endif
    persistent_walk<false, STATS>(sv, n, counter, src, s_stack, st);
#ifdef RT_DEBUG_CLOCK
    if (blockIdx.x % 1024u == 7u && threadIdx.x == 0 && n > 1000000u)
        printf("k_wf_bounce block %u of %u: start %lld end %lld wall ticks (100 MHz), %lld shader ticks\n", blockIdx.x, gridDim.x, w0 % 100000000ll, wall_clock64() % 100000000ll, clock64() - c0);
#endif
    if (STATS) flush_stats(st, stats);
}

// The rays of a level that k_wf_bounce has left undecided (RT_ST_REDO).
__global__ __launch_bounds__(64) void k_wf_bounce_redo(SceneView sv, DevRenderParams p, LightProj lp, RayTree tr,
                                                       uint32_t level, const uint32_t* __restrict__ tile_list,
                                                       const float4* __restrict__ samp, const uint32_t* __restrict__ state,
                                                       float4* dsamp, uint32_t* dstate, uint32_t* __restrict__ dkeys,
                                                       uint32_t* queues, uint32_t* counts) {
    __shared__ uint32_t s_q[RT_QUEUE_LDS + 1];
    if (threadIdx.x == 0) s_q[RT_QUEUE_LDS] = 0u;
    const bool last = level + 1 >= tr.levels;
    const uint32_t n = level ? counts[level] : tr.level_rays;
    const TreeSource src{sv, p, lp, tr, level, tile_list, samp, state, dsamp, dstate, dkeys,
                         queues + (size_t)level * tr.level_rays,
                         last ? nullptr : queues + (size_t)(level + 1) * tr.level_rays,
                         last ? nullptr : counts + level + 1, s_q};
    redo_marked(sv, n, src);
}

// Sun light at a stored hit, src/main.rs:176-241, from its record.
__device__ __forceinline__ Col record_light(const SceneView& sv, rt_f3 sun, float4 rec, uint32_t stt) {
    if (!(stt & RT_ST_LIT) || (stt & RT_ST_OCC)) return Col{0.0f, 0.0f, 0.0f};
    const float* td = sv.tri_data + 6 * (size_t)__float_as_uint(rec.w);
    const rt_f3 sun_dir = g_normalize(sun - f3(rec.x, rec.y, rec.z));
    return cmul(Col{td[3], td[4], td[5]}, g_dot(ld3(td), sun_dir) * sv.objs[stt >> 8].albedo);
}

// Colour a branch of the secondary-ray tree returns to the camera ray's hit: mirror_chain on the
// stored records, folded back in the reference's order (color = direct + col * reflectivity).
__device__ __forceinline__ Col branch_color(const SceneView& sv, const DevRenderParams& p, const RayTree& tr, rt_f3 sun,
                                            const float4* __restrict__ dsamp, const uint32_t* __restrict__ dstate,
                                            size_t j) {
    Col direct[RT_MAX_REFLECT];
    float refl[RT_MAX_REFLECT];
    uint32_t n = 0;
    Col tail;
    for (uint32_t l = 0;; ++l) {
        const size_t i = (size_t)l * tr.level_rays + j;
        const uint32_t st2 = dstate[i];
        if (!(st2 & RT_ST_HIT)) {
            tail = sky();
            break;
        }
        const Col c = record_light(sv, sun, dsamp[i], st2);
        const float r = sv.objs[st2 >> 8].reflectivity;
        if (l + 1 < tr.levels && r > 0.0f) {  // depth l + 2 < reflect_depth: the bounce was traced
            direct[n] = c;
            refl[n] = r;
            ++n;
            continue;
        }
        tail = c;
        break;
    }
    while (n > 0) {
        --n;
        tail = cadd(direct[n], cmul(tail, refl[n]));
    }
    return tail;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_wf_resolve_tree(SceneView sv, DevRenderParams p, RayTree tr,
                                                           const uint32_t* __restrict__ tile_list,
                                                           const float4* __restrict__ samp,
                                                           const uint32_t* __restrict__ state,
                                                           const float4* __restrict__ dsamp,
                                                           const uint32_t* __restrict__ dstate,
                                                           uint8_t* __restrict__ rgba8,
                                                           uint32_t* __restrict__ hit_ids) {
    __shared__ float s_col[3 * BLOCK];
    const PixelMap<BLOCK> pm(p, tile_list);
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    const float4 a = samp[slot];
    const uint32_t stt = state[slot];
    const uint32_t tri = __float_as_uint(a.w);
    const bool hit_any = (stt & RT_ST_HIT) != 0u;
    const uint32_t spp = p.spp;
    Col color{0.0f, 0.0f, 0.0f};
    if (hit_any) {
        // The operations of k_render<2> on the stored hit points, in its order.
        const float* td = sv.tri_data + 6 * (size_t)tri;
        const rt_f3 normal = ld3(td);
        const Col tcol = Col{td[3], td[4], td[5]};
        const rt_f3 sun = f3(p.sun[0], p.sun[1], p.sun[2]);
        const DevObject& ob = sv.objs[stt >> 8];
        color = record_light(sv, sun, a, stt);
        const size_t j0 = (size_t)slot * tr.branches;
        Col reflect_color{0.0f, 0.0f, 0.0f};
        if (tr.mirror && ob.reflectivity > 0.0f)
            reflect_color = cmul(branch_color(sv, p, tr, sun, dsamp, dstate, j0 + tr.scatt), ob.reflectivity);
        const uint64_t sample_index = ((uint64_t)pm.y * p.width + pm.x) * spp + pm.s;
        rt_f3 tang1, tang2;
        tangent_vectors(normal, &tang1, &tang2);
        Col diffuse_color{0.0f, 0.0f, 0.0f};
        for (uint32_t k = 0; k < tr.scatt; ++k) {
            const rt_f3 vel = diffuse_dir(p.seed, sample_index, k, normal, tang1, tang2);
            const float light_power = g_dot(normal, vel);
            const Col col = branch_color(sv, p, tr, sun, dsamp, dstate, j0 + k);
            diffuse_color = cadd(diffuse_color, cmul(cmul_col(col, tcol), light_power * ob.roughness));
        }
        diffuse_color = cdiv(diffuse_color, (float)p.diffuse_scatt);
        color = cadd(color, reflect_color);
        color = cadd(color, diffuse_color);
        color = gamma22(color);
    } else {
        color = pm.inside ? sky() : Col{0.0f, 0.0f, 0.0f};
    }
    if (hit_ids && pm.inside) hit_ids[((uint64_t)pm.y * p.width + pm.x) * spp + pm.s] = tri;
    s_col[3 * threadIdx.x + 0] = color.r;
    s_col[3 * threadIdx.x + 1] = color.g;
    s_col[3 * threadIdx.x + 2] = color.b;
    __syncthreads();
    if (pm.s == 0 && pm.inside) {
        Col acc = color;
        for (uint32_t k = 1; k < spp; ++k) {
            const float* c = &s_col[3 * (threadIdx.x + k)];
            acc = cadd(acc, Col{c[0], c[1], c[2]});
        }
        acc = cdiv(acc, (float)spp);
        const uint32_t px = to_u8(acc.r) | (to_u8(acc.g) << 8) | (to_u8(acc.b) << 16) | 0xff000000u;
        size_t off;
        if (p.compact) off = ((size_t)pm.local_tile * RT_TILE * RT_TILE + (size_t)pm.my * RT_TILE + pm.mx) * 4;
        else off = ((size_t)pm.y * p.width + pm.x) * 4;
        *reinterpret_cast<uint32_t*>(rgba8 + off) = px;
    }
}

__global__ void k_deinterleave(const uint8_t* __restrict__ gathered, const uint32_t* __restrict__ tile_slot,
                               uint32_t slab_tiles, uint32_t width, uint32_t height, uint32_t tiles_x,
                               uint32_t shard_count, uint8_t* __restrict__ rgba8) {
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t shard = (tx + ty) % shard_count;
    const uint32_t slot = tile_slot[tile];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(gathered) +
                          ((size_t)shard * slab_tiles + slot) * RT_TILE * RT_TILE;
    uint32_t* dst = reinterpret_cast<uint32_t*>(rgba8);
    for (uint32_t i = threadIdx.x; i < RT_TILE * RT_TILE; i += blockDim.x) {
        const uint32_t lx = i % RT_TILE, ly = i / RT_TILE;
        const uint32_t x = tx * RT_TILE + lx, y = ty * RT_TILE + ly;
        if (x < width && y < height) dst[(size_t)y * width + x] = src[i];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
namespace rtk {

static inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// Waves a persistent launch starts: every wave slot of the device (256 CUs x 32), or fewer for a
// small batch.
static inline unsigned persistent_grid(uint64_t n) { return (unsigned)std::min<uint64_t>(RT_WALK_MAX_GRID, (n + 63) / 64); }

hipError_t trace_closest(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out, uint32_t* d_counter,
                         char how, hipStream_t s) {
    if (n == 0) return hipSuccess;
    // how = 'l': one ray per lane to the end; 'w': wave-coherent walks; otherwise the persistent per-lane walk.
    // (the first two leave the rare ray whose answer hangs on the reference's order to k_trace_closest_redo)
    const bool fits = n < (1ull << 32) - RT_WALK_HEADROOM;
    const int walk = how == 'l' || !fits ? 1 : (how == 'w' ? 2 : 0);
    if (walk == 0 && d_counter)
        hipLaunchKernelGGL(k_trace_closest_persistent, dim3(persistent_grid(n)), dim3(64), 0, s, sv, d_rays, (uint32_t)n,
                           d_counter, d_out);
    else if (walk == 2)
        hipLaunchKernelGGL(k_trace_closest_wave, dim3(grid_for(n, 64)), dim3(64), (size_t)RT_BEAM_LEVELS * 64 * 8, s, sv,
                           d_rays, n, (uint32_t)RT_BEAM_LEVELS, d_out);
    else {
        hipLaunchKernelGGL(k_trace_closest, dim3(grid_for(n, RT_BLOCK)), dim3(RT_BLOCK), 0, s, sv, d_rays, n, d_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_trace_closest_redo, dim3(std::min(grid_for(n, 64), 4096u)), dim3(64), 0, s, sv, d_rays, (uint32_t)n, d_out);
    return hipGetLastError();
}

hipError_t trace_any(const SceneView& sv, const rt_ray* d_rays, uint64_t n, uint8_t* d_occ, uint32_t* d_counter,
                     char how, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const bool lane_walk = how == 'l';
    if (!lane_walk && d_counter && n < (1ull << 32) - RT_WALK_HEADROOM)
        hipLaunchKernelGGL(k_trace_any_persistent, dim3(persistent_grid(n)), dim3(64), 0, s, sv, d_rays, (uint32_t)n,
                           d_counter, d_occ);
    else
        hipLaunchKernelGGL(k_trace_any, dim3(grid_for(n, RT_BLOCK)), dim3(RT_BLOCK), 0, s, sv, d_rays, n, d_occ);
    return hipGetLastError();
}

hipError_t trace_closest_linear(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                                hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_trace_closest_linear, dim3(grid_for(n, RT_BLOCK)), dim3(RT_BLOCK), 0, s, sv, d_rays, n,
                       d_out);
    return hipGetLastError();
}

hipError_t pack_tris(const float* d_verts, const uint32_t* d_tris, uint32_t n, float* d_tri9, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_tris, dim3(grid_for(n, 256)), dim3(256), 0, s, d_verts, d_tris, n, d_tri9);
    return hipGetLastError();
}

hipError_t leaf_bounds(const float* d_verts, const uint32_t* d_tris, const uint32_t* d_leaf_first, uint32_t n_leaves,
                       float* d_leaf_bounds, float4* d_leaf_gate, hipStream_t s) {
    if (n_leaves == 0) return hipSuccess;
    hipLaunchKernelGGL(k_leaf_bounds, dim3(grid_for(n_leaves, 256)), dim3(256), 0, s, d_verts, d_tris, d_leaf_first,
                       n_leaves, d_leaf_bounds, d_leaf_gate);
    return hipGetLastError();
}

hipError_t quantize_nodes(const rtbvh::Node4* d_nodes, uint32_t n, float4* d_qnodes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize_nodes, dim3(grid_for(n, 256)), dim3(256), 0, s, d_nodes, n, d_qnodes);
    return hipGetLastError();
}

hipError_t gather_records(const float* d_tri9, const uint32_t* d_order, uint32_t n, float4* d_recs, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_records, dim3(grid_for(n, 256)), dim3(256), 0, s, d_tri9, d_order, n, d_recs);
    return hipGetLastError();
}

hipError_t render(const SceneView& sv, const DevRenderParams& p, const uint32_t* d_tile_list, int mode,
                  uint8_t* d_rgba8, uint32_t* d_hit_ids, DevStats* d_stats, hipStream_t s) {
    const uint32_t block = mode == 0 ? RT_BLOCK0 : RT_BLOCK;
    const uint32_t ppb = block / p.spp;
    const uint32_t blocks = p.n_tiles * ((RT_TILE * RT_TILE) / ppb);
    if (blocks == 0) return hipSuccess;
    const size_t lds = (mode > 0 ? RT_STACK_LDS * RT_BLOCK * 8 : 0) + (size_t)block * 12 +
                       (size_t)(block / 64) * p.beam_levels * 64 * 8;
#define RT_LAUNCH(M, S, B)                                                                                        \
    hipLaunchKernelGGL((k_render<M, S, B>), dim3(blocks), dim3(B), lds, s, sv, p, d_tile_list, d_rgba8, d_hit_ids, \
                       d_stats)
    if (d_stats) {
        if (mode == 0) RT_LAUNCH(0, true, RT_BLOCK0);
        else if (mode == 1) RT_LAUNCH(1, true, RT_BLOCK);
        else RT_LAUNCH(2, true, RT_BLOCK);
    } else {
        if (mode == 0) RT_LAUNCH(0, false, RT_BLOCK0);
        else if (mode == 1) RT_LAUNCH(1, false, RT_BLOCK);
        else RT_LAUNCH(2, false, RT_BLOCK);
    }
#undef RT_LAUNCH
    return hipGetLastError();
}

hipError_t wf_primary(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const uint32_t* d_tile_list,
                      float4* d_samp, uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats,
                      hipStream_t s) {
    const uint32_t blocks = p.n_tiles * ((RT_TILE * RT_TILE) / (RT_WF_BLOCK / p.spp));
    if (blocks == 0) return hipSuccess;
    const size_t lds = (size_t)(RT_WF_BLOCK / 64) * p.beam_levels * 64 * 8;
    if (d_stats)
        hipLaunchKernelGGL((k_wf_primary<true, RT_WF_BLOCK>), dim3(xcd_grid(blocks)), dim3(RT_WF_BLOCK), lds, s, sv, p, lp, d_tile_list,
                           d_samp, d_state, d_keys, d_stats);
    else
        hipLaunchKernelGGL((k_wf_primary<false, RT_WF_BLOCK>), dim3(xcd_grid(blocks)), dim3(RT_WF_BLOCK), lds, s, sv, p, lp, d_tile_list,
                           d_samp, d_state, d_keys, d_stats);
    hipLaunchKernelGGL(k_wf_primary_redo, dim3(std::min((blocks * RT_WF_BLOCK + 255u) / 256u, 8192u)), dim3(64), 0, s, sv, p, lp,
                       d_tile_list, blocks * RT_WF_BLOCK, d_samp, d_state, d_keys);
    return hipGetLastError();
}

hipError_t raster_depth(const SceneView& sv, const DevRenderParams& p, uint32_t n_tris_total, float* d_table,
                        unsigned long long* d_zbuf, bool clear, DevStats* d_stats, hipStream_t s) {
    if (clear) {
        hipError_t e = hipMemsetAsync(d_zbuf, 0xFF, (size_t)p.width * p.height * p.spp * 8, s);
        if (e != hipSuccess) return e;
    }
    if (n_tris_total == 0) return hipSuccess;
    const uint32_t lines = (p.width + p.height) * p.grid;
    hipLaunchKernelGGL(k_camera_table, dim3((lines + 255) / 256), dim3(256), 0, s, sv, p, d_table);
    const dim3 grid((n_tris_total + RT_RASTER_TRIS - 1) / RT_RASTER_TRIS);
    if (d_stats) hipLaunchKernelGGL(k_raster<true>, grid, dim3(64), 0, s, sv, p, n_tris_total, d_table, d_zbuf, d_stats);
    else hipLaunchKernelGGL(k_raster<false>, grid, dim3(64), 0, s, sv, p, n_tris_total, d_table, d_zbuf, d_stats);
    return hipGetLastError();
}

hipError_t raster_bin(const SceneView& sv, const DevRenderParams& p, uint32_t n_tris_total, float* d_table,
                      uint32_t n_raster_tiles, uint32_t* d_counts, uint32_t* d_offsets, uint32_t* d_list, uint32_t capacity,
                      float4* d_recs, bool fill, hipStream_t s) {
    if (n_tris_total == 0) return hipSuccess;
    const dim3 grid((n_tris_total + 255u) / 256u);
#ifdef RT_DEBUG_RASTER
    {
        const char *a = std::getenv("RAYS_HIP_DBG_TRI"), *b = std::getenv("RAYS_HIP_DBG_SU"), *c = std::getenv("RAYS_HIP_DBG_SW");
        const uint32_t h[4] = {a ? (uint32_t)std::atoll(a) : 0xFFFFFFFFu, b ? (uint32_t)std::atoll(b) : 0u, c ? (uint32_t)std::atoll(c) : 0u, 0u};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), h, sizeof(h));
    }
#endif
    if (!fill) {
        const uint32_t lines = (p.width + p.height) * p.grid;
        hipLaunchKernelGGL(k_camera_table, dim3((lines + 255) / 256), dim3(256), 0, s, sv, p, d_table);
        hipError_t e = hipMemsetAsync(d_counts, 0, (size_t)n_raster_tiles * 4, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_raster_bin<false>, grid, dim3(256), 0, s, sv, p, n_tris_total, d_table, d_counts, d_offsets, d_list, capacity, d_recs);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, d_counts, n_raster_tiles, d_offsets);
    } else {
        hipLaunchKernelGGL(k_raster_bin<true>, grid, dim3(256), 0, s, sv, p, n_tris_total, d_table, d_counts, d_offsets, d_list, capacity, d_recs);
    }
    return hipGetLastError();
}

hipError_t raster_tiles(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const uint32_t* d_tile_list,
                        const float* d_table, const uint32_t* d_offsets, const uint32_t* d_list, const float4* d_recs,
                        float4* d_samp, uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats, hipStream_t s) {
    const uint32_t waves = p.n_tiles * p.spp;   // g x g raster tiles of 32 x 32 samples per tile of 32 x 32 pixels
    if (waves == 0) return hipSuccess;
    if (d_stats)
        hipLaunchKernelGGL(k_raster_tile<true>, dim3(waves), dim3(64 * RT_RTILE_GROUP), 0, s, sv, p, lp, d_tile_list, d_table, d_offsets, d_list,
                           d_recs, d_samp, d_state, d_keys, d_stats);
    else
        hipLaunchKernelGGL(k_raster_tile<false>, dim3(waves), dim3(64 * RT_RTILE_GROUP), 0, s, sv, p, lp, d_tile_list, d_table, d_offsets, d_list,
                           d_recs, d_samp, d_state, d_keys, d_stats);
    return hipGetLastError();
}

hipError_t raster_records(const SceneView& sv, const DevRenderParams& p, const LightProj& lp,
                          const uint32_t* d_tile_list, unsigned long long* d_zbuf, float4* d_samp,
                          uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats, hipStream_t s) {
    const uint32_t blocks = p.n_tiles * ((RT_TILE * RT_TILE) / (RT_BLOCK / p.spp));
    if (blocks == 0) return hipSuccess;
    if (d_stats)
        hipLaunchKernelGGL((k_raster_records<true, RT_BLOCK>), dim3(blocks), dim3(RT_BLOCK), 0, s, sv, p, lp, d_tile_list,
                           d_zbuf, d_samp, d_state, d_keys, d_stats);
    else
        hipLaunchKernelGGL((k_raster_records<false, RT_BLOCK>), dim3(blocks), dim3(RT_BLOCK), 0, s, sv, p, lp, d_tile_list,
                           d_zbuf, d_samp, d_state, d_keys, d_stats);
    hipLaunchKernelGGL(k_wf_primary_redo, dim3(std::min((blocks * RT_BLOCK + 255u) / 256u, 8192u)), dim3(64), 0, s, sv, p, lp,
                       d_tile_list, blocks * RT_BLOCK, d_samp, d_state, d_keys);
    return hipGetLastError();
}

hipError_t wf_shadow(const SceneView& sv, const DevRenderParams& p, uint32_t n, uint32_t cell_shift,
                     const uint32_t* d_keys, const uint32_t* d_vals, const float4* d_samp, uint32_t* d_state,
                     DevStats* d_stats, hipStream_t s, const uint32_t* d_n_valid, const uint32_t* d_done_flag) {
    if (n == 0) return hipSuccess;
    const size_t lds = (size_t)(RT_WF_BLOCK / 64) * p.beam_levels * 64 * 8;
    const uint32_t n_blocks = xcd_grid(grid_for(n, RT_WF_BLOCK));
    const uint32_t grid = d_done_flag ? std::min(n_blocks, 8192u) : n_blocks;
    if (d_stats)
        hipLaunchKernelGGL((k_wf_shadow<true>), dim3(grid), dim3(RT_WF_BLOCK), lds, s, sv, p, n, cell_shift, d_keys,
                           d_vals, d_samp, d_state, d_stats, d_n_valid, d_done_flag, n_blocks);
    else
        hipLaunchKernelGGL((k_wf_shadow<false>), dim3(grid), dim3(RT_WF_BLOCK), lds, s, sv, p, n, cell_shift, d_keys,
                           d_vals, d_samp, d_state, d_stats, d_n_valid, d_done_flag, n_blocks);
    return hipGetLastError();
}

static LightCells light_cells(const SceneView& sv, const LightProj& lp) {
    LightCells lc{};
    lc.shift = 16u - lp.key_bits / 2u;
    lc.n_cells = 1u << lp.key_bits;
    // 512 u x the diagonal of the scene's bounds (at most sqrt(3) x its largest side)
    lc.margin = 512.0f * 5.9604645e-8f * 1.7320508f * sv.scene_extent * 1.001f;
    lc.slope_u = 1.0f + std::fabs(lp.uc) + lp.uabs;
    lc.slope_v = 1.0f + std::fabs(lp.vc) + lp.vabs;
    lc.away = lp.c_w > lp.sun_w ? 1.0f : -1.0f;
    lc.depth_span = sv.scene_extent;
    return lc;
}

hipError_t lcell_lists(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, uint32_t n_tris_total,
                       uint32_t* d_counts, float* d_lists, uint32_t cap_shift, uint32_t max_mean,
                       uint32_t* d_flags, uint32_t* d_redo, bool redo_all, hipStream_t s) {
    if (lp.key_bits < 2 || lp.key_bits > 24 || (lp.key_bits & 1u) || sv.n_objs > RT_LCELL_MAX_OBJS ||
        n_tris_total >= (1u << RT_LCELL_ID_BITS) || lp.key_bits + cap_shift > 31u)
        return hipErrorInvalidValue;
    const LightCells lc = light_cells(sv, lp);
    hipError_t e = hipMemsetAsync(d_counts, 0, (size_t)lc.n_cells * 4, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_flags, 0, 32, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lcell_bin, dim3(grid_for(n_tris_total, 256)), dim3(256), 0, s, sv, p, lp, lc, n_tris_total, d_counts,
                       d_lists, cap_shift);
    hipLaunchKernelGGL(k_lcell_check, dim3(std::min(64u, grid_for(lc.n_cells, 256))), dim3(256), 0, s, d_counts, lc.n_cells,
                       1u << cap_shift, max_mean, d_flags, CellsTail{sv.leaf_gate, d_redo, redo_all ? 1u : 0u});
    return hipGetLastError();
}

hipError_t shadow_cells(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, uint32_t n,
                        const uint32_t* d_keys, const uint32_t* d_vals, const float4* d_samp, uint32_t* d_state,
                        DevStats* d_stats, const uint32_t* d_n_valid, const uint32_t* d_counts, const float* d_lists,
                        uint32_t cap_shift, uint32_t* d_flags, uint32_t* d_redo, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = xcd_grid(grid_for(n, 64));
#define RT_SC_LAUNCH(ST, ONE)                                                                                          \
    hipLaunchKernelGGL((k_shadow_cells<ST, ONE>), dim3(grid), dim3(64), 0, s, sv, p, lp, d_keys, d_vals, d_samp, d_state, \
                       d_stats, d_n_valid, d_counts, d_lists, cap_shift, d_flags)
    if (d_stats) {
        if (sv.n_objs == 1) RT_SC_LAUNCH(true, true);
        else RT_SC_LAUNCH(true, false);
    } else {
        if (sv.n_objs == 1) RT_SC_LAUNCH(false, true);
        else RT_SC_LAUNCH(false, false);
    }
#undef RT_SC_LAUNCH
    // the rays whose occluder the reference would not have tested (normally none)
    hipLaunchKernelGGL(k_shadow_redo, dim3(std::min(256u, grid_for(n, 64))), dim3(64), 0, s, sv, p, d_redo, d_flags, d_samp, d_state);
    return hipGetLastError();
}

hipError_t wf_resolve(const SceneView& sv, const DevRenderParams& p, const uint32_t* d_tile_list, const float4* d_samp,
                      const uint32_t* d_state, uint8_t* d_rgba8, uint32_t* d_hit_ids, hipStream_t s) {
    const uint32_t blocks = p.n_tiles * (RT_TILE * RT_TILE / (64u * RT_RESOLVE_SLOTS)) * p.spp;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wf_resolve, dim3(blocks), dim3(64), 0, s, sv, p, d_tile_list, d_samp, d_state, d_rgba8, d_hit_ids);
    return hipGetLastError();
}

hipError_t wf_bounce(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const RayTree& tr,
                     uint32_t level, const uint32_t* d_tile_list, uint32_t* d_counter, const float4* d_samp,
                     const uint32_t* d_state, float4* d_dsamp, uint32_t* d_dstate, uint32_t* d_dkeys,
                     uint32_t* d_queues, uint32_t* d_counts, DevStats* d_stats, hipStream_t s) {
    if (tr.level_rays == 0 || level >= tr.levels) return hipSuccess;
    if (tr.level_rays >= (1ull << 32) - RT_WALK_HEADROOM) return hipErrorInvalidValue;
    const dim3 grid(persistent_grid(tr.level_rays));
    if (d_stats)
        hipLaunchKernelGGL((k_wf_bounce<true>), grid, dim3(64), 0, s, sv, p, lp, tr, level, d_tile_list, d_counter,
                           d_samp, d_state, d_dsamp, d_dstate, d_dkeys, d_queues, d_counts, d_stats);
    else
        hipLaunchKernelGGL((k_wf_bounce<false>), grid, dim3(64), 0, s, sv, p, lp, tr, level, d_tile_list, d_counter,
                           d_samp, d_state, d_dsamp, d_dstate, d_dkeys, d_queues, d_counts, d_stats);
    hipLaunchKernelGGL(k_wf_bounce_redo, dim3(4096), dim3(64), 0, s, sv, p, lp, tr, level, d_tile_list, d_samp, d_state,
                       d_dsamp, d_dstate, d_dkeys, d_queues, d_counts);
    return hipGetLastError();
}

hipError_t wf_resolve_tree(const SceneView& sv, const DevRenderParams& p, const RayTree& tr,
                           const uint32_t* d_tile_list, const float4* d_samp, const uint32_t* d_state,
                           const float4* d_dsamp, const uint32_t* d_dstate, uint8_t* d_rgba8, uint32_t* d_hit_ids,
                           hipStream_t s) {
    const uint32_t blocks = p.n_tiles * ((RT_TILE * RT_TILE) / (RT_BLOCK / p.spp));
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((k_wf_resolve_tree<RT_BLOCK>), dim3(blocks), dim3(RT_BLOCK), 0, s, sv, p, tr, d_tile_list,
                       d_samp, d_state, d_dsamp, d_dstate, d_rgba8, d_hit_ids);
    return hipGetLastError();
}

hipError_t deinterleave(const uint8_t* d_gathered, const uint32_t* d_tile_slot, uint32_t slab_tiles, uint32_t width,
                        uint32_t height, uint32_t tiles_x, uint32_t n_tiles_total, uint32_t shard_count,
                        uint8_t* d_rgba8, hipStream_t s) {
    if (n_tiles_total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_deinterleave, dim3(n_tiles_total), dim3(256), 0, s, d_gathered, d_tile_slot, slab_tiles,
                       width, height, tiles_x, shard_count, d_rgba8);
    return hipGetLastError();
}

}  // namespace rtk
