// Test shim: builds both trees over the given boxes with the product's host builder and checks
// their invariants. Compiled by tests/test_host_logic.py together with raytracing_amd/csrc/bvh_build.cpp.
#include <cstdint>
#include <cstring>
#include <vector>

#include "bvh_build.h"

using rtbvh::Child;
using rtbvh::EMPTY;
using rtbvh::LEAF_BIT;

namespace {

struct Checker {
    const float* bounds;
    uint32_t n;
    const rtbvh::Result& r;
    std::vector<uint32_t> seen;
    int error = 0;

    // Exact bounds of the records of a leaf must lie inside the child's box.
    void leaf(const Child& c) {
        const uint32_t first = (c.ref & 0x7FFFFFFFu) >> 3, cnt = (c.ref & 7u) + 1u;
        if (cnt > rtbvh::MAX_LEAF) error = 10;
        for (uint32_t i = 0; i < cnt; ++i) {
            if (first + i >= n) {
                error = 11;
                return;
            }
            const uint32_t t = r.order[first + i];
            seen[first + i]++;
            const float* b = bounds + 6 * (size_t)t;
            for (int k = 0; k < 3; ++k)
                if (!(c.lo[k] <= b[k] && c.hi[k] >= b[3 + k])) error = 12;
        }
    }
    static bool inside(const Child& outer, const float lo[3], const float hi[3]) {
        for (int k = 0; k < 3; ++k)
            if (!(outer.lo[k] <= lo[k] && outer.hi[k] >= hi[k])) return false;
        return true;
    }
    // Returns the depth below; fills lo/hi with the union of the subtree's boxes.
    uint32_t walk4(uint32_t node, float lo[3], float hi[3], uint32_t* pending) {
        uint32_t deepest = 0, used = 0;
        for (int k = 0; k < 3; ++k) lo[k] = 1e30f, hi[k] = -1e30f;
        uint32_t worst = 0;
        for (int i = 0; i < 4; ++i) {
            const Child& c = r.nodes[node].c[i];
            if (c.ref == EMPTY) continue;
            ++used;
            float clo[3], chi[3];
            if (c.ref & LEAF_BIT) {
                leaf(c);
                for (int k = 0; k < 3; ++k) clo[k] = c.lo[k], chi[k] = c.hi[k];
            } else {
                if (c.ref >= r.nodes.size() || c.ref <= node) {
                    error = 20;
                    return 0;
                }
                uint32_t p = 0;
                deepest = std::max(deepest, 1 + walk4(c.ref, clo, chi, &p));
                worst = std::max(worst, p);
                if (!inside(c, clo, chi)) error = 21;
            }
            for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], clo[k]), hi[k] = std::max(hi[k], chi[k]);
        }
        if (used < 1) error = 22;
        *pending = used - 1 + worst;
        return deepest;
    }
    uint32_t walk_wide(uint32_t ref, uint32_t width, float lo[3], float hi[3]) {
        const uint32_t first = ref >> 6, cnt = (ref & 63u) + 1u;
        if (cnt > width || first + cnt > r.wide.size()) {
            error = 30;
            return 0;
        }
        uint32_t deepest = 0;
        for (int k = 0; k < 3; ++k) lo[k] = 1e30f, hi[k] = -1e30f;
        for (uint32_t i = 0; i < cnt; ++i) {
            const Child& c = r.wide[first + i];
            float clo[3], chi[3];
            if (c.ref & LEAF_BIT) {
                leaf(c);
                for (int k = 0; k < 3; ++k) clo[k] = c.lo[k], chi[k] = c.hi[k];
            } else {
                deepest = std::max(deepest, walk_wide(c.ref, width, clo, chi));
                if (!inside(c, clo, chi)) error = 31;
            }
            for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], clo[k]), hi[k] = std::max(hi[k], chi[k]);
        }
        return deepest + 1;
    }
};

}  // namespace

// out: nodes4, wide children, wide depth (reported), wide depth (walked), max_stack (reported),
// pending bound (walked), 0, 0. Returns 0 when every invariant holds.
extern "C" int bvh_check(const float* bounds, uint32_t n, uint32_t width, int threads, int balanced, uint32_t* out) {
    rtbvh::Result r;
    rtbvh::build(bounds, n, 0, 0, width, 0, threads, balanced != 0, &r);
    std::memset(out, 0, 8 * sizeof(uint32_t));
    out[0] = (uint32_t)r.nodes.size();
    out[1] = (uint32_t)r.wide.size();
    out[2] = r.wide_depth;
    out[4] = r.max_stack;
    if (n == 0) return 0;
    if (r.order.size() != n) return 1;
    {
        std::vector<uint32_t> count(n, 0);
        for (uint32_t t : r.order) {
            if (t >= n) return 2;
            count[t]++;
        }
        for (uint32_t c : count)
            if (c != 1) return 3;
    }
    Checker ck{bounds, n, r, std::vector<uint32_t>(n, 0)};
    float lo[3], hi[3];
    uint32_t pending = 0;
    ck.walk4(0, lo, hi, &pending);
    out[5] = pending + 1;
    if (ck.error) return ck.error;
    for (uint32_t s : ck.seen)
        if (s != 1) return 4;  // every record in exactly one leaf of the 4-wide tree
    std::fill(ck.seen.begin(), ck.seen.end(), 0u);
    out[3] = ck.walk_wide(r.wide_root, width, lo, hi);
    if (ck.error) return ck.error;
    for (uint32_t s : ck.seen)
        if (s != 1) return 5;  // and of the wide tree
    if (out[3] != r.wide_depth) return 6;
    if (out[5] > r.max_stack) return 7;
    return 0;
}

// The arithmetic of the wide tree's size limit: the largest child index the builders may hand out
// still leaves LEAF_BIT clear in an inner reference of a full node; the first index beyond it would not.
extern "C" int bvh_wide_limit_check(void) {
    const uint64_t last = rtbvh::MAX_WIDE_CHILDREN - 1;
    if (!rtbvh::wide_children_fit(last) || rtbvh::wide_children_fit(rtbvh::MAX_WIDE_CHILDREN)) return 1;
    const uint32_t ref = rtbvh::wide_ref((uint32_t)last, rtbvh::MAX_WIDE);
    if (ref & LEAF_BIT) return 2;
    if ((ref >> 6) != (uint32_t)last || (ref & 63u) + 1u != rtbvh::MAX_WIDE) return 3;
    if (!(rtbvh::wide_ref((uint32_t)rtbvh::MAX_WIDE_CHILDREN, 1u) & LEAF_BIT)) return 4;
    return 0;
}
