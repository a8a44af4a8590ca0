// bvh_build.cpp -- binned-SAH binary build over single triangles, then a collapse to 4-wide
// nodes laid out in depth-first order. Host code; runs once per scene.

#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rtbvh {
namespace {

constexpr int BINS = 16;
// Every triangle box is widened per axis by PAD_REL of its own largest coordinate plus PAD_ABS of
// the object's largest coordinate. The walks' slab tests and the reference's triangle test both
// round with errors proportional to the coordinates involved (origin, box planes, hit point), so
// for rays that start within about the object's own size of it -- camera rays from the origin,
// shadow and bounce rays from hit points -- the padded box is reached whenever the triangle test
// could accept; for origins much farther away than that the rounding of origin * 1 / direction
// grows with the origin and a grazing hit at a box edge can be missed (a documented edge class).
constexpr float PAD_REL = 6e-6f;
// PAD_ABS is ten times what those roundings need: the nearest-hit walks rely on reaching every triangle
// whose hit lies less than half the padding (as a distance along the ray) behind the nearest one (tie_note
// in rt_kernels.hip, which reads the same constant).
constexpr float PAD_ABS = RT_PAD_ABS;   // bvh_build.h

// Tuning knobs (environment, read once): largest leaf and the price of one traversal step in
// triangle tests.
uint32_t max_leaf() {
    static const uint32_t v = [] {
        const char* e = std::getenv("RAYS_HIP_MAX_LEAF");
        const long x = e ? std::atol(e) : (long)MAX_LEAF;
        return (uint32_t)std::min<long>(std::max<long>(x, 1), 8);
    }();
    return v;
}
// Wide collapse rule: 2 = least summed node area by dynamic programming (default), 0 = greedy,
// open the child with the largest surface area.
int collapse_rule() {
    static const int v = [] {
        const char* e = std::getenv("RAYS_HIP_COLLAPSE");
        return e ? std::atoi(e) : 2;
    }();
    return v;
}
float trav_cost() {
    static const float v = [] {
        const char* e = std::getenv("RAYS_HIP_TRAV_COST");
        return e ? (float)std::atof(e) : 1.0f;
    }();
    return v;
}

struct Prim {
    float lo[3], hi[3], c[3];
};

struct Node2 {
    float lo[3], hi[3];
    uint32_t left = 0, right = 0;   // inner: child node indices
    uint32_t first = 0, count = 0;  // leaf when count > 0: range in idx[]
};

struct Box3 {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    void grow_pt(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.0f) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

class Builder {
   public:
    Builder(std::vector<Prim>& prims, int n_threads, bool balanced)
        : prims_(prims), idx_(prims.size()), nodes_(2 * prims.size() + 1), next_(0), balanced_(balanced) {
        for (size_t i = 0; i < idx_.size(); ++i) idx_[i] = (uint32_t)i;
        int d = 0;
        while ((1 << d) < std::max(1, n_threads) * 2) ++d;
        par_depth_ = n_threads > 1 ? d : 0;
    }

    uint32_t run() { return rec(0, (uint32_t)prims_.size(), par_depth_); }
    uint32_t used() const { return next_.load(); }

    std::vector<Prim>& prims_;
    std::vector<uint32_t> idx_;
    std::vector<Node2> nodes_;

   private:
    std::atomic<uint32_t> next_;
    int par_depth_;
    bool balanced_;  // median splits only: bounds the depth at ceil(log2 n)

    uint32_t rec(uint32_t begin, uint32_t end, int par) {
        const uint32_t me = next_.fetch_add(1);
        Node2& node = nodes_[me];
        const uint32_t n = end - begin;
        Box3 bb, cb;
        for (uint32_t i = begin; i < end; ++i) {
            const Prim& p = prims_[idx_[i]];
            bb.grow(p.lo, p.hi);
            cb.grow_pt(p.c);
        }
        std::memcpy(node.lo, bb.lo, sizeof(bb.lo));
        std::memcpy(node.hi, bb.hi, sizeof(bb.hi));
        if (n == 1) {
            node.first = begin;
            node.count = 1;
            return me;
        }

        if (balanced_) {
            if (n <= max_leaf()) {
                node.first = begin;
                node.count = n;
                return me;
            }
            int k = 0;
            for (int a = 1; a < 3; ++a)
                if (cb.hi[a] - cb.lo[a] > cb.hi[k] - cb.lo[k]) k = a;
            const uint32_t mid = begin + n / 2;
            std::nth_element(idx_.begin() + begin, idx_.begin() + mid, idx_.begin() + end,
                             [&](uint32_t a, uint32_t b) { return prims_[a].c[k] < prims_[b].c[k]; });
            uint32_t l, r;
            if (par > 0 && n > 16384) {
                std::thread th([&]() { l = rec(begin, mid, par - 1); });
                r = rec(mid, end, par - 1);
                th.join();
            } else {
                l = rec(begin, mid, 0);
                r = rec(mid, end, 0);
            }
            node.left = l;
            node.right = r;
            node.count = 0;
            return me;
        }

        // Binned SAH over the three axes.
        float best_cost = FLT_MAX;
        int best_axis = -1, best_split = -1;
        float scale[3];
        for (int k = 0; k < 3; ++k) {
            float ext = cb.hi[k] - cb.lo[k];
            scale[k] = ext > 0.0f ? (float)BINS * (1.0f - 1e-6f) / ext : 0.0f;
        }
        if (scale[0] > 0.0f || scale[1] > 0.0f || scale[2] > 0.0f) {
            Box3 bins[3][BINS];
            uint32_t cnt[3][BINS];
            std::memset(cnt, 0, sizeof(cnt));
            for (uint32_t i = begin; i < end; ++i) {
                const Prim& p = prims_[idx_[i]];
                for (int k = 0; k < 3; ++k) {
                    int b = (int)((p.c[k] - cb.lo[k]) * scale[k]);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    bins[k][b].grow(p.lo, p.hi);
                    cnt[k][b]++;
                }
            }
            for (int k = 0; k < 3; ++k) {
                if (scale[k] <= 0.0f) continue;
                float right_area[BINS];
                uint32_t right_cnt[BINS];
                Box3 acc;
                uint32_t c = 0;
                for (int b = BINS - 1; b > 0; --b) {
                    if (cnt[k][b]) acc.grow(bins[k][b].lo, bins[k][b].hi);
                    c += cnt[k][b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = c;
                }
                Box3 lacc;
                uint32_t lc = 0;
                for (int b = 0; b < BINS - 1; ++b) {
                    if (cnt[k][b]) lacc.grow(bins[k][b].lo, bins[k][b].hi);
                    lc += cnt[k][b];
                    if (lc == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = lacc.half_area() * (float)lc + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = k;
                        best_split = b + 1;  // bins < best_split go left
                    }
                }
            }
        }

        const float area = bb.half_area();
        if (n <= max_leaf()) {
            // Leaf unless the split is clearly cheaper (traversal step priced at one triangle test).
            const float leaf_cost = (float)n;
            const float split_cost = (best_axis >= 0 && area > 0.0f) ? trav_cost() + best_cost / area : FLT_MAX;
            if (!(split_cost < leaf_cost)) {
                node.first = begin;
                node.count = n;
                return me;
            }
        }

        uint32_t mid;
        if (best_axis >= 0) {
            const int k = best_axis;
            const float lo = cb.lo[k], sc = scale[k];
            const int split = best_split;
            uint32_t* first = idx_.data() + begin;
            uint32_t* last = idx_.data() + end;
            uint32_t* m = std::partition(first, last, [&](uint32_t id) {
                int b = (int)((prims_[id].c[k] - lo) * sc);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                return b < split;
            });
            mid = (uint32_t)(m - idx_.data());
        } else {
            mid = begin;  // all centroids coincide
        }
        if (mid == begin || mid == end) mid = begin + n / 2;

        uint32_t l, r;
        if (par > 0 && n > 16384) {
            std::thread th([&]() { l = rec(begin, mid, par - 1); });
            r = rec(mid, end, par - 1);
            th.join();
        } else {
            l = rec(begin, mid, 0);
            r = rec(mid, end, 0);
        }
        node.left = l;
        node.right = r;
        node.count = 0;
        return me;
    }
};

struct Collapser {
    const Builder& b;
    Result& out;
    uint32_t node_base, record_base;
    std::vector<uint32_t> leaf_ref;  // per binary node: the leaf reference once emitted
    std::vector<uint32_t> n_leaves;  // per binary node: leaves below it

    // Optimal wide collapse (collapse rule 2). A wave visits a wide node at a cost that does not
    // depend on how many children it has (one lane tests one child), so the expected work of a walk
    // is the summed surface area of the wide nodes. For every binary node n and slot budget k,
    // g(n, k) = least summed area of the wide nodes needed below n when n's subtree is handed to
    // its parent as at most k children: either n itself becomes one child and a wide node of its
    // own (f(n) = area(n) + best split of `width` slots over its two sides), or the k slots are
    // split between its two sides. Children are created after their parents, so a reverse sweep
    // over the node array is a post-order.
    struct Plan {
        std::vector<uint32_t> dp_off;
        std::vector<uint8_t> dp_m, dp_choice, dp_root_split;
        std::vector<float> dp_g;
    };
    Plan plan_wide, plan_4;

    void plan(uint32_t width, Plan& P) {
        std::vector<uint32_t>& dp_off = P.dp_off;
        std::vector<uint8_t>&dp_m = P.dp_m, &dp_choice = P.dp_choice, &dp_root_split = P.dp_root_split;
        std::vector<float>& dp_g = P.dp_g;
        const uint32_t nn = b.used();
        n_leaves.assign(b.nodes_.size(), 0);
        dp_off.assign(nn + 1, 0);
        dp_m.assign(nn, 0);
        dp_root_split.assign(nn, 0);
        for (uint32_t i = nn; i-- > 0;) {
            const Node2& n = b.nodes_[i];
            n_leaves[i] = n.count > 0 ? 1u : n_leaves[n.left] + n_leaves[n.right];
            dp_m[i] = (uint8_t)std::min(width, n_leaves[i]);
        }
        for (uint32_t i = 0; i < nn; ++i) dp_off[i + 1] = dp_off[i] + dp_m[i];
        dp_g.assign(dp_off[nn], 0.0f);
        dp_choice.assign(dp_off[nn], 0);
        for (uint32_t i = nn; i-- > 0;) {
            const Node2& n = b.nodes_[i];
            if (n.count > 0) continue;  // a leaf: g = 0, one slot
            const uint32_t L = n.left, R = n.right, mL = dp_m[L], mR = dp_m[R], m = dp_m[i];
            const float* gL = &dp_g[dp_off[L]];
            const float* gR = &dp_g[dp_off[R]];
            float* g = &dp_g[dp_off[i]];
            uint8_t* ch = &dp_choice[dp_off[i]];
            for (uint32_t k = 2; k <= m; ++k) {
                float best = FLT_MAX;
                uint32_t arg = 1;
                const uint32_t lo = k > mR ? k - mR : 1u, hi = std::min(k - 1, mL);
                for (uint32_t a = lo; a <= hi; ++a) {
                    const float c = gL[a - 1] + gR[k - a - 1];
                    if (c < best) {
                        best = c;
                        arg = a;
                    }
                }
                g[k - 1] = best;
                ch[k - 1] = (uint8_t)arg;
            }
            const float f = half_area(n) + g[m - 1];  // m >= 2 for an inner node
            dp_root_split[i] = ch[m - 1];
            g[0] = f;
            ch[0] = 0;
            for (uint32_t k = 2; k <= m; ++k)
                if (f <= g[k - 1]) {
                    g[k - 1] = f;
                    ch[k - 1] = 0;
                }
        }
    }

    void collect(const Plan& P, uint32_t c, uint32_t k, std::vector<uint32_t>& kids) const {
        const Node2& n = b.nodes_[c];
        k = std::min<uint32_t>(k, P.dp_m[c]);
        const uint32_t a = n.count > 0 ? 0u : P.dp_choice[P.dp_off[c] + k - 1];
        if (a == 0) {
            kids.push_back(c);
            return;
        }
        collect(P, n.left, a, kids);
        collect(P, n.right, k - a, kids);
    }
    // Children of the wide (or 4-wide) node made of binary node root2, by the plan.
    void planned_kids(Plan& P, uint32_t width, uint32_t root2, std::vector<uint32_t>& kids) {
        if (P.dp_m.empty()) plan(width, P);
        const uint32_t a = P.dp_root_split[root2];
        collect(P, b.nodes_[root2].left, a, kids);
        collect(P, b.nodes_[root2].right, P.dp_m[root2] - a, kids);
    }

    uint32_t emit_leaf(const Node2& n) {
        const uint32_t first = (uint32_t)out.order.size();
        for (uint32_t i = 0; i < n.count; ++i) out.order.push_back(b.idx_[n.first + i]);
        const uint32_t ref = LEAF_BIT | ((record_base + first) << 3) | (n.count - 1);
        leaf_ref[(size_t)(&n - b.nodes_.data())] = ref;
        return ref;
    }

    // Wide collapse of binary inner node `root2` into one node of at most `width` children, by the
    // plan of plan() or by opening the largest inner child until `width` children are reached.
    // Returns the node's reference.
    uint32_t emit_wide(uint32_t root2, uint32_t width, uint32_t wide_base, uint32_t depth) {
        out.wide_depth = std::max(out.wide_depth, depth);
        std::vector<uint32_t> kids;
        if (collapse_rule() == 2) {
            planned_kids(plan_wide, width, root2, kids);
        } else {
            kids.push_back(b.nodes_[root2].left);
            kids.push_back(b.nodes_[root2].right);
        }
        while (collapse_rule() != 2 && kids.size() < width) {
            int pick = -1;
            float best = -1.0f;
            for (size_t i = 0; i < kids.size(); ++i) {
                const Node2& c = b.nodes_[kids[i]];
                if (c.count == 0) {
                    const float a = half_area(c);
                    if (a > best) {
                        best = a;
                        pick = (int)i;
                    }
                }
            }
            if (pick < 0) break;
            const Node2& c = b.nodes_[kids[(size_t)pick]];
            kids[(size_t)pick] = c.left;
            kids.push_back(c.right);
        }
        // Children in order of decreasing box area: the any-hit walk visits pending children in
        // lane order, and a large child is the likelier occluder (measured: 6 % fewer shadow-ray
        // triangle tests than in collapse order; the nearest-hit walk orders by distance itself).
        std::stable_sort(kids.begin(), kids.end(), [&](uint32_t x, uint32_t y) {
            return half_area(b.nodes_[x]) > half_area(b.nodes_[y]);
        });
        const uint32_t first = (uint32_t)out.wide.size();
        out.wide.resize(first + kids.size());
        for (size_t i = 0; i < kids.size(); ++i) {
            const Node2& c = b.nodes_[kids[i]];
            Child ch;
            for (int k = 0; k < 3; ++k) {
                ch.lo[k] = c.lo[k];
                ch.hi[k] = c.hi[k];
            }
            ch.pad = 0;
            ch.ref = c.count > 0 ? leaf_ref[kids[i]] : emit_wide(kids[i], width, wide_base, depth + 1);
            out.wide[first + i] = ch;
        }
        return wide_ref(wide_base + first, (uint32_t)kids.size());
    }

    static float half_area(const Node2& n) {
        float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    // Emits the 4-wide node for binary node `root2` (an inner node) and returns the pending-entry
    // bound of its subtree.
    uint32_t emit_inner(uint32_t root2, uint32_t* my_index) {
        const uint32_t me = (uint32_t)out.nodes.size();
        *my_index = me;
        out.nodes.emplace_back();
        uint32_t kids[4];
        int nk = 0;
        static const bool plan4 = [] {
            const char* e = std::getenv("RAYS_HIP_COLLAPSE4");
            return e ? std::atoi(e) == 2 : true;
        }();
        if (plan4) {
            // The same least-area plan as for the wide tree: a lane's visit of a 4-wide node costs
            // the same whatever its child count.
            std::vector<uint32_t> pk;
            planned_kids(plan_4, 4, root2, pk);
            for (uint32_t k : pk) kids[nk++] = k;
        } else {
            kids[nk++] = b.nodes_[root2].left;
            kids[nk++] = b.nodes_[root2].right;
        }
        // Open the largest inner child until four slots are used.
        while (!plan4 && nk < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int i = 0; i < nk; ++i) {
                const Node2& c = b.nodes_[kids[i]];
                if (c.count == 0) {
                    float a = half_area(c);
                    if (a > best) {
                        best = a;
                        pick = i;
                    }
                }
            }
            if (pick < 0) break;
            const Node2& c = b.nodes_[kids[pick]];
            kids[pick] = c.left;
            kids[nk++] = c.right;
        }
        uint32_t deepest = 0;
        Node4 n4;
        std::memset(&n4, 0, sizeof(n4));
        for (int i = 0; i < 4; ++i) {
            if (i < nk) {
                const Node2& c = b.nodes_[kids[i]];
                for (int k = 0; k < 3; ++k) {
                    n4.c[i].lo[k] = c.lo[k];
                    n4.c[i].hi[k] = c.hi[k];
                }
                if (c.count > 0) {
                    n4.c[i].ref = emit_leaf(c);
                } else {
                    uint32_t ci;
                    uint32_t d = emit_inner(kids[i], &ci);
                    n4.c[i].ref = node_base + ci;
                    deepest = std::max(deepest, d);
                }
            } else {
                // An inverted box: every slab test on it fails for finite rays; the kernels also
                // check the reference itself.
                for (int k = 0; k < 3; ++k) {
                    n4.c[i].lo[k] = INFINITY;
                    n4.c[i].hi[k] = -INFINITY;
                }
                n4.c[i].ref = EMPTY;
            }
        }
        out.nodes[me] = n4;
        return (uint32_t)(nk - 1) + deepest;
    }
};

}  // namespace

void build(const float* tri_bounds, uint32_t n, uint32_t node_index_base, uint32_t record_base,
           uint32_t wide_width, uint32_t wide_base, int n_threads, bool balanced, Result* out) {
    out->nodes.clear();
    out->order.clear();
    out->wide.clear();
    out->max_stack = 0;
    out->wide_root = 0;
    out->wide_depth = 0;
    if (n == 0) return;
    wide_width = std::min(std::max(wide_width, 2u), MAX_WIDE);
    std::vector<Prim> prims(n);
    float largest = 0.0f;
    for (size_t i = 0; i < 6 * (size_t)n; ++i) largest = std::max(largest, std::fabs(tri_bounds[i]));
    const float abs_pad = largest * PAD_ABS + 1e-30f;
    for (uint32_t i = 0; i < n; ++i) {
        const float* b = tri_bounds + 6 * (size_t)i;
        Prim& p = prims[i];
        for (int k = 0; k < 3; ++k) {
            float lo = b[k], hi = b[3 + k];
            float pad = std::max(std::fabs(lo), std::fabs(hi)) * PAD_REL + abs_pad;
            p.lo[k] = lo - pad;
            p.hi[k] = hi + pad;
            p.c[k] = 0.5f * lo + 0.5f * hi;
        }
    }
    Builder b(prims, n_threads, balanced);
    const uint32_t root = b.run();
    out->nodes.reserve(n / 2 + 8);
    out->order.reserve(n);
    Collapser c{b, *out, node_index_base, record_base, std::vector<uint32_t>(b.nodes_.size(), EMPTY)};
    const Node2& r = b.nodes_[root];
    if (r.count > 0) {
        // A single leaf: wrap it in a one-child inner node so that the root is always inner.
        Node4 n4;
        std::memset(&n4, 0, sizeof(n4));
        for (int i = 0; i < 4; ++i) {
            for (int k = 0; k < 3; ++k) {
                n4.c[i].lo[k] = INFINITY;
                n4.c[i].hi[k] = -INFINITY;
            }
            n4.c[i].ref = EMPTY;
        }
        for (int k = 0; k < 3; ++k) {
            n4.c[0].lo[k] = r.lo[k];
            n4.c[0].hi[k] = r.hi[k];
        }
        out->nodes.push_back(n4);
        out->nodes[0].c[0].ref = c.emit_leaf(r);
        out->max_stack = 1;
        Child ch = out->nodes[0].c[0];
        out->wide.push_back(ch);
        out->wide_root = (wide_base << 6) | 0u;
        out->wide_depth = 1;
    } else {
        uint32_t ri;
        out->max_stack = c.emit_inner(root, &ri) + 1;
        out->wide_root = c.emit_wide(root, wide_width, wide_base, 1);
    }
}

}  // namespace rtbvh
