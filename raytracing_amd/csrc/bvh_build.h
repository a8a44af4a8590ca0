// bvh_build.h -- host-side builder of the 4-wide bounding volume hierarchy the traversal kernels
// walk. The reference has no tree (it scans every 8-triangle leaf box of an object in turn,
// src/main.rs:103); the tree only has to be conservative with respect to the reference's
// per-triangle arithmetic, because the nearest hit does not depend on the visiting order (but for the
// reference's rule about triangles within rounding of each other, which the searches settle themselves:
// closest_resolve in rt_kernels.hip).
#ifndef RT_BVH_BUILD_H
#define RT_BVH_BUILD_H

#include <stdint.h>

#include <vector>

// Absolute part of the padding of the trees' boxes, as a fraction of the object's largest coordinate (the
// builders' PAD_ABS; the nearest-hit walks' tie notes count on it).
#define RT_PAD_ABS 1e-5f

namespace rtbvh {

// One 128-byte node: four children of 32 bytes each (box, reference, padding), so that one
// child is one 8-dword load and the node is one cache line.
//   ref == EMPTY              : slot unused
//   ref &  LEAF_BIT           : leaf; bits 3..30 = first triangle record, bits 0..2 = count-1
//   otherwise                 : index of an inner node
struct Child {
    float lo[3], hi[3];
    uint32_t ref;
    uint32_t pad;
};
struct Node4 {
    Child c[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-byte line");

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t MAX_LEAF = 4;

// The same tree collapsed further, for the beam walk in which one lane looks at one child: a
// node is a run of up to `width` (<= 64) consecutive Child entries. An inner reference is
// (index of the first child << 6) | (count - 1); leaf references are those of the 4-wide tree
// and point at the same triangle records.
constexpr uint32_t MAX_WIDE = 64;
// An inner reference keeps the index of its first child in bits 6..30 (bit 31 is LEAF_BIT), so all
// wide trees of a scene together may hold at most 2^25 child records. Both builders (host and
// device) check their totals against this before they hand out indices.
constexpr uint64_t MAX_WIDE_CHILDREN = 1ull << 25;
constexpr bool wide_children_fit(uint64_t n_children_total) { return n_children_total < MAX_WIDE_CHILDREN; }
constexpr uint32_t wide_ref(uint32_t first_child, uint32_t count) { return (first_child << 6) | (count - 1u); }

struct Result {
    std::vector<Node4> nodes;     // nodes[0] is the root (always an inner node)
    std::vector<uint32_t> order;  // triangle record r holds input triangle order[r]
    uint32_t max_stack = 0;       // upper bound on pending entries during any traversal
    std::vector<Child> wide;      // children of the wide nodes
    uint32_t wide_root = 0;       // reference of the wide root node
    uint32_t wide_depth = 0;      // number of inner levels on the longest root-to-leaf path
};

// tri_bounds: n boxes of 6 floats (min xyz, max xyz), exact bounds of each triangle's three
// vertices. Boxes are widened by a small relative margin inside, so that a triangle whose
// computed hit lies a rounding error outside its exact box is still reached.
// node_index_base is added to every inner-node reference (several trees share one node array);
// record_base likewise to every leaf's first triangle record.
// balanced = false: binned surface-area-heuristic splits. balanced = true: median splits only,
// which bounds the tree depth (and with it max_stack) at the cost of looser boxes.
// wide_width (2..64) is the fan-out of the wide collapse; wide_base is added to its child indices.
void build(const float* tri_bounds, uint32_t n, uint32_t node_index_base, uint32_t record_base,
           uint32_t wide_width, uint32_t wide_base, int n_threads, bool balanced, Result* out);

}  // namespace rtbvh

#endif
