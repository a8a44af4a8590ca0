// bvh_build.h -- host-side builder of the 4-wide bounding volume hierarchy the traversal kernels
// walk. The reference has no tree (it scans every 8-triangle leaf box of an object in turn,
// src/main.rs:103); the tree only has to be conservative with respect to the reference's
// per-triangle arithmetic, because the nearest hit does not depend on the visiting order.
#ifndef RT_BVH_BUILD_H
#define RT_BVH_BUILD_H

#include <stdint.h>

#include <vector>

namespace rtbvh {

// One 128-byte node: four child boxes in SoA form plus four child references.
//   child[k] == EMPTY              : slot unused
//   child[k] &  LEAF_BIT           : leaf; bits 3..30 = first triangle record, bits 0..2 = count-1
//   otherwise                      : index of an inner node
struct Node4 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-byte line");

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t MAX_LEAF = 4;

struct Result {
    std::vector<Node4> nodes;     // nodes[0] is the root (always an inner node)
    std::vector<uint32_t> order;  // triangle record r holds input triangle order[r]
    uint32_t max_stack = 0;       // upper bound on pending entries during any traversal
};

// tri_bounds: n boxes of 6 floats (min xyz, max xyz), exact bounds of each triangle's three
// vertices. Boxes are widened by a small relative margin inside, so that a triangle whose
// computed hit lies a rounding error outside its exact box is still reached.
// node_index_base is added to every inner-node reference (several trees share one node array);
// record_base likewise to every leaf's first triangle record.
// balanced = false: binned surface-area-heuristic splits. balanced = true: median splits only,
// which bounds the tree depth (and with it max_stack) at the cost of looser boxes.
void build(const float* tri_bounds, uint32_t n, uint32_t node_index_base, uint32_t record_base,
           int n_threads, bool balanced, Result* out);

}  // namespace rtbvh

#endif
