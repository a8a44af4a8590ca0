// rt_device.h -- device-side data layout shared by the kernels and the host API.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_build.h"

#define RT_BLOCK 256        // threads per workgroup: four 64-lane waves
#define RT_BLOCK0 256       // workgroup of the frame megakernel when it needs no per-lane stacks
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 16     // traversal stack entries kept in LDS per lane
#endif
#ifndef RT_STACK_SPILL
#define RT_STACK_SPILL 80   // further entries in private memory
#endif
#define RT_STACK_MAX (RT_STACK_LDS + RT_STACK_SPILL)
#define RT_MAX_REFLECT 8    // deepest mirror chain the frame kernel keeps on its stack
#define RT_BEAM_LEVELS 16   // wide-tree levels whose pending children a wave keeps in LDS

// One object of the World (src/objects.rs:18-39), as the kernels see it. 64 bytes.
struct DevObject {
    float bbmin[3];        // Object.bounding_box (src/objects.rs:48-58), exact
    float bbmax[3];
    uint32_t tri_base;     // global id of the object's first triangle
    uint32_t n_tris;       // all triangles
    uint32_t wide_root;    // reference of the object's wide root node (beam walk)
    uint32_t root;         // index of the object's 4-wide root node, EMPTY when n_tris < 8
    uint32_t rest_base;    // first entry in rest_verts of the trailing n_tris % 8 triangles
    uint32_t n_rest;
    float albedo;          // Material (src/objects.rs:176-190); albedo already divided by PI
    float reflectivity;
    float roughness;
    uint32_t leaf_base;    // first entry in leaf_bounds of the object's 8-triangle leaves
};
static_assert(sizeof(DevObject) == 64, "DevObject layout");

// Everything a search needs, passed to kernels by value.
// The persistent walks hand out the shared part of a launch's rays through RT_COUNTER_SHARDS counters, one
// per 64-byte line (a single word shared by all 8 XCDs serves about 88 requests a microsecond, which
// bounds launches of cheap rays): a launch's counter block is RT_COUNTER_BYTES, zero before the launch.
#define RT_COUNTER_SHARDS 8u
#define RT_COUNTER_BYTES (RT_COUNTER_SHARDS * 64u)

struct SceneView {
    const DevObject* objs;
    uint32_t n_objs;
    const rtbvh::Node4* nodes;  // all objects' 4-wide trees, 128 B per node
    const float4* qnodes;       // the same nodes in 64 B each: child boxes as bytes on the node's own grid (node_step_q)
    const rtbvh::Child* wide;   // all objects' wide trees, 32 B per child
    const float4* recs;         // triangle records in tree-leaf order, 3 x float4 each:
                                //   (v0.x v0.y v0.z e1.x) (e1.y e1.z e2.x e2.y) (e2.z id - -)
    const float* tri9;          // reference order: e1 xyz, e2 xyz, v0 xyz per triangle
    const float* leaf_bounds;   // reference order: min xyz, max xyz per 8-triangle leaf
    const float4* leaf_gate;    // the same boxes as two float4 a leaf: (min xyz, w) (max xyz, 0), w = 1 for a leaf in a
                                // packet of eight (PackedBoxes::intersect), 0 for one of an object's last
                                // n_leaves % 8 (box_intersection_check): what ref_leaf_gate reads
    uint32_t leaves_uniform;    // 1: no object has trailing triangles, so triangle id lies in leaf id / 8
    uint32_t n_rest_total;      // trailing triangles of all objects together
    const float* tri_data;      // reference order: normal xyz, color rgb per triangle
    const float* rest_verts;    // a xyz, b xyz, c xyz of every trailing triangle
    uint32_t walk_counter_shards;  // counters the persistent walks hand their shared rays out through: 1 or RT_COUNTER_SHARDS
    float beam_limit;           // widest beam footprint worth a beam walk (scene extent / 128)
    float scene_extent;         // largest side of the bounding box of the objects that have a tree
};

struct DevRenderParams {
    uint32_t width, height, grid, spp;
    uint32_t reflect_depth, diffuse_depth, diffuse_scatt;
    float sun[3];
    uint32_t seed;
    uint32_t tiles_x;          // tiles per row of the frame
    uint32_t n_tiles;          // tiles this launch renders
    uint32_t tile_first;       // position of the launch's first tile in the frame's (or shard's) tile order
    uint32_t compact;          // 1: write tile-compact output, 0: write the frame in place
    uint32_t beam_levels;      // deepest wide tree of the scene (levels of pending children kept in LDS)
    uint32_t shard_index, shard_count;  // tile (tx, ty) belongs to shard (tx + ty) % shard_count
    uint32_t shard_magic;               // 2^32 / shard_count rounded up: x / count = (x * magic) >> 32 for x < 2^16
};

// Projection used to order shadow rays: from the sun onto the plane across its dominant axis w.
struct LightProj {
    int w, a1, a2;                 // dominant axis of (scene centre - sun) and the two others
    float sun_w, sun_a1, sun_a2;   // sun position on those axes
    // The projected coordinates are taken relative to the line from the sun through the scene's
    // centre c:  u = ((p.a1 - c.a1) - uc (p.w - c.w)) / (p.w - sun.w),  uc = (c.a1 - sun.a1) / (c.w - sun.w),
    // which is (p.a1 - sun.a1) / (p.w - sun.w) - uc with small operands: a far sun's coordinates would
    // otherwise cost most of the float's digits.
    float c_w, c_a1, c_a2, uc, vc;
    float u0, uscale, v0, vscale;  // quantisation of the projected coordinates to 16 bits each
    float uabs, vabs;              // largest |u|, |v| over the scene's bounds
    uint32_t key_bits;             // leading bits of the 32-bit interleaved key that are kept (even, 8..32)
};

// Secondary rays of a sample in the kernel pipeline. `branches` rays leave the camera ray's hit:
// `scatt` scatter rays, then one mirror ray if `mirror`. Each branch is followed through up to
// `levels` hits, from one to the next by a mirror bounce. The record of ray j (= sample slot *
// branches + branch) at level l sits at l * level_rays + j.
struct RayTree {
    uint32_t branches, scatt, mirror, levels, level_rays;
};

struct DevStats {
    unsigned long long rays, box_tests, tri_tests, node_visits, fetch_bytes;
    unsigned long long walks[4];  // beam walks, mask walks, beam refits, mask-walk node fetches (per wave)
};

#endif
