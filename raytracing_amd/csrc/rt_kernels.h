// rt_kernels.h -- launchers of the gfx950 kernels (rt_kernels.hip). Every launcher enqueues on
// the given stream and returns hipGetLastError(); none synchronises or allocates.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "../../include/rays_hip.h"
#include "rt_device.h"

namespace rtk {

// d_counter: a zeroed device word for the persistent kernel's ray hand-out. how: 'l' one ray per lane to
// the end, 'w' wave-coherent walks (nearest hit only), anything else the persistent per-lane walk.
hipError_t trace_closest(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out, uint32_t* d_counter,
                         char how, hipStream_t s);
hipError_t trace_any(const SceneView& sv, const rt_ray* d_rays, uint64_t n, uint8_t* d_occ, uint32_t* d_counter,
                     char how, hipStream_t s);
hipError_t trace_closest_linear(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                                hipStream_t s);
hipError_t pack_tris(const float* d_verts, const uint32_t* d_tris, uint32_t n, float* d_tri9, hipStream_t s);
// d_leaf_first[l]: first triangle of leaf l; bit 31 set for an object's last n_leaves % 8 leaves.
// d_leaf_bounds: 6 floats a leaf; d_leaf_gate: 2 float4 a leaf (SceneView::leaf_gate).
hipError_t leaf_bounds(const float* d_verts, const uint32_t* d_tris, const uint32_t* d_leaf_first, uint32_t n_leaves,
                       float* d_leaf_bounds, float4* d_leaf_gate, hipStream_t s);
// The 64-byte form of the 4-wide nodes that the persistent per-lane walk reads (node i -> 4 float4 at 4 i).
hipError_t quantize_nodes(const rtbvh::Node4* d_nodes, uint32_t n, float4* d_qnodes, hipStream_t s);
hipError_t gather_records(const float* d_tri9, const uint32_t* d_order, uint32_t n, float4* d_recs, hipStream_t s);
// mode: 0 camera + shadow rays, 1 + mirror chains, 2 + diffuse scattering (see k_render).
hipError_t render(const SceneView& sv, const DevRenderParams& p, const uint32_t* d_tile_list, int mode,
                  uint8_t* d_rgba8, uint32_t* d_hit_ids, DevStats* d_stats, hipStream_t s);
// Camera + shadow frame as separate kernels around a sort (see rt_kernels.hip).
hipError_t wf_primary(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const uint32_t* d_tile_list,
                      float4* d_samp, uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats,
                      hipStream_t s);
// Camera rays by projection of the triangles onto the sample grid (scenes without trailing
// triangles, all of them in front of the camera). raster_depth fills d_zbuf (width * height * spp
// words of 8 bytes: nearest (t, triangle id) per sample, all ones for none) for the whole frame, or
// for the tiles of the shard p names (clear: set every word to "none" first); raster_records turns the
// words of the launch's tiles into the records wf_primary stores and sets them back to "none", so
// that a buffer every tile of which has gone through raster_records needs no clearing. d_table: (1 + 2 * n_objs) * (width + height) * grid floats.
hipError_t raster_depth(const SceneView& sv, const DevRenderParams& p, uint32_t n_tris_total, float* d_table,
                        unsigned long long* d_zbuf, bool clear, DevStats* d_stats, hipStream_t s);
hipError_t raster_records(const SceneView& sv, const DevRenderParams& p, const LightProj& lp,
                          const uint32_t* d_tile_list, unsigned long long* d_zbuf, float4* d_samp,
                          uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats, hipStream_t s);
// Camera rays by projection, tile by tile (see rt_kernels.hip). raster_bin with fill == false writes the
// camera table, counts the triangles of every raster tile (32 x 32 samples; n_raster_tiles =
// ceil(width * grid / 32) * ceil(height * grid / 32)) and turns the counts into offsets
// (d_offsets[n_raster_tiles] = the total); with fill == true it writes the lists (at most `capacity`
// entries). raster_tiles then finds the nearest hits of the launch's tiles and writes their records.
hipError_t raster_bin(const SceneView& sv, const DevRenderParams& p, uint32_t n_tris_total, float* d_table,
                      uint32_t n_raster_tiles, uint32_t* d_counts, uint32_t* d_offsets, uint32_t* d_list, uint32_t capacity,
                      float4* d_recs, bool fill, hipStream_t s);
hipError_t raster_tiles(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const uint32_t* d_tile_list,
                        const float* d_table, const uint32_t* d_offsets, const uint32_t* d_list, const float4* d_recs,
                        float4* d_samp, uint32_t* d_state, uint32_t* d_keys, DevStats* d_stats, hipStream_t s);
hipError_t wf_shadow(const SceneView& sv, const DevRenderParams& p, uint32_t n, uint32_t cell_shift,
                     const uint32_t* d_keys, const uint32_t* d_vals, const float4* d_samp, uint32_t* d_state,
                     DevStats* d_stats, hipStream_t s, const uint32_t* d_n_valid = nullptr,
                     const uint32_t* d_done_flag = nullptr);
// Shadow rays by light-space cells (k_lcell_bin, k_lcell_check, k_shadow_cells). lcell_lists builds
// the per-cell triangle lists for the grid of lp's key: 2^key_bits cells with room for 2^cap_shift entries
// each (d_counts: one word per cell; d_lists: cells << cap_shift entries of six words and four more), and leaves
// d_flags[0] = 1 when every list fits its room and the lists of the cells that have one average at most
// max_mean entries; d_flags[1..3] = the lists' total, the longest, the cells with a list (d_flags: 64
// words; from word 16 on what shadow_cells needs at its end: the leaf boxes and d_redo). redo_all (tests): every ray the lists find
// occluded is decided over again by the reference's own search. shadow_cells then decides the first *d_n_valid sorted rays when d_flags[0] is set and does
// nothing otherwise; wf_shadow with d_done_flag = d_flags does the reverse. d_redo: room for n words
// (the slots of the rays whose occluder fails the reference's leaf test, counted in d_flags[5]; a second
// kernel of the same call runs the reference's own search for them).
hipError_t lcell_lists(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, uint32_t n_tris_total,
                       uint32_t* d_counts, float* d_lists, uint32_t cap_shift, uint32_t max_mean,
                       uint32_t* d_flags, uint32_t* d_redo, bool redo_all, hipStream_t s);
hipError_t shadow_cells(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, uint32_t n,
                        const uint32_t* d_keys, const uint32_t* d_vals, const float4* d_samp, uint32_t* d_state,
                        DevStats* d_stats, const uint32_t* d_n_valid, const uint32_t* d_counts, const float* d_lists,
                        uint32_t cap_shift, uint32_t* d_flags, uint32_t* d_redo, hipStream_t s);
hipError_t wf_resolve(const SceneView& sv, const DevRenderParams& p, const uint32_t* d_tile_list, const float4* d_samp,
                      const uint32_t* d_state, uint8_t* d_rgba8, uint32_t* d_hit_ids, hipStream_t s);
// Diffuse and mirror bounces as a pipeline: one launch of wf_bounce per level of the tree of
// secondary rays (level 0 reads the camera rays' records, level l the records of level l - 1), then
// the shading of all levels. The record arrays and d_queues hold tr.levels * tr.level_rays entries,
// d_counts one word per level. Before level 0 the caller zeroes d_counts and presets the records of
// the deeper levels to "no ray": d_dstate 0, d_dkeys all ones.
hipError_t wf_bounce(const SceneView& sv, const DevRenderParams& p, const LightProj& lp, const RayTree& tr,
                     uint32_t level, const uint32_t* d_tile_list, uint32_t* d_counter, const float4* d_samp,
                     const uint32_t* d_state, float4* d_dsamp, uint32_t* d_dstate, uint32_t* d_dkeys,
                     uint32_t* d_queues, uint32_t* d_counts, DevStats* d_stats, hipStream_t s);
hipError_t wf_resolve_tree(const SceneView& sv, const DevRenderParams& p, const RayTree& tr,
                           const uint32_t* d_tile_list, const float4* d_samp, const uint32_t* d_state,
                           const float4* d_dsamp, const uint32_t* d_dstate, uint8_t* d_rgba8, uint32_t* d_hit_ids,
                           hipStream_t s);
// Stable radix sort of (u32 key, u32 value) pairs on the low key_bits bits of the key. With
// temp == nullptr only *temp_bytes is set.
// The library's own sort of the shadow rays' keys (rt_radix.hip): stable, on the low key_bits bits, of the
// keys that are not all ones; values = the keys' positions in keys_in. keys_out / vals_out receive the
// *d_n_valid sorted pairs; the count stays on the device. `work` = radix_workspace_bytes(n) bytes.
size_t radix_workspace_bytes(size_t n);
hipError_t radix_sort_valid(void* work, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, size_t n,
                            unsigned key_bits, uint32_t* d_n_valid, hipStream_t s);
hipError_t sort_pairs_u32(void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned key_bits, hipStream_t s);
// Tree build on the device (rt_lbvh.hip). Sizes of the two trees over n single-triangle leaves:
struct LbvhLayout {
    uint32_t n_nodes4, depth4;     // 4-wide tree: nodes, levels
    uint32_t n_wide, depth_wide;   // wide tree: child records (leaf level included), node levels
    uint32_t top_count;            // children of the wide root
};
LbvhLayout lbvh_layout(uint32_t n, uint32_t fan);
// Builds both trees of one object over triangles d_tris[0 .. n) (three global vertex indices each)
// into d_nodes_all + node_base, d_wide_all + wide_base and the record order into
// d_order_all + rec_base (global triangle ids). d_keys / d_vals: two scratch arrays of n words each;
// d_temp: sort scratch for n pairs. Returns the roots' references.
hipError_t lbvh_build(const float* d_verts, const uint32_t* d_tris, uint32_t n, const float bbmin[3], const float bbmax[3],
                      uint32_t tri_base, uint32_t rec_base, uint32_t node_base, uint32_t wide_base, uint32_t fan,
                      rtbvh::Node4* d_nodes_all, rtbvh::Child* d_wide_all, uint32_t* d_order_all, uint32_t* d_keys[2],
                      uint32_t* d_vals[2], void* d_temp, size_t temp_bytes, uint32_t* root4, uint32_t* wide_root,
                      hipStream_t s);
// The same with cuts where the sorted Morton codes first differ (a binary radix tree), boxes fitted
// bottom-up and both trees collapsed from it top-down. Better trees than lbvh_build; a few host
// synchronisations per level. Needs room for n 4-wide nodes and 2 n wide child records, and
// lbvh_radix_work_bytes(n) bytes of scratch.
struct LbvhResult {
    uint32_t root4, wide_root;    // references of the roots
    uint32_t n_nodes4, n_wide;    // entries used
    uint32_t depth4, depth_wide;  // levels
};
size_t lbvh_radix_work_bytes(uint32_t n);
hipError_t lbvh_build_radix(const float* d_verts, const uint32_t* d_tris, uint32_t n, const float bbmin[3],
                            const float bbmax[3], uint32_t tri_base, uint32_t rec_base, uint32_t node_base,
                            uint32_t wide_base, uint32_t fan, rtbvh::Node4* d_nodes_all, rtbvh::Child* d_wide_all,
                            uint32_t* d_order_all, uint32_t* d_keys[2], uint32_t* d_vals[2], void* d_temp,
                            size_t temp_bytes, void* d_work, LbvhResult* out, hipStream_t s);
hipError_t deinterleave(const uint8_t* d_gathered, const uint32_t* d_tile_slot, uint32_t slab_tiles, uint32_t width,
                        uint32_t height, uint32_t tiles_x, uint32_t n_tiles_total, uint32_t shard_count,
                        uint8_t* d_rgba8, hipStream_t s);

}  // namespace rtk

#endif
