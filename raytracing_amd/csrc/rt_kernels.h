// rt_kernels.h -- launchers of the gfx950 kernels (rt_kernels.hip). Every launcher enqueues on
// the given stream and returns hipGetLastError(); none synchronises or allocates.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "../../include/rays_hip.h"
#include "rt_device.h"

namespace rtk {

hipError_t trace_closest(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out, hipStream_t s);
hipError_t trace_any(const SceneView& sv, const rt_ray* d_rays, uint64_t n, uint8_t* d_occ, hipStream_t s);
hipError_t trace_closest_linear(const SceneView& sv, const rt_ray* d_rays, uint64_t n, rt_hit* d_out,
                                hipStream_t s);
hipError_t pack_tris(const float* d_verts, const uint32_t* d_tris, uint32_t n, float* d_tri9, hipStream_t s);
hipError_t leaf_bounds(const float* d_verts, const uint32_t* d_tris, const uint32_t* d_leaf_first, uint32_t n_leaves,
                       float* d_leaf_bounds, hipStream_t s);
hipError_t gather_records(const float* d_tri9, const uint32_t* d_order, uint32_t n, float4* d_recs, hipStream_t s);
// mode: 0 camera + shadow rays, 1 + mirror chains, 2 + diffuse scattering (see k_render).
hipError_t render(const SceneView& sv, const DevRenderParams& p, const uint32_t* d_tile_list, int mode,
                  uint8_t* d_rgba8, uint32_t* d_hit_ids, DevStats* d_stats, hipStream_t s);
hipError_t deinterleave(const uint8_t* d_gathered, const uint32_t* d_tile_slot, uint32_t slab_tiles, uint32_t width,
                        uint32_t height, uint32_t tiles_x, uint32_t n_tiles_total, uint32_t shard_count,
                        uint8_t* d_rgba8, hipStream_t s);

}  // namespace rtk

#endif
