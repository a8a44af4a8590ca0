//! Raw bindings of include/rays_hip.h. Layouts are `#[repr(C)]` copies of the header's structs.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_MISS: u32 = 0xFFFF_FFFF;
pub const RT_BUILD_DEVICE: u32 = 1;
pub const RT_BUILD_HOST: u32 = 2;
pub const RT_TILE: u32 = 32;
pub const RT_RENDER_EVENTS: c_int = 8;

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

/// The 128 bytes of an RCCL unique id (rank 0 makes it, every rank needs it).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_comm_id {
    pub bytes: [c_char; 128],
}

#[repr(C)]
pub struct rt_object_desc {
    pub verts: *const f32,
    pub n_verts: u32,
    pub tris: *const u32,
    pub n_tris: u32,
    pub tri_data: *const f32,
    pub albedo_over_pi: f32,
    pub reflectivity: f32,
    pub roughness: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_ray {
    pub o: [f32; 3],
    pub d: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_hit {
    pub tri: u32,
    pub t: f32,
    pub u: f32,
    pub v: f32,
}

#[repr(C)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub grid: u32,
    pub reflect_depth: u32,
    pub diffuse_depth: u32,
    pub diffuse_scatt: u32,
    pub sun: [f32; 3],
    pub seed: u32,
    pub shard_index: u32,
    pub shard_count: u32,
}

#[repr(C)]
#[derive(Default)]
pub struct rt_stats {
    pub rays: u64,
    pub box_tests: u64,
    pub tri_tests: u64,
    pub node_visits: u64,
    pub primary_rays: u64,
    pub primary_box_tests: u64,
    pub primary_tri_tests: u64,
    pub primary_node_visits: u64,
    pub fetch_bytes: u64,
    pub primary_fetch_bytes: u64,
    pub camera_by_projection: u64,
    pub bounce_rays: u64,
    pub bounce_box_tests: u64,
    pub bounce_tri_tests: u64,
    pub bounce_node_visits: u64,
    pub bounce_fetch_bytes: u64,
    pub shadow_by_cells: u64,
    pub shadow_list_entries: u64,
    pub shadow_list_longest: u64,
    pub shadow_list_cells: u64,
    pub shadow_redone: u64,
    pub tree_built_on_device: u64,
}

extern "C" {
    pub fn rt_ctx_create(devices: *const c_int, n: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_ctx_destroy(ctx: *mut rt_ctx);
    pub fn rt_ctx_reload_env(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_scene_build(ctx: *mut rt_ctx, objs: *const rt_object_desc, n_objs: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_build_ex(ctx: *mut rt_ctx, objs: *const rt_object_desc, n_objs: u32, flags: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_num_tris(scene: *const rt_scene) -> u32;
    pub fn rt_scene_sizes(scene: *const rt_scene, sizes: *mut u64) -> c_int;
    pub fn rt_scene_order(scene: *const rt_scene, obj: u32, order: *mut u32) -> c_int;
    pub fn rt_scene_packed(scene: *const rt_scene, obj: u32, tris9: *mut f32, leaf_bounds6: *mut f32) -> c_int;
    pub fn rt_trace_closest(scene: *mut rt_scene, rays: *const rt_ray, n: u64, out: *mut rt_hit) -> c_int;
    pub fn rt_trace_any(scene: *mut rt_scene, rays: *const rt_ray, n: u64, occluded: *mut u8) -> c_int;
    pub fn rt_trace_closest_device(scene: *mut rt_scene, d_rays: *const rt_ray, n: u64, d_out: *mut rt_hit, stream: *mut c_void) -> c_int;
    /// `rt_trace_closest_device` without any allocation (work counter in a per-GPU static array);
    /// at most 2^31 rays per call; overlapping calls on one GPU race.
    pub fn rt_trace_closest_device_noalloc(scene: *mut rt_scene, d_rays: *const rt_ray, n: u64, d_out: *mut rt_hit, stream: *mut c_void) -> c_int;
    pub fn rt_trace_any_device(scene: *mut rt_scene, d_rays: *const rt_ray, n: u64, d_occluded: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rt_trace_closest_linear(scene: *mut rt_scene, rays: *const rt_ray, n: u64, out: *mut rt_hit) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, p: *const rt_render_params, rgba8: *mut u8, hit_ids: *mut u32) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, p: *const rt_render_params, d_rgba8: *mut u8, d_hit_ids: *mut u32, stream: *mut c_void) -> c_int;
    pub fn rt_render_device_events(scene: *mut rt_scene, p: *const rt_render_params, d_rgba8: *mut u8, d_hit_ids: *mut u32, stream: *mut c_void, events: *const *mut c_void, n_events: c_int) -> c_int;
    pub fn rt_render_device_stats(scene: *mut rt_scene, p: *const rt_render_params, d_rgba8: *mut u8, d_hit_ids: *mut u32, stats: *mut rt_stats) -> c_int;
    pub fn rt_shard_tiles(width: u32, height: u32, shard_index: u32, shard_count: u32) -> u32;
    pub fn rt_deinterleave_tiles_device(ctx: *mut rt_ctx, d_gathered: *const u8, slab_tiles: u32, width: u32, height: u32, shard_count: u32, d_rgba8: *mut u8, stream: *mut c_void) -> c_int;
    /// The frame's one exchange step: every rank's tile slab to rank 0 over RCCL, then the scatter
    /// into the frame. `nccl_comm` is an `ncclComm_t`.
    pub fn rt_gather_tiles_device(ctx: *mut rt_ctx, nccl_comm: *mut c_void, d_slab: *const u8, width: u32, height: u32, d_rgba8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rt_comm_id_create(out: *mut rt_comm_id) -> c_int;
    pub fn rt_comm_create(ctx: *mut rt_ctx, id: *const rt_comm_id, rank: c_int, n_ranks: c_int, nccl_comm: *mut *mut c_void) -> c_int;
    pub fn rt_comm_ranks(nccl_comm: *mut c_void, rank: *mut c_int, n_ranks: *mut c_int) -> c_int;
    pub fn rt_comm_destroy(nccl_comm: *mut c_void) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_version() -> *const c_char;
}
