"""raytracing_amd -- MI355X (gfx950) implementation of the reference ray tracer's trace_ray hot
path behind the C ABI of include/rays_hip.h, plus the host-side tools around it.

The HIP library is the only compute path: there is no CPU fallback in this package.
"""
from ._ffi import (Context, Scene, Comm, comm_id_create, RaysError, render_params, shard_tiles, read_obj, material, hip_lib,
                   RT_MISS, RT_TILE, RAY_DTYPE, HIT_DTYPE)

__all__ = ["Context", "Scene", "Comm", "comm_id_create", "RaysError", "render_params", "shard_tiles", "read_obj", "material", "hip_lib",
           "RT_MISS", "RT_TILE", "RAY_DTYPE", "HIT_DTYPE"]
