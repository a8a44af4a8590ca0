"""ctypes binding of the C ABI in include/rays_hip.h (librays_hip.so) and of the few C entry
points of the host mirror (librays_host.so).

There is no fallback: if the HIP library is missing, importing a symbol from it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# RAYS_HIP_LIB selects another build of the same library (A/B runs of kernel variants).
HIP_LIB = os.environ.get("RAYS_HIP_LIB") or os.path.join(LIB_DIR, "librays_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "librays_host.so")

RT_MISS = 0xFFFFFFFF
RT_TILE = 32
RT_BUILD_DEVICE = 1
RT_BUILD_HOST = 2


class RaysError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rays_hip error {code}: {msg}")
        self.code = code


class rt_object_desc(C.Structure):
    _fields_ = [
        ("verts", C.POINTER(C.c_float)),
        ("n_verts", C.c_uint32),
        ("tris", C.POINTER(C.c_uint32)),
        ("n_tris", C.c_uint32),
        ("tri_data", C.POINTER(C.c_float)),
        ("albedo_over_pi", C.c_float),
        ("reflectivity", C.c_float),
        ("roughness", C.c_float),
    ]


class rt_render_params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("grid", C.c_uint32),
        ("reflect_depth", C.c_uint32),
        ("diffuse_depth", C.c_uint32),
        ("diffuse_scatt", C.c_uint32),
        ("sun", C.c_float * 3),
        ("seed", C.c_uint32),
        ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("primary_rays", C.c_uint64),
        ("primary_box_tests", C.c_uint64),
        ("primary_tri_tests", C.c_uint64),
        ("primary_node_visits", C.c_uint64),
        ("fetch_bytes", C.c_uint64),
        ("primary_fetch_bytes", C.c_uint64),
        ("camera_by_projection", C.c_uint64),
        ("bounce_rays", C.c_uint64),
        ("bounce_box_tests", C.c_uint64),
        ("bounce_tri_tests", C.c_uint64),
        ("bounce_node_visits", C.c_uint64),
        ("bounce_fetch_bytes", C.c_uint64),
        ("shadow_by_cells", C.c_uint64),
        ("shadow_list_entries", C.c_uint64),
        ("shadow_list_longest", C.c_uint64),
        ("shadow_list_cells", C.c_uint64),
        ("shadow_redone", C.c_uint64),
        ("tree_built_on_device", C.c_uint64),
    ]


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3)])
HIT_DTYPE = np.dtype([("tri", np.uint32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])

# Every symbol include/rays_hip.h declares: (restype, argtypes).
_VP = C.c_void_p
SYMBOLS = {
    "rt_ctx_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)]),
    "rt_ctx_destroy": (None, [_VP]),
    "rt_ctx_reload_env": (C.c_int, [_VP]),
    "rt_scene_build": (C.c_int, [_VP, C.POINTER(rt_object_desc), C.c_uint32, C.POINTER(_VP)]),
    "rt_scene_build_ex": (C.c_int, [_VP, C.POINTER(rt_object_desc), C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "rt_scene_destroy": (None, [_VP]),
    "rt_scene_num_tris": (C.c_uint32, [_VP]),
    "rt_scene_sizes": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "rt_scene_order": (C.c_int, [_VP, C.c_uint32, _VP]),
    "rt_scene_packed": (C.c_int, [_VP, C.c_uint32, _VP, _VP]),
    "rt_trace_closest": (C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    "rt_trace_any": (C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    "rt_trace_closest_device": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP]),
    "rt_trace_any_device": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP]),
    "rt_trace_closest_device_noalloc": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP]),
    "rt_trace_closest_linear": (C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    "rt_render": (C.c_int, [_VP, C.POINTER(rt_render_params), _VP, _VP]),
    "rt_render_device": (C.c_int, [_VP, C.POINTER(rt_render_params), _VP, _VP, _VP]),
    "rt_render_device_events": (C.c_int, [_VP, C.POINTER(rt_render_params), _VP, _VP, _VP, C.POINTER(_VP), C.c_int]),
    "rt_render_device_stats": (C.c_int, [_VP, C.POINTER(rt_render_params), _VP, _VP, C.POINTER(rt_stats)]),
    "rt_shard_tiles": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_deinterleave_tiles_device": (
        C.c_int,
        [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP, _VP],
    ),
    "rt_gather_tiles_device": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, _VP, _VP]),
    "rt_comm_id_create": (C.c_int, [_VP]),
    "rt_comm_create": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "rt_comm_ranks": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_comm_destroy": (C.c_int, [_VP]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
}

_hip = None
_host = None


def hip_lib():
    """Loads librays_hip.so and types every declared symbol. Raises if the library is absent."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RaysError(-2, f"{HIP_LIB} not found: build it with `python -m raytracing_amd.build`")
        # PyTorch wheels bundle a HIP runtime of their own. Loaded before this library it serves both;
        # loaded after it the process holds two runtimes whose streams, events and pointers do not mix
        # (the library then refuses its device entry points). So where torch is installed it goes first.
        import importlib.util
        import sys
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None and not os.environ.get("RAYS_HIP_NO_TORCH"):
            import torch  # noqa: F401
        lib = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _hip = lib
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB):
            raise RaysError(-2, f"{HOST_LIB} not found: build it with `python -m raytracing_amd.build`")
        lib = C.CDLL(HOST_LIB)
        lib.rh_read_obj.restype = C.c_int
        lib.rh_read_obj.argtypes = [
            C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
            C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
            C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32),
            C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_char_p),
        ]
        lib.rh_free.restype = None
        lib.rh_free.argtypes = [_VP]
        lib.rh_calculate_normal.restype = None
        lib.rh_calculate_normal.argtypes = [C.POINTER(C.c_float)] * 4
        lib.rh_material_albedo.restype = C.c_float
        lib.rh_material_albedo.argtypes = [C.c_float]
        _host = lib
    return _host


def check(rc):
    if rc != 0:
        raise RaysError(rc, hip_lib().rt_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(_VP) if a is not None else None


def make_descs(objects, desc_type=rt_object_desc):
    """objects: list of dicts with verts (n,3) f32, tris (m,3) u32, tri_data (m,6) f32 and
    material = (albedo_over_pi, reflectivity, roughness). Returns (ctypes array, keepalive)."""
    keep = []
    arr = (desc_type * len(objects))()
    for i, ob in enumerate(objects):
        v = np.ascontiguousarray(ob["verts"], dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(ob["tris"], dtype=np.uint32).reshape(-1, 3)
        d = np.ascontiguousarray(ob["tri_data"], dtype=np.float32).reshape(-1, 6)
        assert len(t) == len(d)
        keep += [v, t, d]
        arr[i].verts = v.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].n_verts = len(v)
        arr[i].tris = t.ctypes.data_as(C.POINTER(C.c_uint32))
        arr[i].n_tris = len(t)
        arr[i].tri_data = d.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].albedo_over_pi, arr[i].reflectivity, arr[i].roughness = [float(x) for x in ob["material"]]
    return arr, keep


def render_params(width, height, grid=2, reflect_depth=4, diffuse_depth=2, diffuse_scatt=10,
                  sun=(0.0, 25000.0, 100000.0), seed=0, shard_index=0, shard_count=1):
    p = rt_render_params()
    p.width, p.height, p.grid = width, height, grid
    p.reflect_depth, p.diffuse_depth, p.diffuse_scatt = reflect_depth, diffuse_depth, diffuse_scatt
    p.sun[0], p.sun[1], p.sun[2] = [float(x) for x in sun]
    p.seed, p.shard_index, p.shard_count = seed, shard_index, shard_count
    return p


def _switches():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("RAYS_HIP_")))


class Context:
    """rt_ctx: one GPU."""

    def __init__(self, device=0):
        self.lib = hip_lib()
        self.handle = _VP()
        dev = C.c_int(device)
        check(self.lib.rt_ctx_create(C.byref(dev), 1, C.byref(self.handle)))
        self._env = _switches()

    def sync_env(self):
        """The library reads its RAYS_HIP_* switches when the context is made; tests and benchmark legs set
        them between calls, so every call through this binding has them read again if they have changed."""
        env = _switches()
        if env != self._env and self.handle:
            check(self.lib.rt_ctx_reload_env(self.handle))
            self._env = env

    def close(self):
        if self.handle:
            self.lib.rt_ctx_destroy(self.handle)
            self.handle = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def deinterleave_device(self, gathered_ptr, slab_tiles, width, height, shard_count, rgba_ptr, stream=0):
        check(self.lib.rt_deinterleave_tiles_device(self.handle, gathered_ptr, slab_tiles, width, height,
                                                    shard_count, rgba_ptr, stream))


RT_COMM_ID_BYTES = 128


def comm_id_create():
    """rt_comm_id_create: the 128 bytes rank 0 hands to the other ranks."""
    buf = C.create_string_buffer(RT_COMM_ID_BYTES)
    check(hip_lib().rt_comm_id_create(buf))
    return buf.raw


class Comm:
    """An RCCL communicator over the ranks of one node, made by rt_comm_create (collective)."""

    def __init__(self, ctx, comm_id, rank, n_ranks):
        self.lib = ctx.lib
        self.ctx = ctx
        self.handle = _VP()
        buf = C.create_string_buffer(bytes(comm_id), RT_COMM_ID_BYTES)
        check(self.lib.rt_comm_create(ctx.handle, buf, rank, n_ranks, C.byref(self.handle)))

    def ranks(self):
        r, n = C.c_int(), C.c_int()
        check(self.lib.rt_comm_ranks(self.handle, C.byref(r), C.byref(n)))
        return r.value, n.value

    def gather_tiles_device(self, slab_ptr, width, height, rgba_ptr, stream=0):
        check(self.lib.rt_gather_tiles_device(self.ctx.handle, self.handle, slab_ptr, width, height, rgba_ptr, stream))

    def close(self):
        if self.handle:
            self.lib.rt_comm_destroy(self.handle)
            self.handle = _VP()


class Scene:
    """rt_scene: the packed World on one GPU."""

    def __init__(self, ctx, objects, device_build=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self.n_objs = len(objects)
        self.obj_tris = [len(np.asarray(o["tris"]).reshape(-1, 3)) for o in objects]
        descs, keep = make_descs(objects)
        self.handle = _VP()
        # device_build: True / False say where the trees are built, None leaves it to the library (by size).
        flags = 0 if device_build is None else (RT_BUILD_DEVICE if device_build else RT_BUILD_HOST)
        check(self.lib.rt_scene_build_ex(ctx.handle, descs, len(objects), flags, C.byref(self.handle)))
        del keep

    def close(self):
        if self.handle:
            self.lib.rt_scene_destroy(self.handle)
            self.handle = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_tris(self):
        return self.lib.rt_scene_num_tris(self.handle)

    def sizes(self):
        """Device bytes of the scene's arrays: wide tree, 4-wide tree, triangle records, tri_data."""
        out = (C.c_uint64 * 4)()
        check(self.lib.rt_scene_sizes(self.handle, out))
        return dict(zip(("wide", "nodes4", "recs", "tri_data"), [int(x) for x in out]))

    def order(self, obj):
        out = np.empty(self.obj_tris[obj], dtype=np.uint32)
        check(self.lib.rt_scene_order(self.handle, obj, _ptr(out)))
        return out

    def packed(self, obj):
        n = self.obj_tris[obj]
        tris9 = np.empty((n, 9), dtype=np.float32)
        leaves = np.empty((n // 8, 6), dtype=np.float32)
        check(self.lib.rt_scene_packed(self.handle, obj, _ptr(tris9), _ptr(leaves)))
        return tris9, leaves

    @staticmethod
    def _rays(rays):
        r = np.ascontiguousarray(rays)
        if r.dtype != RAY_DTYPE:
            r = np.ascontiguousarray(np.asarray(rays, dtype=np.float32).reshape(-1, 6)).view(RAY_DTYPE).reshape(-1)
        return r

    def trace_closest(self, rays):
        self.ctx.sync_env()
        r = self._rays(rays)
        out = np.empty(len(r), dtype=HIT_DTYPE)
        check(self.lib.rt_trace_closest(self.handle, _ptr(r), len(r), _ptr(out)))
        return out

    def trace_closest_linear(self, rays):
        self.ctx.sync_env()
        r = self._rays(rays)
        out = np.empty(len(r), dtype=HIT_DTYPE)
        check(self.lib.rt_trace_closest_linear(self.handle, _ptr(r), len(r), _ptr(out)))
        return out

    def trace_any(self, rays):
        self.ctx.sync_env()
        r = self._rays(rays)
        out = np.empty(len(r), dtype=np.uint8)
        check(self.lib.rt_trace_any(self.handle, _ptr(r), len(r), _ptr(out)))
        return out

    def trace_closest_device(self, rays_ptr, n, out_ptr, stream=0):
        self.ctx.sync_env()
        check(self.lib.rt_trace_closest_device(self.handle, rays_ptr, n, out_ptr, stream))

    def trace_closest_device_noalloc(self, rays_ptr, n, out_ptr, stream=0):
        self.ctx.sync_env()
        check(self.lib.rt_trace_closest_device_noalloc(self.handle, rays_ptr, n, out_ptr, stream))

    def trace_any_device(self, rays_ptr, n, out_ptr, stream=0):
        self.ctx.sync_env()
        check(self.lib.rt_trace_any_device(self.handle, rays_ptr, n, out_ptr, stream))

    def render(self, params, want_hit_ids=False):
        self.ctx.sync_env()
        n_pix = params.width * params.height
        rgba = np.zeros((params.height, params.width, 4), dtype=np.uint8)
        ids = np.full(n_pix * params.grid * params.grid, RT_MISS, dtype=np.uint32) if want_hit_ids else None
        check(self.lib.rt_render(self.handle, C.byref(params), _ptr(rgba), _ptr(ids)))
        return (rgba, ids) if want_hit_ids else rgba

    def render_device(self, params, rgba_ptr, hit_ids_ptr=None, stream=0):
        self.ctx.sync_env()
        check(self.lib.rt_render_device(self.handle, C.byref(params), rgba_ptr, hit_ids_ptr, stream))

    def render_device_stats(self, params, rgba_ptr, hit_ids_ptr=None):
        self.ctx.sync_env()
        st = rt_stats()
        check(self.lib.rt_render_device_stats(self.handle, C.byref(params), rgba_ptr, hit_ids_ptr, C.byref(st)))
        return {name: getattr(st, name) for name, _ in rt_stats._fields_}

    def render_device_events(self, params, rgba_ptr, events, hit_ids_ptr=None, stream=0):
        """events: sequence of RT_RENDER_EVENTS raw hipEvent_t handles (ints)."""
        self.ctx.sync_env()
        arr = (_VP * len(events))(*events)
        check(self.lib.rt_render_device_events(self.handle, C.byref(params), rgba_ptr, hit_ids_ptr, stream, arr,
                                               len(events)))


def shard_tiles(width, height, shard_index, shard_count):
    return hip_lib().rt_shard_tiles(width, height, shard_index, shard_count)


def read_obj(path, offset, color):
    """read_obj of the reference (src/main.rs:38-72) through the host mirror's C entry point.
    Returns dict(verts, tris, tri_data)."""
    lib = host_lib()
    off = (C.c_float * 3)(*[float(x) for x in offset])
    col = (C.c_float * 3)(*[float(x) for x in color])
    pv, pt, pd = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
    nv, nt = C.c_uint32(), C.c_uint32()
    err = C.c_char_p()
    rc = lib.rh_read_obj(os.fsencode(path), off, col, C.byref(pv), C.byref(nv), C.byref(pt), C.byref(nt),
                         C.byref(pd), C.byref(err))
    if rc != 0:
        raise RaysError(rc, (err.value or b"read_obj failed").decode())
    try:
        verts = np.ctypeslib.as_array(pv, shape=(max(nv.value, 1) * 3,))[: nv.value * 3].copy().reshape(-1, 3)
        tris = np.ctypeslib.as_array(pt, shape=(max(nt.value, 1) * 3,))[: nt.value * 3].copy().reshape(-1, 3)
        data = np.ctypeslib.as_array(pd, shape=(max(nt.value, 1) * 6,))[: nt.value * 6].copy().reshape(-1, 6)
    finally:
        lib.rh_free(pv)
        lib.rh_free(pt)
        lib.rh_free(pd)
    return {"verts": verts, "tris": tris, "tri_data": data}


def material(albedo, reflectivity, roughness):
    """Material::new (src/objects.rs:183-190): albedo is stored divided by PI (f32)."""
    return (np.float32(albedo) / np.float32(np.pi), np.float32(reflectivity), np.float32(roughness))
