"""CPU-side tests: packing and ordering of the oracle against independent statements of
optimize_model / pack_model, the scene generators, the golden vectors, the C ABI's symbol table,
and the tile sharding logic. No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytracing_amd import _ffi, scenes
from raytracing_amd import distributed as rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "vectors.npz"))


# ---- optimize_model / pack_model -------------------------------------------------------------
def test_default_scene_counts(default_objects, oracle):
    """teapot.obj: 3644 'v' and 6320 'f' lines; 12,642 triangles in all (src/main.rs:456-459).
    Per teapot 790 leaves = 98 packets + 6 leftover leaves + 0 trailing triangles; the floor is two
    trailing triangles (src/opt.rs:64-72)."""
    assert [len(o["verts"]) for o in default_objects] == [3644, 3644, 4]
    assert [len(o["tris"]) for o in default_objects] == [6320, 6320, 2]
    sc = oracle.OracleScene(default_objects)
    assert sc.num_tris == 12642
    for i in (0, 1):
        assert len(sc.packed_bounds(i)) == 98 and len(sc.rest_bounds(i)) == 6 and sc.num_rest_tri(i) == 0
        assert len(sc.packed_tris(i)) == 790
    assert len(sc.packed_bounds(2)) == 0 and sc.num_rest_tri(2) == 2


def test_optimize_model_order_is_the_stable_sort(default_objects, oracle):
    """src/opt.rs:14-24: ascending smallest vertex index, then DESCENDING largest, stable."""
    sc = oracle.OracleScene(default_objects)
    t = default_objects[0]["tris"].astype(np.int64)
    key = list(zip(t.min(axis=1), -t.max(axis=1)))
    expect = sorted(range(len(t)), key=lambda i: key[i])  # Python's sort is stable
    assert np.array_equal(sc.order(0), np.array(expect, dtype=np.uint32))
    assert not np.array_equal(sc.order(0), np.arange(len(t)))  # the teapot does get reordered


def test_pack_model_matches_an_independent_statement(default_objects, oracle):
    sc = oracle.OracleScene(default_objects)
    ob = default_objects[1]
    order = sc.order(1)
    tris = ob["tris"][order]
    v = ob["verts"]
    a, b, c = v[tris[:, 0]], v[tris[:, 1]], v[tris[:, 2]]
    n8 = len(tris) // 8
    pt = sc.packed_tris(1)
    assert pt.shape == (n8, 9, 8)
    expect = np.concatenate([(b - a), (c - a), a], axis=1).astype(np.float32)[: n8 * 8].reshape(n8, 8, 9)
    assert np.array_equal(pt.transpose(0, 2, 1).view(np.uint32), expect.view(np.uint32))
    # leaf boxes: min / max over the 24 vertices of 8 consecutive triangles (src/opt.rs:46-60)
    allv = np.stack([a, b, c], axis=1)[: n8 * 8].reshape(n8, 24, 3)
    lo, hi = allv.min(axis=1), allv.max(axis=1)
    pb = sc.packed_bounds(1)
    got = np.concatenate([pb.transpose(0, 2, 1).reshape(-1, 6), sc.rest_bounds(1)])
    assert np.array_equal(got.view(np.uint32), np.concatenate([lo, hi], axis=1).astype(np.float32).view(np.uint32))
    # object box over ALL vertices (src/objects.rs:48-53)
    assert np.array_equal(sc.object_bounds(1), np.concatenate([v.min(axis=0), v.max(axis=0)]))


def test_read_obj_and_normals(default_objects, oracle):
    """read_obj adds the offset in f32 (src/main.rs:53); normals are calculate_normal of the three
    vertices (src/objects.rs:206-219). A degenerate triangle would store a NaN normal on both sides."""
    raw = [l.split() for l in open(scenes.TEAPOT) if l.startswith("v ")]
    v0 = np.array([[float(x) for x in l[1:4]] for l in raw], dtype=np.float32)
    off = np.array([-0.9, -2.0, 9.0], dtype=np.float32)
    assert np.array_equal(default_objects[0]["verts"], (v0 + off).astype(np.float32))
    ob = default_objects[0]
    t = ob["tris"]
    n = scenes.calculate_normals(ob["verts"][t[:, 0]], ob["verts"][t[:, 1]], ob["verts"][t[:, 2]])
    got = ob["tri_data"][:, :3]
    both_nan = np.isnan(n).any(axis=1) & np.isnan(got).any(axis=1)
    assert np.array_equal(n[~both_nan].view(np.uint32), got[~both_nan].view(np.uint32))
    one = oracle.calculate_normal(ob["verts"][t[5, 0]], ob["verts"][t[5, 1]], ob["verts"][t[5, 2]])
    assert np.array_equal(one.view(np.uint32), got[5].view(np.uint32))
    assert np.allclose(ob["tri_data"][0, 3:], np.array([0x50, 0xC8, 0x78], dtype=np.float32) / 255.0)
    assert abs(float(ob["material"][0]) - 1.0 / np.pi) < 1e-7


def test_read_obj_rejects_what_the_reference_panics_on(tmp_path):
    """`collect_tuple().unwrap()` panics unless a line yields exactly three parsed tokens
    (src/main.rs:48-59); 'f' entries like 1/2/3 do not parse as usize."""
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2/2 3/3\n")
    with pytest.raises(_ffi.RaysError):
        _ffi.read_obj(str(p), (0, 0, 0), (1, 1, 1))
    p.write_text("v 0 0\n")
    with pytest.raises(_ffi.RaysError):
        _ffi.read_obj(str(p), (0, 0, 0), (1, 1, 1))
    p.write_text("# comment\nv 0 0 0\nv 1 0  0\nv 0 1 0\nvn 0 0 1\nf 1 2 3\n")  # double space: empty token dropped
    ob = _ffi.read_obj(str(p), (1, 2, 3), (1, 1, 1))
    assert ob["verts"].shape == (3, 3) and ob["tris"].tolist() == [[0, 1, 2]]
    with pytest.raises(_ffi.RaysError):
        _ffi.read_obj(str(tmp_path / "missing.obj"), (0, 0, 0), (1, 1, 1))


# ---- scene generators ------------------------------------------------------------------------
def _pcg32_reference(seed, n):
    mask = (1 << 64) - 1
    mult, inc = 6364136223846793005, 1442695040888963407
    state = (0 * mult + inc) & mask
    state = ((state + seed) * mult + inc) & mask
    out = []
    for _ in range(n):
        old = state
        state = (old * mult + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        out.append(((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF)
    return np.array(out, dtype=np.uint32)


def test_pcg32_vectorised_matches_scalar():
    assert np.array_equal(scenes.pcg32_stream(0x5EED0001, 5000), _pcg32_reference(0x5EED0001, 5000))
    # chunk boundary
    # across the generator's internal chunk boundary (2^22 outputs): compare with a single-chunk run
    a = scenes.pcg32_stream(42, (1 << 22) + 17)
    assert np.array_equal(a[:100], _pcg32_reference(42, 100))
    mask, mult, inc = (1 << 64) - 1, 6364136223846793005, 1442695040888963407
    state = ((((0 * mult + inc) & mask) + 42) * mult + inc) & mask
    # jump the scalar generator ahead by 2^22 steps with the closed form of the LCG
    n, acc_m, acc_p, cur_m, cur_p = 1 << 22, 1, 0, mult, inc
    while n:
        if n & 1:
            acc_m, acc_p = (acc_m * cur_m) & mask, (acc_p * cur_m + cur_p) & mask
        cur_p, cur_m = ((cur_m + 1) * cur_p) & mask, (cur_m * cur_m) & mask
        n >>= 1
    state = (acc_m * state + acc_p) & mask
    tail = []
    for _ in range(17):
        old = state
        state = (old * mult + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        tail.append(((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF)
    assert a[-17:].tolist() == tail


def test_soup_recipe(oracle):
    s = scenes.soup(4096, 0x5EED0001)[0]
    assert len(s["tris"]) == 4096 and len(s["tris"]) % 64 == 0
    v = s["verts"].reshape(-1, 3, 3)
    centre = v.mean(axis=1)
    assert centre[:, :2].min() > -10.2 and centre[:, :2].max() < 10.2
    assert centre[:, 2].min() > 4.8 and centre[:, 2].max() < 25.2
    assert np.abs(v[:, 1] - v[:, 0]).max() <= 0.1001 and np.abs(v[:, 2] - v[:, 0]).max() <= 0.1001
    # own vertices in triangle order: optimize_model's sort is the identity
    sc = oracle.OracleScene([s])
    assert np.array_equal(sc.order(0), np.arange(4096))
    assert sc.num_rest_tri(0) == 0 and len(sc.rest_bounds(0)) == 0
    # Morton order keeps groups of eight tight: leaf boxes are far smaller than the scene
    pb = sc.packed_bounds(0)
    ext = (pb[:, 3:6, :] - pb[:, 0:3, :])
    assert np.median(ext) < 3.0
    assert scenes.soup(64, 1)[0]["verts"].tobytes() == scenes.soup(64, 1)[0]["verts"].tobytes()
    assert scenes.soup(64, 1)[0]["verts"].tobytes() != scenes.soup(64, 2)[0]["verts"].tobytes()


def test_camera_rays_follow_the_reference_formula():
    """src/main.rs:314,322-330: aspect = WIDTH / HEIGHT in usize, dir not normalised, four samples in
    the order (.25,.25) (.75,.75) (.25,.75) (.75,.25)."""
    r = scenes.camera_rays(1920, 1080, 2)
    assert r.shape == (1920 * 1080 * 4, 6)
    assert np.all(r[:, :3] == 0) and np.all(r[:, 5] == 1.0)
    # 1920 // 1080 == 1: the horizontal extent is [-1, 1], not [-16/9, 16/9]
    assert -1.0 < r[:, 3].min() < -0.999 and 0.999 < r[:, 3].max() < 1.0
    x, y = 100, 200
    base = (y * 1920 + x) * 4
    f = np.float32
    for s, (ox, oy) in enumerate([(0.25, 0.25), (0.75, 0.75), (0.25, 0.75), (0.75, 0.25)]):
        xx = (f(2.0) * (f(x) + f(ox)) / f(1920) - f(1.0)) * f(1.0) * f(1.0)
        yy = (f(1.0) - f(2.0) * (f(y) + f(oy)) / f(1080)) * f(1.0)
        assert r[base + s, 3] == xx and r[base + s, 4] == yy
    # 800x400: aspect 2 stretches x
    assert abs(scenes.camera_rays(800, 400, 2)[:, 3].max() - 2.0) < 0.01


# ---- golden vectors ----------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["teapot", "soup4k"])
def test_oracle_reproduces_golden_vectors(name, oracle, default_objects):
    objs = default_objects if name == "teapot" else scenes.soup(4096, 0x5EED0001)
    sc = oracle.OracleScene(objs)
    rays = GOLD[f"{name}_rays"]
    h = sc.trace_closest(rays)
    assert np.array_equal(h["tri"], GOLD[f"{name}_tri"])
    tuv = np.stack([h["t"], h["u"], h["v"]], axis=1).view(np.uint32)
    hit = h["tri"] != oracle.MISS
    assert np.array_equal(tuv[hit], GOLD[f"{name}_tuv"][hit])
    assert np.array_equal(sc.trace_any(rays), GOLD[f"{name}_occluded"])
    rgba, ids, n = sc.render(64, 36, 2, reflect_depth=1, diffuse_depth=1, want_hit_ids=True)
    assert np.array_equal(rgba, GOLD[f"{name}_frame_direct"]) and np.array_equal(ids, GOLD[f"{name}_frame_direct_ids"])
    assert n == int(GOLD[f"{name}_frame_direct_rays"][0])
    rgba, _, n = sc.render(64, 36, 2, seed=99)
    # powf / sinf / cosf come from the host libm: allow 1 LSB between hosts
    assert np.abs(rgba.astype(np.int16) - GOLD[f"{name}_frame_full"].astype(np.int16)).max() <= 1
    assert abs(n - int(GOLD[f"{name}_frame_full_rays"][0])) <= 0.001 * n


def test_oracle_frame_resolve_rules(oracle):
    """Sky is not gamma-corrected, hits are (src/main.rs:302-310); the four samples are averaged in
    order and converted with a truncating, saturating cast (src/objects.rs:90-97)."""
    tri = {"verts": np.array([[-100, -100, 5], [100, -100, 5], [0, 100, 5]], dtype=np.float32),
           "tris": np.array([[0, 1, 2]], dtype=np.uint32),
           "tri_data": np.array([[0, 0, -1, 0.5, 0.5, 0.5]], dtype=np.float32), "material": (1.0, 0.0, 1.0)}
    sc = oracle.OracleScene([tri])
    # sun behind the camera: normal (0,0,-1) faces it, nothing occludes
    rgba, ids, n = sc.render(4, 4, 2, reflect_depth=1, diffuse_depth=1, sun=(0, 0, -1000), want_hit_ids=True)
    assert (ids == 0).all() and n == 4 * 4 * 4 * 2
    c = np.float32(0.5) * np.float32(1.0)  # colour * (light_power ~ 1 * albedo 1)
    expect = int(min(float(np.float32(c) ** np.float32(1 / 2.2)) * 255.0, 255.0))
    assert abs(int(rgba[2, 2, 0]) - expect) <= 1 and rgba[2, 2, 3] == 255
    empty = {"verts": np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], dtype=np.float32), "tris": tri["tris"],
             "tri_data": tri["tri_data"], "material": tri["material"]}
    rgba, ids, n = oracle.OracleScene([empty]).render(4, 4, 2, want_hit_ids=True)
    assert (ids == oracle.MISS).all() and n == 64
    assert rgba[0, 0].tolist() == [0x87, 0xCE, 0xEB, 0xFF]


# ---- C ABI -------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rays_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 19
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    lib = _ffi.hip_lib()   # (not ctypes.CDLL directly: the binding loads torch's HIP runtime first, see hip_lib)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared but not exported"
    assert b"gfx950" in _ffi.hip_lib().rt_version()


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: without a device the context cannot be created."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.RaysError) as e:
        _ffi.Context(0)
    assert e.value.code in (-2, -1)


def test_two_hip_runtimes_are_refused_not_aborted_on():
    """A PyTorch wheel brings a HIP runtime of its own; loaded after this library the process holds two, and
    handles passed between them abort inside the runtime. The library's device entry points check for that
    first and return RT_ERR_UNSUPPORTED naming both copies. Runs in a child process, which loads the library
    before torch (the binding itself imports torch first) and needs no device: the check comes before any
    HIP call."""
    import subprocess
    import sys

    import importlib.util
    if importlib.util.find_spec("torch") is None:
        pytest.skip("no torch, no second runtime")
    code = f"""
import ctypes
lib = ctypes.CDLL({_ffi.HIP_LIB!r}, mode=ctypes.RTLD_GLOBAL)
lib.rt_last_error.restype = ctypes.c_char_p
import torch
paths = sorted({{l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}})
h, dev = ctypes.c_void_p(), ctypes.c_int(0)
rc = lib.rt_ctx_create(ctypes.byref(dev), 1, ctypes.byref(h))
print(len(paths), rc, lib.rt_last_error().decode())
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    n_paths, rc, msg = out.stdout.strip().splitlines()[-1].split(" ", 2)
    if int(n_paths) < 2:
        pytest.skip("torch uses the system's HIP runtime here")
    assert int(rc) == -4 and "two HIP runtimes" in msg and "torch" in msg and msg.count("libamdhip64") == 2


def test_native_libraries_do_not_link_the_oracle():
    import subprocess

    for lib in (_ffi.HIP_LIB, _ffi.HOST_LIB):
        out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "oracle" not in out


# ---- tile sharding -----------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,n", [(3840, 2160, 1), (3840, 2160, 8), (200, 120, 3), (33, 31, 2), (7680, 4320, 4)])
def test_tile_layout_matches_the_library(w, h, n):
    tx, ty, shard, slot, counts = rd.tile_layout(w, h, n)
    assert tx == -(-w // 32) and ty == -(-h // 32)
    for s in range(n):
        assert _ffi.shard_tiles(w, h, s, n) == counts[s]
    assert counts.sum() == tx * ty and counts.max() - counts.min() <= max(tx, ty)
    # every tile has a unique (shard, slot)
    assert len({(a, b) for a, b in zip(shard.tolist(), slot.tolist())}) == tx * ty
    assert _ffi.shard_tiles(w, h, n, n) == 0  # index out of range


def test_deinterleave_numpy_round_trip():
    rng = np.random.default_rng(3)
    w, h, n = 200, 120, 3
    frame = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    tx, ty, shard, slot, counts = rd.tile_layout(w, h, n)
    slab = int(counts.max())
    g = np.zeros((n, slab, 32, 32, 4), dtype=np.uint8)
    padded = np.zeros((ty * 32, tx * 32, 4), dtype=np.uint8)
    padded[:h, :w] = frame
    for t in range(tx * ty):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        g[shard[t], slot[t]] = padded[y0:y0 + 32, x0:x0 + 32]
    assert np.array_equal(rd.deinterleave_numpy(g, w, h, n), frame)


@pytest.fixture(scope="module")
def bvh_shim(tmp_path_factory):
    """The product's host tree builder behind a small checking shim (tests/native/bvh_check.cpp)."""
    import subprocess

    out = tmp_path_factory.mktemp("bvh") / "libbvh_check.so"
    csrc = os.path.join(ROOT, "raytracing_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", csrc, "-o", str(out),
                    os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(csrc, "bvh_build.cpp")],
                   check=True)
    lib = ctypes.CDLL(str(out))
    lib.bvh_check.restype = ctypes.c_int
    lib.bvh_check.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p]
    return lib


def _tri_boxes(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "soup":
        c = rng.uniform(-10, 10, (n, 3))
        e = rng.uniform(0, 0.2, (n, 3))
    elif kind == "line":        # centroids on one line: long thin splits
        c = np.stack([np.linspace(0, 100, n), np.zeros(n), np.zeros(n)], 1)
        e = np.full((n, 3), 0.01)
    else:                       # "same": every box identical, no split plane exists
        c = np.zeros((n, 3))
        e = np.ones((n, 3))
    return np.concatenate([c - e, c + e], 1).astype(np.float32)


@pytest.mark.parametrize("kind,n", [("soup", 8), ("soup", 9), ("soup", 4096), ("soup", 60000), ("line", 5000),
                                    ("same", 700)])
@pytest.mark.parametrize("width", [4, 32, 64])
def test_host_tree_builder_invariants(bvh_shim, kind, n, width):
    """Both trees hold every triangle exactly once, every box contains what is below it, nodes
    have at most `width` children, and the reported depth and stack bound are the real ones. The
    least-area collapse must also leave fewer wide nodes than triangles / 8 on a soup."""
    boxes = _tri_boxes(n, 1234 + n, kind)
    out = (ctypes.c_uint32 * 8)()
    for balanced in (0, 1):
        rc = bvh_shim.bvh_check(boxes.ctypes.data, n, width, 4, balanced, out)
        assert rc == 0, f"invariant {rc} violated (kind {kind}, n {n}, width {width}, balanced {balanced})"
        nodes4, wide_children, depth = out[0], out[1], out[2]
        assert nodes4 >= 1 and wide_children >= n / rtb_max_leaf() and depth >= 1
    if kind == "soup" and n >= 4096 and width == 64:
        inner = out[1] - n   # with one triangle per leaf
        assert inner <= n / 8


def rtb_max_leaf():
    return 4


def test_wide_tree_size_limit_keeps_inner_references_clear_of_the_leaf_bit(bvh_shim):
    """Both builders refuse scenes whose wide trees would need 2^25 child records or more: an inner
    reference is (first child << 6) | (count - 1) and bit 31 marks a leaf."""
    bvh_shim.bvh_wide_limit_check.restype = ctypes.c_int
    assert bvh_shim.bvh_wide_limit_check() == 0


def test_sun_at_follows_the_reference_circle():
    """src/main.rs:476-484: (sin(t / 10) * 100000, 25000, cos(t / 10) * 100000) in single precision."""
    from raytracing_amd import scenes
    assert scenes.sun_at(0.0) == scenes.SUN_T0
    for t in (1.0, 15.70796, 40.0):
        x, y, z = scenes.sun_at(t)
        a = np.float32(t) / np.float32(10.0)
        assert y == 25000.0
        assert abs(x - float(np.sin(a)) * 1e5) <= 0.02 and abs(z - float(np.cos(a)) * 1e5) <= 0.02
        assert abs(np.hypot(x, z) - 1e5) <= 0.05
