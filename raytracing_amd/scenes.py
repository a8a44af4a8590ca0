"""Scene descriptions for tests and benchmarks: the reference's default scene
(src/main.rs:406-448) and the synthetic triangle soups of the benchmark configurations.

Every object is a dict: verts (n,3) f32, tris (m,3) u32, tri_data (m,6) f32 (normal, colour),
material (albedo / pi, reflectivity, roughness).
"""
import os

import numpy as np

from . import _ffi

SUN_T0 = (0.0, 25000.0, 100000.0)  # src/main.rs:479-483 at t = 0


def sun_at(t):
    """The animated sun of src/main.rs:476-484: a circle of radius 100,000 about the y axis at
    height 25,000, one turn every 20 pi seconds; single precision as in the reference."""
    a = np.float32(t) / np.float32(10.0)
    return (float(np.sin(a, dtype=np.float32) * np.float32(100000.0)), 25000.0,
            float(np.cos(a, dtype=np.float32) * np.float32(100000.0)))


TEAPOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "teapot.obj")


def color_u8(r, g, b):
    """Color::from_u8, src/objects.rs:105-112."""
    return (np.array([r, g, b], dtype=np.float32) / np.float32(255.0)).astype(np.float32)


def default_scene(teapot_path=TEAPOT):
    """Two teapots and the two-triangle floor of src/main.rs:406-448."""
    objs = []
    a = _ffi.read_obj(teapot_path, (-0.9, -2.0, 9.0), color_u8(0x50, 0xC8, 0x78))
    a["material"] = _ffi.material(1.0, 0.9, 0.1)
    objs.append(a)
    b = _ffi.read_obj(teapot_path, (3.0, -2.0, 6.0), color_u8(0xFF, 0, 0))
    b["material"] = _ffi.material(1.0, 0.025, 0.975)
    objs.append(b)
    grey = color_u8(128, 128, 128)
    floor = {
        "verts": np.array([[-10000.0, -3.0, -10000.0], [-10000.0, -3.0, 10000.0], [10000.0, -3.0, 10000.0],
                           [10000.0, -3.0, -10000.0]], dtype=np.float32),
        "tris": np.array([[0, 1, 2], [0, 3, 2]], dtype=np.uint32),
        "tri_data": np.array([[0.0, 1.0, 0.0, *grey], [0.0, 1.0, 0.0, *grey]], dtype=np.float32),
        "material": _ffi.material(1.0, 0.0, 1.0),
    }
    objs.append(floor)
    return objs


# ---- PCG32 (XSH-RR 64/32), vectorised by jump-ahead -------------------------------------------
_PCG_MULT = np.uint64(6364136223846793005)
_PCG_INC = np.uint64(1442695040888963407)


def pcg32_stream(seed, n):
    """n outputs of PCG32 seeded as pcg32_srandom_r(seed, stream) with the default increment:
    state = 0; step; state += seed; step; then output i is produced from the state before its
    step. Vectorised: state_i = a^i * s0 + inc * (a^i - 1) / (a - 1) in wrapping u64 arithmetic."""
    with np.errstate(over="ignore"):
        s0 = np.uint64(0) * _PCG_MULT + _PCG_INC
        s0 = (s0 + np.uint64(seed)) * _PCG_MULT + _PCG_INC
        out = np.empty(n, dtype=np.uint32)
        chunk = 1 << 22
        state = s0
        # a^i and the geometric sums for one chunk, reused for every chunk
        m = min(chunk, n)
        ap = np.empty(m + 1, dtype=np.uint64)
        ap[0] = np.uint64(1)
        ap[1:] = _PCG_MULT
        ap = np.cumprod(ap, dtype=np.uint64)                  # a^0 .. a^m
        geo = np.empty(m + 1, dtype=np.uint64)
        geo[0] = np.uint64(0)
        geo[1:] = np.cumsum(ap[:-1], dtype=np.uint64)         # sum_{k<i} a^k
        done = 0
        while done < n:
            k = min(m, n - done)
            st = ap[:k] * state + geo[:k] * _PCG_INC
            xs = (((st >> np.uint64(18)) ^ st) >> np.uint64(27)).astype(np.uint32)
            rot = (st >> np.uint64(59)).astype(np.uint32)
            out[done:done + k] = (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))
            state = ap[k] * state + geo[k] * _PCG_INC
            done += k
    return out


def _unit_floats(u32):
    """24 random bits to [0, 1), exactly representable in f32."""
    return (u32 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _morton30(c, lo, hi):
    q = np.clip(((c - lo) / (hi - lo) * 1024.0).astype(np.int64), 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


def calculate_normals(a, b, c):
    """calculate_normal (src/objects.rs:5-7) on arrays, f32, glam's operation order:
    cross = (y*rz - ry*z, z*rx - rz*x, x*ry - rx*y); normalize = v * (1 / sqrt(dot))."""
    e1 = (b - a).astype(np.float32)
    e2 = (c - a).astype(np.float32)
    x, y, z = e1[:, 0], e1[:, 1], e1[:, 2]
    rx, ry, rz = e2[:, 0], e2[:, 1], e2[:, 2]
    n = np.stack([y * rz - ry * z, z * rx - rz * x, x * ry - rx * y], axis=1).astype(np.float32)
    d = (n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]).astype(np.float32) + n[:, 2] * n[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (np.float32(1.0) / np.sqrt(d.astype(np.float32))).astype(np.float32)
    return (n * r[:, None]).astype(np.float32)


def soup(n_tris, seed, material=(1.0, 0.0, 1.0)):
    """Random triangle soup, one object. Triangle i draws, in this order, centre x y z, first
    edge x y z, second edge x y z, colour r g b from one PCG32 stream (12 draws per triangle).
    Centre uniform in [-10,10]^2 x [5,25], edges uniform in [-0.1,0.1]^3, colour in [0,1)^3.
    Vertices: a = centre - (e1 + e2) / 3, b = a + e1, c = a + e2. Triangles are emitted in Morton
    order of the centre (30 bits) with their own three vertices, so optimize_model's sort is the
    identity and consecutive groups of eight are spatially tight."""
    u = _unit_floats(pcg32_stream(seed, 12 * n_tris)).reshape(n_tris, 12)
    lo = np.array([-10.0, -10.0, 5.0], dtype=np.float32)
    hi = np.array([10.0, 10.0, 25.0], dtype=np.float32)
    centre = (lo + u[:, 0:3] * (hi - lo)).astype(np.float32)
    e1 = (np.float32(-0.1) + u[:, 3:6] * np.float32(0.2)).astype(np.float32)
    e2 = (np.float32(-0.1) + u[:, 6:9] * np.float32(0.2)).astype(np.float32)
    colour = u[:, 9:12].astype(np.float32)
    order = np.argsort(_morton30(centre, lo, hi), kind="stable")
    centre, e1, e2, colour = centre[order], e1[order], e2[order], colour[order]
    a = (centre - (e1 + e2) / np.float32(3.0)).astype(np.float32)
    b = (a + e1).astype(np.float32)
    c = (a + e2).astype(np.float32)
    verts = np.stack([a, b, c], axis=1).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    normals = calculate_normals(a, b, c)
    tri_data = np.concatenate([normals, colour], axis=1).astype(np.float32)
    return [{"verts": verts, "tris": tris, "tri_data": tri_data, "material": _ffi.material(*material)}]


def sample_offsets(grid):
    """Sub-sample offsets in the reference's order for grid 2 (src/main.rs:322-364), row-major
    (i + 0.5) / n otherwise."""
    if grid == 2:
        return np.array([[0.25, 0.25], [0.75, 0.75], [0.25, 0.75], [0.75, 0.25]], dtype=np.float32)
    k = np.arange(grid * grid)
    ox = ((k % grid).astype(np.float32) + np.float32(0.5)) / np.float32(grid)
    oy = ((k // grid).astype(np.float32) + np.float32(0.5)) / np.float32(grid)
    return np.stack([ox, oy], axis=1).astype(np.float32)


def camera_rays(width, height, grid=2, pixels=None):
    """Primary rays of draw (src/main.rs:313-330) as an (n, 6) f32 array, ordered
    [pixel][sample]. aspect_ratio is the reference's integer division. pixels: optional array of
    pixel indices (y * width + x) to restrict to."""
    if pixels is None:
        pixels = np.arange(width * height, dtype=np.int64)
    pixels = np.asarray(pixels, dtype=np.int64)
    x = (pixels % width).astype(np.float32)
    y = (pixels // width).astype(np.float32)
    off = sample_offsets(grid)
    aspect = np.float32(width // height)
    W, H = np.float32(width), np.float32(height)
    xx = ((np.float32(2.0) * (x[:, None] + off[None, :, 0]) / W - np.float32(1.0)) * aspect * np.float32(1.0))
    yy = ((np.float32(1.0) - np.float32(2.0) * (y[:, None] + off[None, :, 1]) / H) * np.float32(1.0))
    rays = np.zeros((len(pixels), len(off), 6), dtype=np.float32)
    rays[:, :, 3] = xx
    rays[:, :, 4] = yy
    rays[:, :, 5] = 1.0
    return rays.reshape(-1, 6)
