"""The CPU oracle against the reference's own known answer, hand-derived exact vectors, the
quirk list (tie order, t-min rules, NaN / inf handling), and an exact-rational model of the 8-wide
triangle test's rounding sequence."""
from fractions import Fraction

import numpy as np
import pytest

INF = float("inf")


def _tris72(tris):
    """tris: list of 8 (a, b, c) vertex triples -> e1.x[8] e1.y e1.z e2.x.. v0.z (src/simd_accel.rs:61-64)."""
    t = np.asarray(tris, dtype=np.float32).reshape(8, 3, 3)
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    e1, e2 = (b - a).astype(np.float32), (c - a).astype(np.float32)
    return np.concatenate([e1.T, e2.T, a.T], axis=0).astype(np.float32).reshape(72)


def test_reference_known_answer_box(oracle):
    """objects::tests::it_works (src/objects.rs:339-353): the only usable vector the reference holds."""
    inv = np.float32(1.0) / np.array([0.0, 0.0, 1.0], dtype=np.float32)
    assert np.isinf(inv[0]) and np.isinf(inv[1])
    assert oracle.box_intersection_check([0, 0, -2], inv, [-1, -1, -1], [1, 1, 1]) is True


def test_scalar_box_is_a_loose_test(oracle):
    """t_near.min(0) <= t_far (src/objects.rs:332) is true whenever the LINE meets the box (also
    behind the origin) and whenever t_far >= 0 with t_near > 0 (a box beside the ray)."""
    inv = np.array([1.0, 1.0, 1.0], dtype=np.float32)
    assert oracle.box_intersection_check([0, 0, 0], inv, [5, -10, -10], [6, -9, -9]) is False   # t_near > 0 > t_far
    assert oracle.box_intersection_check([0, 0, 0], inv, [5, 1, 1], [6, 2, 2]) is True          # t_near > t_far >= 0
    assert oracle.box_intersection_check([0, 0, 0], inv, [-3, -3, -3], [-1, -1, -1]) is True    # wholly behind
    assert oracle.box_intersection_check([0, 0, 0], inv, [-3, 1, 1], [-1, 2, 2]) is False       # line misses, behind


def test_exact_triangle_vector(oracle):
    """origin 0, dir +z, triangle (-1,-1,5) (1,-1,5) (0,1,5): t = 5, u = 0.25, v = 0.5, all exact."""
    tri = [[-1, -1, 5], [1, -1, 5], [0, 1, 5]]
    far = [[100, 100, 100], [101, 100, 100], [100, 101, 100]]
    t, u, v, mask = oracle.packed_tris_intersect(_tris72([tri] + [far] * 7), [0, 0, 0, 0, 0, 1], INF)
    assert (t[0], u[0], v[0]) == (5.0, 0.25, 0.5)
    assert mask == 0xFE and np.all(t[1:] == -1.0)
    assert oracle.triangle_ray_hits(*tri, [0, 0, 0, 0, 0, 1], INF) == (5.0, 0.25, 0.5)


def test_triangle_fail_conditions(oracle):
    tri = [[-1, -1, 5], [1, -1, 5], [0, 1, 5]]
    pk = _tris72([tri] * 8)
    ray = [0, 0, 0, 0, 0, 1]
    # t > ray_length fails, t == ray_length passes the mask (the consumer's strict compare rejects it)
    assert oracle.packed_tris_intersect(pk, ray, 4.999)[3] == 0xFF
    assert oracle.packed_tris_intersect(pk, ray, 5.0)[3] == 0x00
    # behind the origin: t < 0
    assert oracle.packed_tris_intersect(pk, [0, 0, 10, 0, 0, 1], INF)[3] == 0xFF
    # outside: u < 0, v < 0, u + v > 1
    for o in ([-2, 0, 0], [0, -2, 0], [2, 2, 0]):
        assert oracle.packed_tris_intersect(pk, o + [0, 0, 1], INF)[3] == 0xFF
    # parallel ray: |a| < EPSILON
    assert oracle.packed_tris_intersect(pk, [0, 0, 0, 1, 0, 0], INF)[3] == 0xFF
    # no back-face culling: the flipped triangle is hit as well
    flipped = [tri[0], tri[2], tri[1]]
    t = oracle.packed_tris_intersect(_tris72([flipped] * 8), ray, INF)[0]
    assert np.all(t == 5.0)


def test_t_min_rules_differ_between_paths(oracle):
    """8-wide: mask accepts t >= +0 and the consumer needs t > 0 (src/main.rs:117); scalar path needs
    t > EPSILON (src/objects.rs:300)."""
    tri = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    eps = np.float32(1.1920929e-7)
    # origin just in front of the plane by less than EPSILON: scalar rejects, 8-wide accepts
    o = [0, 0, -float(eps) / 2]
    t = oracle.packed_tris_intersect(_tris72([tri] * 8), o + [0, 0, 1], INF)[0]
    assert np.all(t > 0) and np.all(t < eps)
    assert oracle.triangle_ray_hits(*tri, o + [0, 0, 1], INF) is None
    # scalar path: u must lie in [0, 1] (NaN fails), t < min is strict
    assert oracle.triangle_ray_hits(*tri, [0, 0, -2, 0, 0, 1], 2.0) is None
    assert oracle.triangle_ray_hits(*tri, [0, 0, -2, 0, 0, 1], 2.0001)[0] == 2.0


def test_packed_boxes_quirks(oracle):
    """src/simd_accel.rs:342-377: no clamp of tmin to 0, ray_length caps tmax, signum() == 1.0 swap rule."""
    lo = np.array([[1, -1, -1]] * 8, dtype=np.float32)
    hi = np.array([[2, 1, 1]] * 8, dtype=np.float32)
    boxes = np.concatenate([lo.T, hi.T], axis=0)
    # straight through
    t, m = oracle.packed_boxes_intersect(boxes, [0, 0, 0, 1, 0, 0], INF)
    assert m == 0x00 and np.all(t == 1.0)
    # box wholly behind the origin still passes (tmin is not clamped)
    assert oracle.packed_boxes_intersect(boxes, [5, 0, 0, 1, 0, 0], INF)[1] == 0x00
    # culled by the current closest distance
    assert oracle.packed_boxes_intersect(boxes, [0, 0, 0, 1, 0, 0], 0.5)[1] == 0xFF
    assert oracle.packed_boxes_intersect(boxes, [0, 0, 0, 1, 0, 0], 1.0)[1] == 0x00
    # dir component -0.0: inv = -inf, signum() == -1, the axis is swapped; +0.0 is not
    assert oracle.packed_boxes_intersect(boxes, [0, 0, 0, 1, -0.0, 0.0], INF)[1] == 0x00
    # origin on a slab plane with a zero direction component: 0 * inf = NaN.
    # _mm256_max_ps(a, b) returns b when either is NaN, so a NaN in the LAST axis reaches the compare
    # and the box is rejected (a reference false negative that the restatement keeps).
    lo2 = np.array([[1, -1, 0]] * 8, dtype=np.float32)
    hi2 = np.array([[2, 1, 1]] * 8, dtype=np.float32)
    b2 = np.concatenate([lo2.T, hi2.T], axis=0)
    assert oracle.packed_boxes_intersect(b2, [0, 0, 0, 1, 0, 0], INF)[1] == 0xFF
    # the same NaN in the FIRST axis is dropped by the max chain and the box passes
    lo3 = np.array([[0, -1, 1]] * 8, dtype=np.float32)
    hi3 = np.array([[1, 1, 2]] * 8, dtype=np.float32)
    b3 = np.concatenate([lo3.T, hi3.T], axis=0)
    assert oracle.packed_boxes_intersect(b3, [0, 0, 0, 0, 0, 1], INF)[1] == 0x00


# ---- exact-rational model of the rounding sequence --------------------------------------------
def _rnd(x):
    """Round a Fraction to the nearest binary32 value (ties to even), returned as a Fraction."""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    q = x / ulp
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return s * n * ulp


def _F(v):
    return Fraction(float(np.float32(v)))


def _model_tri(e1, e2, v0, o, d):
    """src/simd_accel.rs:136-153,288-305 with every rounding made explicit."""
    mul = lambda a, b: _rnd(a * b)
    fma = lambda a, b, c: _rnd(a * b + c)
    sub = lambda a, b: _rnd(a - b)
    cross = lambda a, b: [fma(a[1], b[2], -mul(b[1], a[2])), fma(a[2], b[0], -mul(b[2], a[0])),
                          fma(a[0], b[1], -mul(b[0], a[1]))]
    dot = lambda a, b: fma(a[2], b[2], fma(a[1], b[1], mul(a[0], b[0])))
    q = cross(d, e2)
    a = dot(e1, q)
    if a == 0:
        return None
    f = _rnd(Fraction(1) / a)
    s = [sub(o[i], v0[i]) for i in range(3)]
    u = mul(f, dot(s, q))
    r = cross(s, e1)
    v = mul(f, dot(d, r))
    t = mul(f, dot(e2, r))
    return t, u, v


def test_rounding_sequence_matches_exact_rational_model(oracle):
    rng = np.random.default_rng(12345)
    checked = 0
    for _ in range(40):
        c = rng.uniform([-10, -10, 5], [10, 10, 25], size=(8, 3))
        e1 = rng.uniform(-0.1, 0.1, size=(8, 3))
        e2 = rng.uniform(-0.1, 0.1, size=(8, 3))
        a = (c - (e1 + e2) / 3).astype(np.float32)
        b = (a + e1.astype(np.float32)).astype(np.float32)
        cc = (a + e2.astype(np.float32)).astype(np.float32)
        pk = _tris72(np.stack([a, b, cc], axis=1))
        # aim the ray at the first triangle's centroid so that some lanes are near hits
        target = (a[0] + b[0] + cc[0]) / 3
        o = rng.uniform(-1, 1, size=3).astype(np.float32)
        d = ((target - o) * rng.uniform(0.5, 2.0)).astype(np.float32)
        ray = np.concatenate([o, d]).astype(np.float32)
        t, u, v, _ = oracle.packed_tris_intersect(pk, ray, INF)
        E1, E2, V0 = pk[0:24].reshape(3, 8), pk[24:48].reshape(3, 8), pk[48:72].reshape(3, 8)
        for lane in range(8):
            m = _model_tri([_F(E1[k, lane]) for k in range(3)], [_F(E2[k, lane]) for k in range(3)],
                           [_F(V0[k, lane]) for k in range(3)], [_F(x) for x in o], [_F(x) for x in d])
            if m is None:
                continue
            assert _F(u[lane]) == m[1] and _F(v[lane]) == m[2], "u / v rounding differs from the model"
            if t[lane] != -1.0:
                assert _F(t[lane]) == m[0], "t rounding differs from the model"
            checked += 1
    assert checked > 250


def test_tie_goes_to_lowest_index_and_rest_tri_quirk(oracle):
    """Strict compare in index order (src/main.rs:117): equal t keeps the first triangle. The trailing
    n % 8 triangles never lower `closest` (src/main.rs:157-169): the last qualifying one wins."""
    tri = np.array([[-1, -1, 5], [1, -1, 5], [0, 1, 5]], dtype=np.float32)

    def obj(n, z_of=lambda i: 5.0):
        verts = np.concatenate([tri + np.array([0, 0, z_of(i) - 5.0], dtype=np.float32) for i in range(n)])
        return {"verts": verts, "tris": np.arange(3 * n, dtype=np.uint32).reshape(n, 3),
                "tri_data": np.tile(np.array([[0, 0, -1, 1, 1, 1]], dtype=np.float32), (n, 1)),
                "material": (0.3, 0.0, 1.0)}

    ray = np.array([[0, 0, 0, 0, 0, 1]], dtype=np.float32)
    h = oracle.OracleScene([obj(8)]).trace_closest(ray)
    assert h["tri"][0] == 0 and h["t"][0] == 5.0
    # three trailing triangles at z = 5, 7, 6: every one has t < closest (= inf), the last wins,
    # and the returned distance stays inf
    h = oracle.OracleScene([obj(3, lambda i: [5.0, 7.0, 6.0][i])]).trace_closest(ray)
    assert h["tri"][0] == 2 and h["t"][0] == 6.0 and np.isinf(h["closest"][0])
    # a later object still compares against the stale distance and replaces the hit even though it
    # is farther away
    far8 = obj(8, lambda i: 9.0)
    h = oracle.OracleScene([obj(3, lambda i: 5.0), far8]).trace_closest(ray)
    assert h["tri"][0] == 3 and h["t"][0] == 9.0


def test_shadow_search_accepts_t_zero(oracle):
    """Any lane with a clear sign bit occludes (src/main.rs:199-204): no t > 0 re-check."""
    tri = np.array([[-1, -1, 0], [0, 1, 0], [1, -1, 0]], dtype=np.float32)  # this winding gives t = +0
    o8 = {"verts": np.tile(tri, (8, 1)), "tris": np.arange(24, dtype=np.uint32).reshape(8, 3),
          "tri_data": np.tile(np.array([[0, 0, 1, 1, 1, 1]], dtype=np.float32), (8, 1)), "material": (0.3, 0.0, 1.0)}
    sc = oracle.OracleScene([o8])
    rays = np.array([[0, -0.5, 0, 0, 0, 1], [0, 0, -1, 0, 0, 1], [0, 0, 1, 0, 0, 1]], dtype=np.float32)
    hits = sc.trace_closest(rays)
    occ = sc.trace_any(rays)
    # the ray that starts in the plane: t = +0 -> no nearest hit, but an occluder
    t0 = oracle.packed_tris_intersect(_tris72([tri] * 8), rays[0], INF)[0][0]
    assert t0 == 0.0 and not np.signbit(t0)
    assert hits["tri"][0] == 0xFFFFFFFF and occ[0] == 1
    assert hits["tri"][1] == 0 and occ[1] == 1
    assert hits["tri"][2] == 0xFFFFFFFF and occ[2] == 0


def test_counter_rng_is_uniform_and_keyed(oracle):
    v = np.array([oracle.rand(7, i, k, w) for i in range(200) for k in range(5) for w in range(2)])
    assert v.min() >= 0.0 and v.max() < 1.0 and abs(v.mean() - 0.5) < 0.03
    assert oracle.rand(7, 1, 0, 0) != oracle.rand(8, 1, 0, 0) != oracle.rand(7, 1, 0, 1)
    assert oracle.rand(7, 1, 0, 0) == oracle.rand(7, 1, 0, 0)


def test_injected_sincos_is_accurate_and_exact_at_the_axes(oracle):
    """The sine / cosine of the scatter angle that oracle and kernels share in place of libm
    (src/main.rs:264-268): within 1.2e-7 of the true values over the draws the generator can make,
    on the unit circle, exact at psi = 0."""
    assert oracle.sincos(0.0) == (np.float32(0.0), np.float32(1.0))
    worst = 0.0
    for k in range(0, 1 << 24, 4099):
        u = np.float32(k) * np.float32(1.0 / 16777216.0)
        psi = np.float32(np.float32(u * np.float32(2.0)) * np.float32(3.14159265358979323846))
        s, c = oracle.sincos(psi)
        worst = max(worst, abs(float(s) - np.sin(float(psi))), abs(float(c) - np.cos(float(psi))))
        assert abs(float(s) ** 2 + float(c) ** 2 - 1.0) < 4e-7
    assert worst < 1.2e-7
