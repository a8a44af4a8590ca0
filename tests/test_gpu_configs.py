"""The benchmark configurations at their full sizes, through the C ABI, against the CPU oracle.

The oracle scans every 8-triangle leaf box per ray (as the reference does), so at a million and
ten million triangles it is run on strided subsets of the frames' pixels; the rest of each frame
is covered by properties that do not depend on its size (determinism, sky pixels, pipeline ==
one-kernel frame, sharded == whole). Bar as everywhere: hit ids agree on >= 99.999 % of the rays
compared, t / u / v bit for bit where they agree, colours within 1 LSB.
"""
import numpy as np
import pytest

from raytracing_amd import RT_MISS, Scene, render_params, scenes
from raytracing_amd import distributed as rd

pytestmark = pytest.mark.gpu

SKY = np.array([0x87, 0xCE, 0xEB, 0xFF], dtype=np.uint8)


def _compare_hits(gpu, ref, max_frac=1e-5):
    same = gpu["tri"] == ref["tri"]
    frac = 1.0 - same.mean()
    assert frac <= max_frac, f"{(~same).sum()} of {len(ref)} hit ids differ ({frac:.2e})"
    hit = same & (ref["tri"] != RT_MISS)
    for f in ("t", "u", "v"):
        assert np.array_equal(gpu[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f"{f} differs in bits"
    return int((~same).sum())


def _shadow_rays(rays, hits, tri_data, sun):
    """The shadow rays trace_ray casts from nearest hits (src/main.rs:121-124,176-189), in f32:
    origin = ray.at(t) + normal * 1e-5, direction = normalize(sun - origin); only where
    normal . direction > 0."""
    hit = np.flatnonzero(hits["tri"] != RT_MISS)
    o, d, t = rays[hit, 0:3], rays[hit, 3:6], hits["t"][hit, None]
    n = tri_data[hits["tri"][hit], 0:3]
    pos = ((o + d * t).astype(np.float32) + (n * np.float32(0.00001)).astype(np.float32)).astype(np.float32)
    v = (np.asarray(sun, dtype=np.float32)[None, :] - pos).astype(np.float32)
    dot = ((v[:, 0] * v[:, 0]).astype(np.float32) + v[:, 1] * v[:, 1]).astype(np.float32) + v[:, 2] * v[:, 2]
    sd = (v * (np.float32(1.0) / np.sqrt(dot.astype(np.float32)))[:, None]).astype(np.float32)
    lit = ((n[:, 0] * sd[:, 0]).astype(np.float32) + n[:, 1] * sd[:, 1] + n[:, 2] * sd[:, 2]) > 0
    return np.concatenate([pos[lit], sd[lit]], axis=1).astype(np.float32)


def _scatter_rays(rays, hits, rng, n):
    """Rays leaving nearest hits in random directions: the incoherent rays of the diffuse bounce."""
    hit = np.flatnonzero(hits["tri"] != RT_MISS)
    pick = rng.choice(hit, size=min(n, len(hit)), replace=False)
    o = rays[pick, 0:3] + rays[pick, 3:6] * hits["t"][pick, None]
    d = rng.normal(size=(len(pick), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([(o + d * np.float32(1e-3)).astype(np.float32), d.astype(np.float32)], axis=1).astype(np.float32)


def _strided_frame_check(rgba, ids, ref_rgba, ref_ids, pix, spp):
    """Hit ids and colours of the pixels `pix` of a frame against the oracle's strided render."""
    sub = ids.reshape(-1, spp)[pix]
    ref = ref_ids.reshape(-1, spp)[pix]
    same = sub == ref
    assert 1.0 - same.mean() <= 1e-5, f"{(~same).sum()} of {same.size} primary ids differ"
    ok = same.all(axis=1)
    d = np.abs(rgba.reshape(-1, 4)[pix][ok].astype(np.int16) - ref_rgba.reshape(-1, 4)[pix][ok].astype(np.int16))
    assert d.max() <= 1, f"{(d > 1).sum()} channels differ by more than 1 LSB (max {d.max()})"
    return d


# ---- config 3: soup-1M, 3840 x 2160, camera + shadow rays (the benchmark frame) -----------------
def test_config3_soup_1m_full_frame(ctx, oracle, monkeypatch):
    objs = scenes.soup(1_000_000, 0x5EED0001)
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    W, H = 3840, 2160
    kw = dict(reflect_depth=1, diffuse_depth=1)
    p = render_params(W, H, 2, **kw)
    rgba, ids = g.render(p, want_hit_ids=True)
    assert np.array_equal(g.render(p), rgba), "frame is not deterministic"
    assert (rgba[..., 3] == 255).all()
    hit = ids != RT_MISS
    assert 0.5 < hit.mean() < 1.0 and ids[hit].max() < 1_000_000
    sky = ~hit.reshape(-1, 4).any(axis=1)
    assert sky.any() and (rgba.reshape(-1, 4)[sky] == SKY).all()
    # Every 157th pixel (52,831 pixels, 211,324 camera rays and their shadow rays) through the oracle's
    # own frame driver: ids and colours of the frame as rendered.
    stride = 157
    pix = np.arange(0, W * H, stride)
    ref_rgba, ref_ids, n_rays = o.render(W, H, 2, pixel_stride=stride, want_hit_ids=True, **kw)
    assert n_rays > 4 * len(pix)
    _strided_frame_check(rgba, ids, ref_rgba, ref_ids, pix, 4)
    # The same camera rays and the shadow rays off the oracle's hits through the batch calls.
    rays = scenes.camera_rays(W, H, 2, pixels=pix)
    ref = o.trace_closest(rays)
    assert np.array_equal(ref["tri"], ref_ids.reshape(-1, 4)[pix].reshape(-1))
    assert _compare_hits(g.trace_closest(rays), ref) <= 2
    sh = _shadow_rays(rays, ref, objs[0]["tri_data"], scenes.SUN_T0)
    assert len(sh) > 50_000
    occ_ref, occ = o.trace_any(sh), g.trace_any(sh)
    assert (occ_ref != occ).mean() <= 1e-5, f"{(occ_ref != occ).sum()} of {len(sh)} shadow results differ"
    assert 0.2 < occ_ref.mean() < 0.98
    # the tree-walk frame (camera rays by beam walk instead of by projection) is the same frame
    monkeypatch.setenv("RAYS_HIP_RASTER", "0")
    rgba_walk, ids_walk = g.render(p, want_hit_ids=True)
    monkeypatch.delenv("RAYS_HIP_RASTER")
    assert np.array_equal(ids_walk, ids) and np.array_equal(rgba_walk, rgba)
    g.close()
    o.close()


# ---- configs 4 and 5: soup-10M -------------------------------------------------------------------
@pytest.fixture(scope="module")
def soup10m(ctx, oracle):
    objs = scenes.soup(10_000_000, 0x5EED0002)
    g = Scene(ctx, objs)
    o = oracle.OracleScene(objs)
    yield objs, g, o
    g.close()
    o.close()


def test_config4_soup_10m_diffuse_bounces(soup10m, monkeypatch):
    """3840 x 2160, four samples, ten scatter rays off every hit and the shadow rays of all of them."""
    import torch

    objs, g, o = soup10m
    W, H = 3840, 2160
    kw = dict(reflect_depth=4, diffuse_depth=2, diffuse_scatt=10, seed=4)
    p = render_params(W, H, 2, **kw)
    rgba, ids = g.render(p, want_hit_ids=True)
    hit = ids != RT_MISS
    assert hit.mean() > 0.9 and ids[hit].max() < 10_000_000
    sky = ~hit.reshape(-1, 4).any(axis=1)
    if sky.any():
        assert (rgba.reshape(-1, 4)[sky] == SKY).all()
    # deterministic
    frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    g.render_device(p, frame.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), rgba), "frame is not deterministic"
    # The pipeline of kernels against the one-kernel frame on a strided subset of the tiles: shard 5 of
    # 61 owns every 61st diagonal of tiles (134 of 8,160).
    part = g.render(render_params(W, H, 2, shard_index=5, shard_count=61, **kw))
    own = part[..., 3] == 255
    assert 0.01 < own.mean() < 0.03 and np.array_equal(part[own], rgba[own])
    monkeypatch.setenv("RAYS_HIP_MEGAKERNEL", "1")
    one = g.render(render_params(W, H, 2, shard_index=5, shard_count=61, **kw))
    monkeypatch.delenv("RAYS_HIP_MEGAKERNEL")
    assert np.array_equal(one, part), f"{(one != part).sum()} bytes differ from the one-kernel frame"
    # Strided pixels through the oracle's own frame driver, bounces and all (every 509th pixel:
    # 16,297 pixels, about 1.4 million rays over ten million triangles).
    stride = 509
    pix = np.arange(0, W * H, stride)
    ref_rgba, ref_ids, n_rays = o.render(W, H, 2, pixel_stride=stride, want_hit_ids=True, **kw)
    assert n_rays > 20 * len(pix)
    _strided_frame_check(rgba, ids, ref_rgba, ref_ids, pix, 4)
    # Camera rays and scatter rays through the batch call.
    pix = np.arange(0, W * H, 1999)
    rays = scenes.camera_rays(W, H, 2, pixels=pix)
    ref = o.trace_closest(rays)
    _compare_hits(g.trace_closest(rays), ref)
    sc = _scatter_rays(rays, ref, np.random.default_rng(4), 16000)
    ref_sc = o.trace_closest(sc)
    _compare_hits(g.trace_closest(sc), ref_sc)
    assert 0.3 < (ref_sc["tri"] != RT_MISS).mean() < 1.0
    assert np.array_equal(g.trace_any(sc), o.trace_any(sc))


def test_config5_soup_10m_8k_16spp_sharded(soup10m, ctx):
    """7680 x 4320, 4 x 4 samples per pixel, rendered whole and as the 8 shards of an 8-GPU node: the
    frame assembled from the shards' tiles is the whole frame; strided ids against the oracle."""
    import torch

    objs, g, o = soup10m
    W, H, n = 7680, 4320, 8
    kw = dict(reflect_depth=1, diffuse_depth=1)
    p = render_params(W, H, 4, **kw)
    whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    d_ids = torch.full((W * H * 16,), -1, dtype=torch.int32, device="cuda")
    g.render_device(p, whole.data_ptr(), d_ids.data_ptr())
    slab = rd.slab_tiles(W, H, n)
    gathered = torch.zeros((n, slab * rd.TILE_BYTES), dtype=torch.uint8, device="cuda")
    for r in range(n):
        g.render_device(render_params(W, H, 4, shard_index=r, shard_count=n, **kw), gathered[r].data_ptr())
    out = torch.zeros_like(whole)
    ctx.deinterleave_device(gathered.data_ptr(), slab, W, H, n, out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(out, whole), "the frame assembled from 8 shards differs from the whole frame"
    assert bool((whole[..., 3] == 255).all())
    # every 2,039th pixel: 16,273 pixels, 260,368 camera rays and their shadow rays
    stride = 2039
    pix = np.arange(0, W * H, stride)
    ref_rgba, ref_ids, _ = o.render(W, H, 4, pixel_stride=stride, want_hit_ids=True, **kw)
    pix_t = torch.from_numpy(pix).cuda()
    ids = d_ids.view(-1, 16)[pix_t].cpu().numpy().view(np.uint32)
    rgba = whole.view(-1, 4)[pix_t].cpu().numpy()
    ref = ref_ids.reshape(-1, 16)[pix]
    same = ids == ref
    assert 1.0 - same.mean() <= 1e-5, f"{(~same).sum()} of {same.size} primary ids differ"
    ok = same.all(axis=1)
    d = np.abs(rgba[ok].astype(np.int16) - ref_rgba.reshape(-1, 4)[pix][ok].astype(np.int16))
    assert d.max() <= 1
    del d_ids, whole, out, gathered
    torch.cuda.empty_cache()
