#!/usr/bin/env python3
"""Benchmark of the trace_ray hot path on MI355X.

Workload (the configuration the headline metric is quoted on): the synthetic
1,000,000-triangle soup (PCG32 seed 0x5EED0001), one 3840x2160 frame at four samples per pixel:
camera rays generated on the device, nearest-hit search, one sun shadow ray per lit hit, shading,
sub-sample average, RGBA8. One step = one frame. Mrays/s counts nearest-hit plus shadow searches.
In this configuration the library finds the camera rays' nearest hits by projecting the triangles
onto the sample grid (same results as the tree walk); `value` is that default path and
`value_tree_walk` the same frame with every ray walked through the tree (RAYS_HIP_RASTER=0).
A short run of the diffuse-bounce workload (soup-10M, config 4) adds a roofline entry for the
kernel that traces the incoherent secondary rays.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 the same frame is split into 32x32-pixel tiles dealt out diagonally to the ranks
(every rank holds the whole scene) and rank 0 collects the tiles with one RCCL gather per frame,
overlapped with the next frame's rendering ("scaling": "strong"). The gather is the library's own
rt_gather_tiles_device (grouped ncclSend / ncclRecv into rank 0 + tile scatter) on a communicator
made by rt_comm_create.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK = 157.3e12   # vector FP32, FLOP/s
HBM_PEAK = 8.0e12      # B/s
FLOP_BOX, FLOP_TRI = 18, 46          # per slab test / per triangle test
BYTES_BOX, BYTES_TRI = 32, 36        # structure bytes per slab test / per triangle test


def cpu_baseline(objects, width, height, target_seconds=20.0):
    """Times the CPU oracle (a C++ restatement of the reference's linear packet scan, AVX2/FMA)
    on a strided pixel subsample of the same frame. Reported next to the GPU number; never part
    of the measured path."""
    from oracle import oracle_ffi as O

    cores = O.default_threads()
    sc = O.OracleScene(objects)
    kw = dict(grid=2, reflect_depth=1, diffuse_depth=1)
    stride = 8192
    t0 = time.perf_counter()
    _, _, rays = sc.render(width, height, pixel_stride=stride, n_threads=cores, **kw)
    dt = time.perf_counter() - t0
    rate = rays / dt
    # Scale the sample so that the timed run takes about target_seconds.
    want = max(rays, int(rate * target_seconds))
    stride = max(1, int(stride * rays / want))
    t0 = time.perf_counter()
    _, _, rays = sc.render(width, height, pixel_stride=stride, n_threads=cores, **kw)
    dt = time.perf_counter() - t0
    sc.close()
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"every {stride}th pixel of the {width}x{height} frame, 4 samples each: {rays} rays "
                  f"(nearest-hit + shadow) in {dt:.1f} s; C++ restatement of the reference's AVX2 linear scan, "
                  f"built with g++ -O3 -mavx2 -mfma -ffp-contract=off (not -march=native: it is built on "
                  f"one machine and run on another), {cores} threads over image rows",
    }


def teapot_leg(ctx, args, with_cpu):
    """The reference's own scene (src/main.rs:406-448: two teapots and a two-triangle floor, 12,642
    triangles, sun frozen at t = 0) at 1920x1080 with four samples a pixel, camera + shadow rays
    (config 2 of the metric), timed like the headline: K frames between synchronisations, one frame at a
    time. Beside it the same scene with the reference's full constants at its own 800x800 (mirror depth 4,
    ten scatter rays off every camera hit), and the CPU oracle on the whole config-2 frame."""
    import torch

    from raytracing_amd import Scene, render_params, scenes

    objects = scenes.default_scene()
    scene = Scene(ctx, objects)
    sh = torch.cuda.current_stream().cuda_stream
    out = {"workload": "the reference's default scene (two teapots + floor, 12,642 triangles, sun at t = 0), 1920x1080, "
                       "4 samples/pixel, primary + sun-shadow rays, shade + resolve to RGBA8 on device"}

    def timed(p, steps):
        frame = torch.zeros((p.height, p.width, 4), dtype=torch.uint8, device="cuda")
        st = scene.render_device_stats(p, frame.data_ptr())
        for _ in range(3):
            scene.render_device(p, frame.data_ptr(), None, sh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            scene.render_device(p, frame.data_ptr(), None, sh)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return st, dt

    W, H = 1920, 1080
    st, dt = timed(render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0), args.steps)
    out.update({"mrays_per_s": st["rays"] / dt / 1e6, "ms_per_frame": dt * 1e3, "rays_per_frame": st["rays"],
                "camera_rays_by": "projection" if st["camera_by_projection"] else "beam walk",
                "shadow_rays_by": "light-space cells" if st["shadow_by_cells"] else "beam walk",
                "box_tests_per_ray": st["box_tests"] / max(1, st["rays"]), "tri_tests_per_ray": st["tri_tests"] / max(1, st["rays"])})
    os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "2"
    _, dt2 = timed(render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0), args.steps)
    os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
    out["two_frames_in_flight"] = {"mrays_per_s": st["rays"] / dt2 / 1e6, "ms_per_frame": dt2 * 1e3}
    st8, dt8 = timed(render_params(800, 800, 2, reflect_depth=4, diffuse_depth=2, diffuse_scatt=10, sun=scenes.SUN_T0, seed=1),
                     max(3, args.steps // 4))
    out["reference_constants_800x800"] = {"mrays_per_s": st8["rays"] / dt8 / 1e6, "ms_per_frame": dt8 * 1e3,
                                          "rays_per_frame": st8["rays"], "frames_per_s": 1.0 / dt8}
    scene.close()
    if with_cpu:
        from oracle import oracle_ffi as O

        cores = O.default_threads()
        sc = O.OracleScene(objects)
        t0 = time.perf_counter()
        _, _, rays = sc.render(W, H, grid=2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0, n_threads=cores)
        dtc = time.perf_counter() - t0
        sc.close()
        out["cpu_oracle"] = {"mrays_per_s": rays / dtc / 1e6, "cores": cores, "kind": "port",
                             "sample": f"the whole {W}x{H} frame, 4 samples a pixel: {rays} rays in {dtc:.1f} s"}
        out["gpu_over_cpu"] = out["mrays_per_s"] / out["cpu_oracle"]["mrays_per_s"]
    return out


def config5_leg(ctx, args, rank, world, comm, dist, red_dev, scene=None):
    """Config 5 of the metric: soup-10M at 7680x4320 with 4 x 4 samples a pixel, camera + shadow rays, the
    frame's tiles dealt out to the ranks and gathered on rank 0 (the configuration the tile split was
    designed for: its per-ray work dwarfs a rank's fixed set-up). K // 4 frames between fences."""
    import torch

    from raytracing_amd import Scene, render_params, scenes
    from raytracing_amd import distributed as rd

    W, H = 7680, 4320
    own = scene is None
    if own:
        scene = Scene(ctx, scenes.soup(args.bounce_tris, 0x5EED0002))
    p = render_params(W, H, 4, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0, shard_index=rank, shard_count=world)
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    if world == 1:
        bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)]
        full = None
    else:
        slab_bytes = rd.slab_tiles(W, H, world) * rd.TILE_BYTES
        bufs = [torch.zeros(slab_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        full = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    st = scene.render_device_stats(p, bufs[0].data_ptr())
    rays = torch.tensor([st["rays"]], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(rays)

    def frame(i):
        b = bufs[i % len(bufs)]
        scene.render_device(p, b.data_ptr(), None, sh)
        if world > 1 and comm is not None:
            comm.gather_tiles_device(b.data_ptr(), W, H, full.data_ptr() if rank == 0 else None, sh)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    steps = max(3, args.steps // 4)
    for i in range(2):
        frame(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        frame(i)
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item()) / steps
    if own:
        scene.close()
    return {"workload": f"soup-{args.bounce_tris} seed 0x5eed0002, {W}x{H}, 16 samples/pixel, primary + sun-shadow rays, "
                        f"tiles over {world} rank(s)" + (", gathered on rank 0 by rt_gather_tiles_device" if world > 1 and comm is not None else ""),
            "value": int(rays.item()) / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt * 1e3, "steps": steps,
            "rays_per_frame": int(rays.item()), "n_gpus": world}


TRAFFIC_PROFILE = "profiles/r03_pmc_traffic.json"
TRAFFIC_NAMES = {"k_shadow_cells": "k_shadow_cells<false, true>", "k_raster_tile": "k_raster_tile<false>",
                 "k_wf_shadow": "k_wf_shadow<false>", "k_wf_bounce": "k_wf_bounce<false>"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the counter passes kept with the repository, or null."""
    out = {"traffic": None, "traffic_profile": TRAFFIC_PROFILE + " (rocprofv3 --pmc passes of this command, tools/measure.sh)"}
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), TRAFFIC_PROFILE)) as f:
            k = json.load(f)["kernels"].get(TRAFFIC_NAMES.get(kernel, kernel))
        if k:
            out["traffic"] = k["bytes_per_launch"]
            out["traffic_fetch_doubled"] = k["bytes_per_launch_fetch_doubled"]
    except (OSError, ValueError, KeyError):
        pass
    return out


def scatter_order_probe(scene, objects):
    """Sorted against unsorted scatter rays: ten cosine-weighted rays off every camera-ray hit of a
    960 x 540 x 4 frame of the diffuse-bounce scene, traced by the batch nearest-hit call (the same
    persistent per-lane walk as k_wf_bounce) in the frame's own order, sorted by origin cell and direction
    octant, and shuffled. The rates are the walk's alone: the sort that would feed it is not in them
    (torch does it here; in a frame it would be a radix sort of the keys plus a gather of 24 bytes a ray)."""
    import torch

    dev, W, H = "cuda", 960, 540
    sh = torch.cuda.current_stream().cuda_stream
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    blk = ((ys // 8) * ((W + 7) // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8     # pixels in 8 x 8 blocks, as a frame has them
    order = torch.argsort(blk.reshape(-1))
    px = xs.reshape(-1)[order].repeat_interleave(4).float()
    py = ys.reshape(-1)[order].repeat_interleave(4).float()
    smp = torch.arange(4, device=dev).repeat(W * H)
    ox = torch.where(smp % 2 == 1, 0.75, 0.25)
    oy = torch.where((smp == 1) | (smp == 2), 0.75, 0.25)
    d = torch.stack([(2 * (px + ox) / W - 1) * float(W // H), 1 - 2 * (py + oy) / H, torch.ones_like(px)], 1)
    rays = torch.cat([torch.zeros_like(d), d], 1).contiguous()
    n0 = rays.shape[0]
    out = torch.zeros(n0 * 4, dtype=torch.int32, device=dev)
    scene.trace_closest_device(rays.data_ptr(), n0, out.data_ptr(), sh)
    torch.cuda.synchronize()
    o4 = out.view(n0, 4)
    hit = o4[:, 0] != -1
    t = o4[:, 1].view(torch.float32)
    normals = torch.from_numpy(np.ascontiguousarray(objects[0]["tri_data"][:, :3])).to(dev)
    nrm = normals[o4[hit, 0].long()]            # a soup's hit ids are its input order
    pos = d[hit] * t[hit, None] + nrm * 1e-5
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    nn, pp = nrm.repeat_interleave(10, 0), pos.repeat_interleave(10, 0)
    u1 = torch.rand(nn.shape[0], device=dev, generator=g)
    u2 = torch.rand(nn.shape[0], device=dev, generator=g)
    st_, ct, psi = torch.sqrt(u1), torch.sqrt(1 - u1), 2 * np.pi * u2
    arb = torch.where(nn[:, 2:3].abs() < 1e-6, torch.tensor([0.0, 0.0, 1.0], device=dev), torch.tensor([1.0, 0.0, 0.0], device=dev))
    t1 = torch.nn.functional.normalize(torch.cross(nn, arb, dim=1), dim=1)
    t2 = torch.nn.functional.normalize(torch.cross(nn, t1, dim=1), dim=1)
    dd = torch.nn.functional.normalize((st_ * torch.cos(psi))[:, None] * t1 + (st_ * torch.sin(psi))[:, None] * t2 + ct[:, None] * nn, dim=1)
    n = dd.shape[0]

    def spread3(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        return (v | (v << 2)) & 0x9249249

    q = torch.clamp(((pp - torch.tensor([-10.0, -10.0, 5.0], device=dev)) / 20.0 * 1024).long(), 0, 1023)
    morton = spread3(q[:, 0]) | (spread3(q[:, 1]) << 1) | (spread3(q[:, 2]) << 2)
    octant = (dd[:, 0] > 0).long() | ((dd[:, 1] > 0).long() << 1) | ((dd[:, 2] > 0).long() << 2)
    orders = {"frame_order": torch.arange(n, device=dev),
              "sorted_by_origin_cell_and_octant": torch.argsort(((morton >> 15) << 3) | octant, stable=True),
              "shuffled": torch.randperm(n, device=dev, generator=g)}
    res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    rates = {}
    for name, idx in orders.items():
        r = torch.cat([pp[idx], dd[idx]], 1).contiguous()
        scene.trace_closest_device(r.data_ptr(), n, res.data_ptr(), sh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            scene.trace_closest_device(r.data_ptr(), n, res.data_ptr(), sh)
        torch.cuda.synchronize()
        rates[name] = n / ((time.perf_counter() - t0) / 3) / 1e6
    return {"scatter_rays": n, "unit": "Mrays/s (batch nearest-hit call, the walk alone)", **rates,
            "note": "k_wf_bounce traces its rays in frame order: the gain of sorted rays is smaller than a sort of the rays would cost"}


def bounce_roofline(ctx, args, new_event, elapsed_ms, scene):
    """A short run of the diffuse-bounce workload (config 4: soup-10M, 3840x2160, four samples, ten
    scatter rays off every hit and the shadow rays of all of them): frame time, and the roofline
    of k_wf_bounce, the persistent per-lane walk that traces the scatter rays. One launch = one level
    of a frame's secondary rays; this scene has no mirrors, so one launch per frame."""
    import torch

    from raytracing_amd import Scene, render_params, scenes

    W, H = args.width, args.height
    # One run of tiles for the whole frame (the default workspace limit would cut this frame in two, and
    # the events around the bounce launches cover one run).
    os.environ.setdefault("RAYS_HIP_BOUNCE_RAYS", str(-(-W // 32) * -(-H // 32) * 1024 * 4 * 10))
    p = render_params(W, H, 2, reflect_depth=4, diffuse_depth=2, diffuse_scatt=10, sun=scenes.SUN_T0, seed=1)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    sh = torch.cuda.current_stream().cuda_stream
    st = scene.render_device_stats(p, frame.data_ptr())     # instrumented build: counters of the bounce launches
    scene.render_device(p, frame.data_ptr(), None, sh)
    torch.cuda.synchronize()
    n = 3
    events = [[new_event() for _ in range(8)] for _ in range(n)]
    t0 = time.perf_counter()
    for i in range(n):
        scene.render_device_events(p, frame.data_ptr(), [e.value for e in events[i]], None, sh)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    bounce_ms = float(np.mean([elapsed_ms(ev[5], ev[6]) for ev in events]))
    sizes = scene.sizes()
    # Unit = one scatter ray; one launch = the scatter rays of one frame (st["bounce_rays"] of them,
    # out of samples * 10 ray slots: samples that hit nothing cast none). Algorithmic bytes of a
    # launch: per slot a 24-byte record out (hit point + id 16, state 4, sort key 4), per sample
    # its 20-byte record in, plus the 4-wide tree (in the 64-byte form of its nodes that this walk reads:
    # half of sizes["nodes4"]), the triangle records and the normals / colours once.
    samples = W * H * 4
    slots = samples * 10
    nodes_compact = sizes["nodes4"] // 2
    alg_bytes = 24 * slots + 20 * samples + nodes_compact + sizes["recs"] + sizes["tri_data"]
    flops = FLOP_BOX * st["bounce_box_tests"] + FLOP_TRI * st["bounce_tri_tests"]
    sec = bounce_ms * 1e-3
    return {
        "workload": f"soup-{args.bounce_tris} seed 0x5eed0002, {W}x{H}, 4 samples/pixel, 10 scatter rays per hit + shadow rays",
        "frame_ms": dt * 1e3,
        "rays_per_frame": st["rays"],
        "mrays_per_s": st["rays"] / dt / 1e6,
        "bound": "hbm",
        "kernel": "k_wf_bounce",
        "kernel_ms": bounce_ms,
        "scatter_rays": st["bounce_rays"],
        "scatter_mrays_per_s": st["bounce_rays"] / sec / 1e6,
        "box_tests_per_ray": st["bounce_box_tests"] / max(1, st["bounce_rays"]),
        "tri_tests_per_ray": st["bounce_tri_tests"] / max(1, st["bounce_rays"]),
        "achieved": alg_bytes / sec / 1e9,
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": alg_bytes / sec / HBM_PEAK,
        **pmc_traffic("k_wf_bounce"),
        "alg_bytes_per_launch": alg_bytes,
        "alg_bytes_parts": {"samples": 20 * samples, "ray_slots": 24 * slots, "nodes4_compact": nodes_compact, "recs": sizes["recs"],
                            "tri_data": sizes["tri_data"]},
        "requested_bytes_per_launch": st["bounce_fetch_bytes"],
        "requested_tb_per_s": st["bounce_fetch_bytes"] / sec / 1e12,
        "valu_tflops": flops / sec / 1e12,
        "valu_frac": flops / sec / FP32_PEAK,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the tree-walk and diffuse-bounce legs")
    ap.add_argument("--bounce-tris", type=int, default=10_000_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import Comm, Context, Scene, comm_id_create, render_params, scenes
    from raytracing_amd import distributed as rd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 needs torch.distributed.run with N processes")
        args.gpus = world
    # RAYS_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks
    # share the visible GPUs and the per-frame gather is staged through host memory. Never a result.
    backend = os.environ.get("RAYS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # The library keeps two camera + shadow frames in flight by default (a frame's memory-bound phases
    # overlap the next frame's compute-bound ones). The headline is timed with one frame at a time, so
    # that ms_per_step is also a frame's latency and the phase and kernel times stand for themselves;
    # the throughput with two in flight is timed separately below ("two_frames_in_flight").
    os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
    W, H = args.width, args.height
    objects = scenes.soup(args.tris, args.seed)
    ctx = Context(dev_index)
    t0 = time.perf_counter()
    scene = Scene(ctx, objects)
    build_s = time.perf_counter() - t0
    # The library's own RCCL communicator for the per-frame gather: rank 0 makes the id, the others
    # get it over the process group, every rank joins.
    comm = None
    ranks_seen = 1
    if world > 1 and backend == "nccl":
        # If the library cannot bring up its communicator on every rank (RCCL not loadable, say), the
        # ranks agree to gather through torch.distributed instead; the bench line says which it was.
        try:
            box = [comm_id_create() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: rt_comm_id_create failed: {e}", file=sys.stderr)
            box = [None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                comm = Comm(ctx, box[0], rank, world)
                ranks_seen = comm.ranks()[1]
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: rank {rank}: rt_comm_create failed: {e}", file=sys.stderr)
                comm = None
        agreed = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0 and comm is not None:
            comm.close()
            comm = None

    p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0,
                      shard_index=rank, shard_count=world)
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    if world == 1:
        frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)]
        slab_bytes = 0
    else:
        slab = rd.slab_tiles(W, H, world)
        slab_bytes = slab * rd.TILE_BYTES
        frames = [torch.zeros(slab_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        full = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        comm_stream = torch.cuda.Stream(device=dev)
        render_done = [torch.cuda.Event() for _ in range(2)]
        gather_done = [torch.cuda.Event() for _ in range(2)]

    # Counters of one frame from the instrumented build (not timed).
    st = scene.render_device_stats(p, frames[0].data_ptr())
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    counts = torch.tensor([st["rays"], st["box_tests"], st["tri_tests"]], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(counts)
    rays_total, box_total, tri_total = [int(x) for x in counts.tolist()]

    # Raw HIP events around the frame's phases, recorded by the library on the stream it launches
    # on: [0] start, [1] camera-ray kernel done, [2] sort done, [3] shadow kernel done, [4] resolve done.
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")

    def new_event():
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def elapsed_ms(a, b):
        ms = ctypes.c_float()
        assert hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value

    phase_events = [[new_event() for _ in range(8)] for _ in range(args.steps)]

    def step(i, timed_index=None):
        b = i % 2
        if world == 1:
            if timed_index is not None:
                scene.render_device_events(p, frames[0].data_ptr(), [e.value for e in phase_events[timed_index]], None, sh)
            else:
                scene.render_device(p, frames[0].data_ptr(), None, sh)
            return
        # The buffer must not be overwritten before its previous gather has read it.
        stream.wait_event(gather_done[b])
        if timed_index is not None:
            scene.render_device_events(p, frames[b].data_ptr(), [e.value for e in phase_events[timed_index]], None, sh)
        else:
            scene.render_device(p, frames[b].data_ptr(), None, sh)
        render_done[b].record(stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(render_done[b])
            if comm is not None:
                comm.gather_tiles_device(frames[b].data_ptr(), W, H, full.data_ptr() if rank == 0 else None,
                                         comm_stream.cuda_stream)
            elif backend == "nccl":
                gathered, _ = rd.gather_slabs(frames[b], slab_bytes, dst=0)
                if rank == 0:
                    ctx.deinterleave_device(gathered.data_ptr(), slab, W, H, world, full.data_ptr(),
                                            comm_stream.cuda_stream)
                    gathered.record_stream(comm_stream)
            else:
                host, _ = rd.gather_slabs(frames[b].cpu(), slab_bytes, dst=0)
                if rank == 0:
                    gathered = host.to(dev)
                    ctx.deinterleave_device(gathered.data_ptr(), slab, W, H, world, full.data_ptr(),
                                            comm_stream.cuda_stream)
                    gathered.record_stream(comm_stream)
            gather_done[b].record(comm_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, i)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # The same K steps with two frames in flight inside the library (an extra leg: --no-extras, which the
    # profiled runs use, leaves it out, so that a kernel's average duration there is that of the headline).
    elapsed_in_flight = None
    if not args.no_extras:
        os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "2"
        for i in range(max(args.warmup, 2)):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        fence()
        t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        elapsed_in_flight = float(t2.item())
        os.environ["RAYS_HIP_FRAMES_IN_FLIGHT"] = "1"
    # The same frame with every camera ray walked through the tree (no projection), timed the same
    # way: K steps between fences. Single GPU only, after the headline run.
    tree_walk = None
    if world == 1 and not args.no_extras:
        os.environ["RAYS_HIP_RASTER"] = "0"
        os.environ["RAYS_HIP_SHADOW"] = "walk"
        st_walk = scene.render_device_stats(p, frames[0].data_ptr())
        walk_events = [[new_event() for _ in range(8)] for _ in range(args.steps)]
        for i in range(args.warmup):
            scene.render_device(p, frames[0].data_ptr(), None, sh)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            scene.render_device_events(p, frames[0].data_ptr(), [e.value for e in walk_events[i]], None, sh)
        fence()
        dt_walk = time.perf_counter() - t0
        del os.environ["RAYS_HIP_RASTER"]
        del os.environ["RAYS_HIP_SHADOW"]
        walk_ms = [float(np.mean([elapsed_ms(ev[k], ev[k + 1]) for ev in walk_events])) for k in range(4)]
        tree_walk = {
            "value": st_walk["rays"] * args.steps / dt_walk / 1e6,
            "ms_per_step": dt_walk / args.steps * 1e3,
            "phase_ms": {"camera_rays": walk_ms[0], "shadow_sort": walk_ms[1], "shadow_rays": walk_ms[2], "resolve": walk_ms[3]},
            "box_tests_per_ray": st_walk["box_tests"] / max(1, st_walk["rays"]),
            "tri_tests_per_ray": st_walk["tri_tests"] / max(1, st_walk["rays"]),
            "valu_frac": (FLOP_BOX * st_walk["box_tests"] + FLOP_TRI * st_walk["tri_tests"]) * args.steps / dt_walk / FP32_PEAK,
        }

    # Self-check of the sharded path (not timed): the frame rank 0 assembled from the ranks' tiles
    # must equal the same frame rendered whole on rank 0's GPU.
    frame_ok = None
    if world > 1 and rank == 0:
        whole = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        p1 = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0)
        scene.render_device(p1, whole.data_ptr(), None, sh)
        torch.cuda.synchronize()
        frame_ok = bool(torch.equal(whole, full))
    phase_ms = [float(np.mean([elapsed_ms(ev[k], ev[k + 1]) for ev in phase_events])) for k in range(4)]
    # Config 5 on every rank (multi-rank runs; a single GPU runs it with the other extras below).
    config5 = None
    sizes = scene.sizes()
    if world > 1 and not args.no_extras and (comm is not None or backend != "nccl"):   # (a rehearsal renders without the gather)
        scene.close()
        config5 = config5_leg(ctx, args, rank, world, comm, dist, red_dev)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total * args.steps / elapsed / 1e6
        # Roofline of the frame's longest kernel, per launch. Algorithmic bytes = what the launch has
        # to move to and from HBM; the tree and the triangle records count once (they fit in L2 /
        # Infinity Cache, so a launch needs them from HBM at most once). "requested_bytes_per_launch"
        # is what the kernel's loads ask the caches for (32 B per child box a wave tests + 48 B per
        # triangle record a wave fetches, from the kernel's own counters in the instrumented build):
        # per-ray structure bytes at the granularity the loads are issued. Its ratio to `traffic` is
        # the cache amplification.
        n_samples_rank = st["primary_rays"]
        n_shadow_rank = st["rays"] - st["primary_rays"]
        projected = bool(st["camera_by_projection"])
        by_cells = bool(st["shadow_by_cells"])
        # the camera rays' main kernel alone: from event [7] (set-up and binning done) to event [1]
        camera_kernel_ms = float(np.mean([elapsed_ms(ev[7], ev[1]) for ev in phase_events]))

        def entry(kernel, kernel_ms, alg_parts, requested_bytes, flops, unit, units):
            alg_bytes = sum(alg_parts.values())
            achieved = alg_bytes / (kernel_ms * 1e-3)
            return {
                "bound": "hbm", "kernel": kernel, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                # HBM traffic needs hardware counters (rocprofv3 --pmc passes of this very command, one
                # pass per counter block: tools/measure.sh): not measurable inside this process. The
                # figure is the one kept under profiles/ for this build (bytes per launch, FETCH_SIZE +
                # WRITE_SIZE as reported; the file also holds the upper bound with the fetch doubled).
                **pmc_traffic(kernel),
                "kernel_ms": kernel_ms, "launch_unit": unit, "units_per_launch": units,
                "alg_bytes_per_launch": alg_bytes, "alg_bytes_per_unit": alg_bytes / max(1, units),
                "alg_bytes_parts": alg_parts,
                "requested_bytes_per_launch": requested_bytes,
                "requested_tb_per_s": requested_bytes / (kernel_ms * 1e-3) / 1e12,
                "valu_tflops": flops / (kernel_ms * 1e-3) / 1e12,
                "valu_frac": flops / (kernel_ms * 1e-3) / FP32_PEAK,
            }

        if projected:
            # k_raster_tile: per entry of a tile's triangle list its id and prepared record in (4 + 96 B;
            # the instrumented build counts them), per sample its record out (hit point + id 16 B, state
            # 4 B, sort key 4 B), the normals/colours once. Nearest hits live in LDS: no bytes.
            parts = {"tile_list_entries": st["primary_fetch_bytes"], "sample_records": 24 * n_samples_rank,
                     "tri_data": sizes["tri_data"]}
            camera = entry("k_raster_tile", camera_kernel_ms, parts, sum(parts.values()),
                           FLOP_TRI * st["primary_tri_tests"] + FLOP_BOX * st["primary_box_tests"], "camera sample",
                           n_samples_rank)
        else:
            # k_wf_primary: 24 B written per sample (hit point + id 16 B, state 4 B, sort key 4 B; camera
            # rays are generated on the device, 0 B in), the wide tree, the triangle records and the
            # normals/colours once.
            parts = {"sample_records": 24 * n_samples_rank, "wide_tree": sizes["wide"], "triangle_records": sizes["recs"],
                     "tri_data": sizes["tri_data"]}
            camera = entry("k_wf_primary", camera_kernel_ms, parts, 24 * n_samples_rank + st["primary_fetch_bytes"],
                           FLOP_BOX * st["primary_box_tests"] + FLOP_TRI * st["primary_tri_tests"], "camera sample",
                           n_samples_rank)
        my_box, my_tri = st["box_tests"] - st["primary_box_tests"], st["tri_tests"] - st["primary_tri_tests"]
        if by_cells:
            # k_shadow_cells: per shadow ray its sorted key (4 B), slot number (4 B), hit point (16 B) and
            # state word read and written back (8 B); every list entry once (box 16 B, id 4 B, depth 4 B),
            # a count per cell (4 B), the triangles' data once (36 B each).
            parts = {"per_ray": 32 * n_shadow_rank, "list_entries": 24 * st["shadow_list_entries"],
                     "cell_counts": 4 * st["shadow_list_cells"], "triangles": 36 * args.tris}
            # what the loads ask the caches for: every wave reads its cells' lists (24 B per box test of a
            # wave; the instrumented build counts box tests per unoccluded ray, so this is a lower bound)
            # and 36 B of triangle data per pair tested; a light-space box test is 3 subtractions or
            # comparisons and 2 more comparisons
            shadow = entry("k_shadow_cells", phase_ms[2], parts, parts["per_ray"] + 36 * my_tri + 24 * my_box // 64,
                           5 * my_box + FLOP_TRI * my_tri, "shadow ray", n_shadow_rank)
        else:
            # k_wf_shadow: the sorted key of every sample slot in (4 B; slots without a shadow ray end
            # there), per shadow ray its slot number (4 B), hit point (16 B) and state word read and
            # written back (8 B), the wide tree and the triangle records once.
            parts = {"per_ray": 4 * n_samples_rank + 28 * n_shadow_rank, "wide_tree": sizes["wide"],
                     "triangle_records": sizes["recs"]}
            shadow = entry("k_wf_shadow", phase_ms[2], parts,
                           parts["per_ray"] + st["fetch_bytes"] - st["primary_fetch_bytes"],
                           FLOP_BOX * my_box + FLOP_TRI * my_tri, "shadow ray", n_shadow_rank)
        # the frame's longest kernel first
        roofline, roofline_other = (camera, shadow) if camera["kernel_ms"] >= shadow["kernel_ms"] else (shadow, camera)
        out = {
            "metric": "Mrays/s (1M-tri soup, 3840x2160 primary + shadow rays; camera rays' nearest hits by "
                      "projection of the triangles onto the sample grid, shadow rays against per-cell triangle "
                      "lists of a light-space grid; value_tree_walk = every ray by tree walk)",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"soup-{args.tris} seed {args.seed:#x}, {W}x{H}, 4 samples/pixel, primary + sun-shadow rays, "
                            f"shade + resolve to RGBA8 on device",
                "rays_per_frame": rays_total,
                "box_tests_per_ray": box_total / max(1, rays_total),
                "tri_tests_per_ray": tri_total / max(1, rays_total),
                "phase_ms": {"camera_rays": phase_ms[0], "shadow_sort": phase_ms[1], "shadow_rays": phase_ms[2],
                             "resolve": phase_ms[3]},
                "camera_rays_by": "projection of the triangles onto the sample grid" if projected else "beam walk",
                "shadow_rays_by": "light-space cells" if by_cells else "beam walk",
                "shadow_lists": {"entries": st["shadow_list_entries"], "longest": st["shadow_list_longest"],
                                 "cells_with_a_list": st["shadow_list_cells"]},
                "camera_phase_counters": {"triangle_tests": st["primary_tri_tests"], "object_box_tests": st["primary_box_tests"],
                                          "node_visits_or_sample_blocks": st["primary_node_visits"]},
                "scene_build_s": build_s,
                "parallelism": f"tiles{world}" if world > 1 else "single",
                "sharded_frame_equals_whole_frame": frame_ok,
                "backend": backend if world > 1 else None,
                "gather": ("rt_gather_tiles_device (RCCL send/recv to rank 0)" if comm is not None else
                           (("torch.distributed.gather over RCCL (the library's communicator did not come up)"
                             if backend == "nccl" else "host-staged rehearsal") if world > 1 else None)),
                "ranks_seen": ranks_seen,
            },
            "roofline": roofline,
            "roofline_second_kernel": roofline_other,
        }
        if elapsed_in_flight is not None:
            out["two_frames_in_flight"] = {
                "value": rays_total * args.steps / elapsed_in_flight / 1e6,
                "ms_per_step": elapsed_in_flight / args.steps * 1e3,
                "note": "the same K steps with the library's default of two frames in flight (RAYS_HIP_FRAMES_IN_FLIGHT=2)",
            }
        if tree_walk is not None:
            out["value_tree_walk"] = tree_walk["value"]
            out["ms_per_step_tree_walk"] = tree_walk["ms_per_step"]
            out["tree_walk"] = tree_walk
        if world == 1 and not args.no_extras:
            # Free the headline scene's frame workspace before the 10M-triangle scene comes in.
            scene.close()
            out["teapot"] = teapot_leg(ctx, args, not args.no_cpu_baseline)
            t0 = time.perf_counter()
            objects10 = scenes.soup(args.bounce_tris, 0x5EED0002)
            scene10 = Scene(ctx, objects10)
            build10_s = time.perf_counter() - t0
            out["roofline_bounce"] = bounce_roofline(ctx, args, new_event, elapsed_ms, scene10)
            out["roofline_bounce"]["scene_generate_and_build_s"] = build10_s
            out["roofline_bounce"]["sorted_vs_unsorted"] = scatter_order_probe(scene10, objects10)
            out["config5"] = config5_leg(ctx, args, 0, 1, None, dist, red_dev, scene10)
            scene10.close()
        if config5 is not None:
            out["config5"] = config5
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(objects, W, H)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
