#!/usr/bin/env python3
"""Phase times and per-wave counters of the benchmark frame under the current environment knobs.
usage: [PROBE_SCENE=teapot|<triangles>] [PROBE_W=.. PROBE_H=..] python tools/frame_probe.py [label]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import Context, Scene, render_params, scenes

W, H = int(os.environ.get("PROBE_W", "3840")), int(os.environ.get("PROBE_H", "2160"))
which = os.environ.get("PROBE_SCENE", "1000000")
objects = scenes.default_scene() if which == "teapot" else scenes.soup(int(which), 0x5EED0001)
import time
ctx = Context(0)
t0 = time.time()
scene = Scene(ctx, objects)
t_build = time.time() - t0
p = render_params(W, H, 2, reflect_depth=1, diffuse_depth=1, sun=scenes.SUN_T0)
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
st = scene.render_device_stats(p, frame.data_ptr())
hip = ctypes.CDLL("libamdhip64.so")
def ev():
    e = ctypes.c_void_p(); assert hip.hipEventCreate(ctypes.byref(e)) == 0; return e
def ms(a, b):
    m = ctypes.c_float(); assert hip.hipEventElapsedTime(ctypes.byref(m), a, b) == 0; return m.value
sh = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    scene.render_device(p, frame.data_ptr(), None, sh)
evs = [[ev() for _ in range(8)] for _ in range(8)]   # RT_RENDER_EVENTS handles a frame
for e in evs:
    scene.render_device_events(p, frame.data_ptr(), [x.value for x in e], None, sh)
torch.cuda.synchronize()
ph = [np.mean([ms(e[k], e[k + 1]) for e in evs]) for k in range(4)]
waves = st["primary_rays"] / 64
sw = (st["rays"] - st["primary_rays"]) / 64
label = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{label:28s} build {t_build:6.2f} s  total {sum(ph):6.2f} ms  cam {ph[0]:5.2f} sort {ph[1]:5.2f} shd {ph[2]:5.2f} res {ph[3]:5.2f} | "
      f"cam/wave: nodes {st['primary_node_visits']/waves:6.1f} box {st['primary_box_tests']/waves:7.1f} "
      f"tri/ray {st['primary_tri_tests']/st['primary_rays']:6.2f} KB {st['primary_fetch_bytes']/waves/1024:6.1f} | "
      f"shd/wave: nodes {(st['node_visits']-st['primary_node_visits'])/sw:6.1f} box {(st['box_tests']-st['primary_box_tests'])/sw:7.1f} "
      f"tri/ray {(st['tri_tests']-st['primary_tri_tests'])/(st['rays']-st['primary_rays']):6.2f} "
      f"KB {(st['fetch_bytes']-st['primary_fetch_bytes'])/sw/1024:6.1f}", flush=True)
scene.close(); ctx.close()
