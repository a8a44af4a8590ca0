#!/usr/bin/env python3
"""bench.py's diffuse-bounce leg by itself in a fresh process (the same function, the same frame)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracing_amd import Context, Scene, scenes
args = argparse.Namespace(width=3840, height=2160, bounce_tris=10_000_000)
hip = ctypes.CDLL("libamdhip64.so")
def new_event():
    e = ctypes.c_void_p(); assert hip.hipEventCreate(ctypes.byref(e)) == 0; return e
def elapsed_ms(a, b):
    m = ctypes.c_float(); assert hip.hipEventElapsedTime(ctypes.byref(m), a, b) == 0; return m.value
ctx = Context(0)
sc = Scene(ctx, scenes.soup(args.bounce_tris, 0x5EED0002))
r = bench.bounce_roofline(ctx, args, new_event, elapsed_ms, sc)
print({k: r[k] for k in ("frame_ms", "kernel_ms", "scatter_mrays_per_s")})
