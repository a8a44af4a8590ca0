"""Builds the native libraries in-tree: librays_hip.so (gfx950 kernels + C ABI) and
librays_host.so (C++ mirror of the reference's host API). `python -m raytracing_amd.build`."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_all(verbose=True):
    for sub in ("csrc", "host"):
        cmd = ["make", "-C", os.path.join(_HERE, sub)]
        if not verbose:
            cmd.insert(1, "-s")
        subprocess.check_call(cmd)


if __name__ == "__main__":
    build_all()
