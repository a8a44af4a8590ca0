import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) device")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi

    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def ctx():
    # PyTorch wheels bundle a HIP runtime of their own (soname libamdhip64.so.7, like the system's).
    # Loaded first, it is the copy the library binds to as well, and streams, events and pointers can
    # pass between the two; loaded second, the process holds two runtimes that do not know each
    # other's handles. The GPU tests take their device buffers and streams from torch.
    import torch  # noqa: F401

    from raytracing_amd import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def default_objects():
    from raytracing_amd import scenes

    return scenes.default_scene()


@pytest.fixture(scope="session")
def soup64k():
    from raytracing_amd import scenes

    return scenes.soup(65536, 0x5EED0001)
