"""rt_trace_closest_device_noalloc: the allocation-free form of rt_trace_closest_device must give the
same hits, bit for bit, on two calls in a row, the second with a smaller input (its work counter
lives in one per-GPU array that keeps the last call's contents)."""
import numpy as np
import pytest

from raytracing_amd import HIT_DTYPE, Scene, scenes

pytestmark = pytest.mark.gpu


def _both(g, rays):
    import torch

    d_rays = torch.from_numpy(rays).cuda()
    out_a = torch.full((len(rays) * 4,), -1, dtype=torch.int32, device="cuda")
    out_b = torch.full((len(rays) * 4,), -1, dtype=torch.int32, device="cuda")
    sh = torch.cuda.current_stream().cuda_stream
    g.trace_closest_device(d_rays.data_ptr(), len(rays), out_a.data_ptr(), sh)
    g.trace_closest_device_noalloc(d_rays.data_ptr(), len(rays), out_b.data_ptr(), sh)
    torch.cuda.synchronize()
    return out_a.cpu().numpy().view(HIT_DTYPE), out_b.cpu().numpy().view(HIT_DTYPE)


def test_no_alloc_entry_point_equals_trace_closest_device(ctx, oracle, soup64k):
    g = Scene(ctx, soup64k)
    try:
        big = scenes.camera_rays(320, 180, 2)          # 230,400 rays
        small = scenes.camera_rays(97, 41, 1)          # 3,977 rays: fewer than one chunk per wave slot
        for rays in (big, small):
            ref, got = _both(g, rays)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "hits differ from rt_trace_closest_device"
            assert (got["tri"] != 0xFFFFFFFF).any()
        # and the smaller batch against the CPU checker
        want = oracle.OracleScene(soup64k).trace_closest(small)
        assert np.array_equal(got["tri"], want["tri"])
        hit = want["tri"] != 0xFFFFFFFF
        assert np.array_equal(got["t"][hit].view(np.uint32), want["t"][hit].view(np.uint32))
        # sizes beyond the stated maximum are refused with a message, nothing is launched
        with pytest.raises(Exception, match="at most"):
            g.trace_closest_device_noalloc(1, (1 << 31) + 1, 1, 0)
    finally:
        g.close()
