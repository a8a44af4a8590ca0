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


def test_no_alloc_entry_point_allocates_nothing_and_shares_one_array_per_gpu(ctx, soup64k):
    """The device's free memory is the same before and after the call (hipMemGetInfo through torch, which
    allocates nothing itself once the buffers exist), and a second context on the same GPU uses the same
    per-GPU array: calls through the two, one after the other on one stream, give the same hits."""
    import torch

    from raytracing_amd import Context

    ctx2 = Context(0)
    g, g2 = Scene(ctx, soup64k), Scene(ctx2, soup64k)
    try:
        rays = scenes.camera_rays(160, 90, 2)
        d_rays = torch.from_numpy(rays).cuda()
        outs = [torch.full((len(rays) * 4,), -1, dtype=torch.int32, device="cuda") for _ in range(3)]
        sh = torch.cuda.current_stream().cuda_stream
        g.trace_closest_device(d_rays.data_ptr(), len(rays), outs[0].data_ptr(), sh)   # takes its counter pool on first use
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        g.trace_closest_device_noalloc(d_rays.data_ptr(), len(rays), outs[1].data_ptr(), sh)
        g2.trace_closest_device_noalloc(d_rays.data_ptr(), len(rays), outs[2].data_ptr(), sh)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the allocation-free entry point changed the device's free memory"
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        g.close()
        g2.close()
        ctx2.close()
