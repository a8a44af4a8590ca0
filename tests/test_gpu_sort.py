"""The library's own sort of the shadow rays' keys (raytracing_amd/csrc/rt_radix.hip) against the
library radix sort on the same keys, through a small shim built with the library
(tests/native/radix_check.hip -> raytracing_amd/lib/libradix_check.so)."""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    path = os.path.join(ROOT, "raytracing_amd", "lib", "libradix_check.so")
    assert os.path.exists(path), "libradix_check.so is missing: run `make -C raytracing_amd/csrc`"
    lib = ctypes.CDLL(path)
    lib.radix_check.restype = ctypes.c_longlong
    lib.radix_check.argtypes = [ctypes.c_uint32] * 5
    return lib


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100_000, 524_288 + 17, 5_000_000])
@pytest.mark.parametrize("key_bits,invalid", [(16, 138), (16, 0), (16, 256), (8, 100), (18, 138), (24, 30), (32, 200)])
def test_own_sort_equals_the_library_sort(shim, n, key_bits, invalid):
    """Same order as a stable library sort of the valid keys, the count of valid keys right; sizes
    around the block and group limits, every key width in use, from no invalid keys to all."""
    for clustered in (0, 1):
        rc = shim.radix_check(n, key_bits, 7 * n + key_bits + clustered, invalid, clustered)
        assert rc == 0, f"n={n} key_bits={key_bits} invalid={invalid}/256 clustered={clustered}: differs at {rc - 1}"


@pytest.mark.parametrize("key_bits,invalid", [(16, 138), (24, 30)])
def test_own_sort_with_more_groups_than_the_offsets_kernel_folds(shim, key_bits, invalid):
    """70 M keys are 134 groups of 256 blocks: beyond 128 groups the groups' bases come from a kernel of
    their own instead of being worked out by every workgroup of the offsets kernel."""
    n = 70_000_000
    rc = shim.radix_check(n, key_bits, 7 * n + key_bits, invalid, 0)
    assert rc == 0, f"n={n} key_bits={key_bits} invalid={invalid}/256: differs at {rc - 1}"
