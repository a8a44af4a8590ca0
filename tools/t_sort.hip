// Checks rtk::sort_pairs_u32 with a partial bit range against a host stable sort.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#include "../raytracing_amd/csrc/rt_kernels.h"
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 518400;
    const unsigned bits = argc > 2 ? atoi(argv[2]) : 32;
    std::vector<uint32_t> k(n), v(n);
    uint32_t x = 12345;
    for (size_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        k[i] = (x >> 3) % 3 == 0 ? 0xFFFFFFFFu : x;
        v[i] = (uint32_t)i;
    }
    uint32_t *dk0, *dk1, *dv0, *dv1; void* tmp;
    hipMalloc(&dk0, n * 4); hipMalloc(&dk1, n * 4); hipMalloc(&dv0, n * 4); hipMalloc(&dv1, n * 4);
    hipMemcpy(dk0, k.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv0, v.data(), n * 4, hipMemcpyHostToDevice);
    size_t tb = 0;
    hipError_t e = rtk::sort_pairs_u32(nullptr, &tb, nullptr, nullptr, nullptr, nullptr, n, 32, 0);
    printf("query: %d temp %zu\n", (int)e, tb);
    hipMalloc(&tmp, tb ? tb : 16);
    e = rtk::sort_pairs_u32(tmp, &tb, dk0, dk1, dv0, dv1, n, bits, 0);
    hipDeviceSynchronize();
    printf("sort: %d\n", (int)e);
    std::vector<uint32_t> ok(n), ov(n), ik(n);
    hipMemcpy(ok.data(), dk1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ov.data(), dv1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ik.data(), dk0, n * 4, hipMemcpyDeviceToHost);
    size_t bad_pair = 0, bad_in = 0, bad_order = 0;
    std::vector<char> seen(n, 0);
    size_t dup = 0;
    for (size_t i = 0; i < n; ++i) {
        if (ik[i] != k[i]) ++bad_in;
        if (ov[i] >= n || ok[i] != k[ov[i]]) ++bad_pair;
        else { if (seen[ov[i]]) ++dup; seen[ov[i]] = 1; }
    }
    {
        const uint32_t m = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        size_t b = 0, unstable = 0;
        for (size_t i = 1; i < n; ++i) {
            if ((ok[i - 1] & m) > (ok[i] & m)) ++b;
            if ((ok[i - 1] & m) == (ok[i] & m) && ov[i - 1] > ov[i]) ++unstable;
        }
        printf("bits %u: out of order %zu, unstable %zu\n", bits, b, unstable);
    }
    printf("n %zu input changed %zu bad pairs %zu dup %zu\n", n, bad_in, bad_pair, dup);
    return 0;
}
