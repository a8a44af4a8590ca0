// rt_sort.hip -- key/value radix sort used to put shadow rays into light-space order. A plain
// library sort (rocPRIM); kept in its own translation unit because the header is heavy.
#include <cstdint>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "rt_kernels.h"

namespace rtk {

hipError_t sort_pairs_u32(void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned key_bits, hipStream_t s) {
    // Without a value array the values are the positions 0..n-1, generated on the fly: the producer
    // of the keys then writes, and the first pass reads, four bytes less per pair.
    const rocprim::counting_iterator<uint32_t> iota(0u);
    if (!temp) {  // size query: one workspace must serve both forms
        size_t a = 0, b = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, a, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, s);
        if (e != hipSuccess) return e;
        e = rocprim::radix_sort_pairs(nullptr, b, keys_in, keys_out, iota, vals_out, n, 0u, key_bits, s);
        *temp_bytes = a > b ? a : b;
        return e;
    }
    size_t bytes = *temp_bytes;
    if (vals_in) return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, s);
    return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, iota, vals_out, n, 0u, key_bits, s);
}

}  // namespace rtk
