// rt_sort.hip -- key/value radix sort used to put shadow rays into light-space order. A plain
// library sort (rocPRIM); kept in its own translation unit because the header is heavy.
#include <cstdint>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "rt_kernels.h"

namespace rtk {

hipError_t sort_pairs_u32(void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned key_bits, hipStream_t s) {
    size_t bytes = *temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, s);
    *temp_bytes = bytes;
    return e;
}

}  // namespace rtk
