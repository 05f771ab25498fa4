// index_radix.hip — the library sort behind the per-log indexes when a row is 8192 events or longer (index_kernels.hip:
// shorter rows are ordered by the counting sort of length_sort.hip): rocPRIM's LSD radix sort, descending, 16 key bits, stable.
// Alone in its translation unit on purpose: HIP loads a translation unit's code objects at the first launch of any of its
// kernels, and rocPRIM's take ~7 ms — more than half a fold of the 10 M-aggregate log (round 5's one-shot index cost).
#include <rocprim/device/device_radix_sort.hpp>

#include "replay_internal.h"

namespace surge {

constexpr unsigned kLenKeyBits = 16;

// scratch bytes rocPRIM needs to sort n (length, id) pairs
hipError_t index_temp_bytes(int64_t n, size_t* bytes) {
  size_t a = 0;
  hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int64_t*)nullptr,
                                                (int64_t*)nullptr, (size_t)(n > 0 ? n : 1), 0u, kLenKeyBits, (hipStream_t) nullptr);
  if (e != hipSuccess) return e;
  *bytes = a;
  return hipSuccess;
}

hipError_t launch_radix_sort_pairs_desc(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const int64_t* vals_in,
                                        int64_t* vals_out, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  return rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, kLenKeyBits, stream);
}

}  // namespace surge
