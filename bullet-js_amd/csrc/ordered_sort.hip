// ordered_sort.hip — the ONE library primitive in the engine: rocPRIM's device radix sort, compiled into its own object because of what it costs to
// compile. It runs when the value-ordered view of an index is (re)built (include/bmx.h bmx_index_set_ordered: the view is a second copy of the index
// columns sorted by (value, position)) — never on the merge path and never per query. gfx950 only, like everything else here.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstddef>
#include <cstdint>

namespace bmx {

// keys = the index's value column (signed; a tombstone is the type's minimum and sorts first), values = positions 0..n-1. Stable, so equal values keep
// ascending positions. tmp == nullptr: *tmp_bytes receives the scratch size. Only enqueues.
hipError_t sort_pairs_i32(void* tmp, size_t* tmp_bytes, const int32_t* kin, int32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0, 32, s);
}
hipError_t sort_pairs_i64(void* tmp, size_t* tmp_bytes, const int64_t* kin, int64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0, 64, s);
}

}  // namespace bmx
