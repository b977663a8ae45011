// ordered_sort.hip — the ONE library primitive in the engine: rocPRIM's device radix sort, compiled into its own object because of what it costs to
// compile. It runs when the value-ordered view of an index is (re)built (include/bmx.h bmx_index_set_ordered: the view is a second copy of the index
// columns sorted by (value, position)) — never on the merge path and never per query. gfx950 only, like everything else here.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <climits>
#include <cstddef>
#include <cstdint>

namespace bmx {

// Keys are sorted REBASED: 0 for a tombstone (the column type's minimum: it sorts in front of every value), v - lo + 1 otherwise, lo = the smallest value of
// the column — an unsigned key of which only the low `bits` bits can be set, so a column of few distinct values (an age, a level, a status: what an index is
// usually built on) takes ONE radix pass instead of three (int32) or six (int64). The caller turns the keys back into values (k_gather_ids). values = positions
// 0..n-1; stable, so equal values keep ascending positions. tmp == nullptr: *tmp_bytes receives the scratch size. Only enqueues.
struct Rebase32 { int32_t lo; __host__ __device__ uint32_t operator()(int32_t v) const { return v == INT32_MIN ? 0u : (uint32_t)((int64_t)v - (int64_t)lo + 1); } };
struct Rebase64 { int64_t lo; __host__ __device__ uint64_t operator()(int64_t v) const { return v == INT64_MIN ? 0ull : (uint64_t)v - (uint64_t)lo + 1ull; } };

hipError_t sort_pairs_i32(void* tmp, size_t* tmp_bytes, const int32_t* kin, int32_t lo, unsigned bits, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, rocprim::make_transform_iterator(kin, Rebase32{lo}), kout, vin, vout, n, 0, bits, s);
}
hipError_t sort_pairs_i64(void* tmp, size_t* tmp_bytes, const int64_t* kin, int64_t lo, unsigned bits, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, rocprim::make_transform_iterator(kin, Rebase64{lo}), kout, vin, vout, n, 0, bits, s);
}

}  // namespace bmx
