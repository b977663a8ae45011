// scan_kernels.h — index build and index scans (K5/K6 of SURVEY §2.1), gfx950.
//
// An index (reference: BulletQuery.index/_buildIndex, src/bullet-query.js:30-73) is a pair of dense,
// coalesced columns (node id u64, value i32 or i64) of the rows carrying one field, compacted from the
// table in slot order. range()/equals()/count() (src/bullet-query.js:186-261, 293-313) stream the value
// column once (16 B per lane per load), and compact the matching node ids with select.h. The scan is
// HBM-stream bound: w*R bytes in + 8*M bytes out.
#pragma once
#include "select.h"
#include "merge_kernels.h"
#include "../../include/bmx.h"

namespace bmx {

// A value in the int32 column. A tombstone (VAL_DELETED) becomes INT32_MIN, which no int32 scan matches (bounds are clamped to INT32_MIN + 1), and does
// not count as a wide value; a real -2^31 therefore does count as wide (the index then scans its int64 column).
__device__ __forceinline__ int32_t v32_of(int64_t v) { return v == VAL_DELETED ? INT32_MIN : (int32_t)v; }
__device__ __forceinline__ bool is_wide(int64_t v) { return v != VAL_DELETED && (v != (int64_t)(int32_t)v || v == (int64_t)INT32_MIN); }

// ---- predicates over the resident table (index build, dump) ----
struct PredSlotField {  // slots holding a row of `field`
  static constexpr int E = 2;
  const Slot* slots; uint32_t field;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      uint64_t s = first + e;
      if (s < n) {
        uint4 lo = reinterpret_cast<const uint4*>(slots + s)[0];
        bool occ = !(lo.x == 0xFFFFFFFFu && lo.y == 0xFFFFFFFFu);
        if (occ && lo.z == field) m |= 1u << e;
      }
    }
    return m;
  }
};
struct PredSlotAny {  // every occupied slot that holds data (tombstones are not dumped)
  static constexpr int E = 2;
  const Slot* slots;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      uint64_t s = first + e;
      if (s < n) {
        const uint4* q = reinterpret_cast<const uint4*>(slots + s);
        uint4 lo = q[0], hi = q[1];
        const int64_t v = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
        if (!(lo.x == 0xFFFFFFFFu && lo.y == 0xFFFFFFFFu) && v != VAL_DELETED) m |= 1u << e;
      }
    }
    return m;
  }
};
struct EmitIndex {  // slot -> index columns (+ the slot's position in its index: what the incremental maintenance looks up)
  const Slot* slots; uint64_t* ids; int64_t* v64; int32_t* v32; uint32_t* wide;  // *wide set if a value does not fit i32
  uint32_t* slot_pos;   // [nslots] or nullptr
  __device__ void operator()(uint64_t pos, uint64_t s) const {
    const Slot& sl = slots[s];
    int64_t v = sl.val;
    ids[pos] = sl.id; v64[pos] = v; v32[pos] = v32_of(v);
    if (is_wide(v)) *wide = 1u;
    if (slot_pos) slot_pos[s] = (uint32_t)pos;
  }
};

// ---- incremental index maintenance (the device-side _updateIndices, src/bullet-query.js:82-110) ----
// While an index exists, the compaction of every merge appends one entry per winner to a change log: the winner's slot (bit 31: the row was
// created by this batch) and its field. Before the next scan the log is applied to the dense columns: created rows of the indexed field are
// appended in log order (ordered select over the log: deterministic), the others overwrite their value at the position the build recorded
// for their slot. Values are read from the TABLE at that moment, not from the log, so a row that appears twice gets the same (current) value
// from both entries. ~0.1-0.15 ms per logged 1M-delta batch instead of a rebuild that reads the whole table twice (0.5 ms at 10M rows, 3.5 ms at 100M).
constexpr uint32_t POS_NONE = 0xFFFFFFFFu;
constexpr uint32_t CHG_CREATED = 0x80000000u;
struct ChgLog {   // k_compact_winners' view; chg == nullptr: no logging
  uint2* chg; const unsigned long long* base; unsigned long long* next;   // entries so far (read), entries after this batch (written by the last block)
  const uint32_t* slot_of; const uint32_t* field; const bmx_delta_rec* recs; uint64_t cap;
};
struct PredLogCreated {   // log entries that create a row of `field` which the index does not hold yet
  static constexpr int E = 2;
  const uint2* chg; const unsigned long long* n_dev; uint32_t field; const uint32_t* slot_pos;
  __device__ uint32_t mask(uint64_t first, uint64_t) const {
    const uint64_t n = *n_dev;
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const uint64_t i = first + e;
      if (i < n) {
        const uint2 x = chg[i];
        if ((x.x & CHG_CREATED) && x.y == field && slot_pos[x.x & ~CHG_CREATED] == POS_NONE) m |= 1u << e;
      }
    }
    return m;
  }
};
struct EmitAppend {   // created row -> the end of the index columns, in log order
  const uint2* chg; const Slot* slots; uint64_t* ids; int64_t* v64; int32_t* v32; uint32_t* wide; uint32_t* slot_pos; uint64_t base, cap;
  __device__ void operator()(uint64_t rank, uint64_t i) const {
    const uint64_t pos = base + rank;
    if (pos >= cap) return;                       // the host sees the total and rebuilds
    const uint32_t s = chg[i].x & ~CHG_CREATED;
    const Slot& sl = slots[s];
    const int64_t v = sl.val;
    ids[pos] = sl.id; v64[pos] = v; v32[pos] = v32_of(v);
    if (is_wide(v)) *wide = 1u;
    slot_pos[s] = (uint32_t)pos;
  }
};
// every log entry of `field` that changed a row the index holds: refresh its value from the table (both value columns: the declarative
// filter reads the int64 one whatever the range scans use). Entries that created their row were appended with the current value already.
// track = 1: compare before writing, wide[1] says whether any value really changed (the index has a value-ordered view that is not current anyway).
// track = 2: CAPTURE (the view is current and will be patched, view_kernels.h): the new value is EXCHANGED into the int64 column, so of several log
//            entries of one row exactly one sees the value the view still holds; log entry i leaves (position, old value) of a real change in cl[i] and
//            (POS_NONE, -) otherwise — no counter, no shared atomic (a first version appended through one counter: 15 600 same-address atomics of 17 ns
//            each made this a 228-us kernel); the holes are taken out by an ordered select afterwards (PredChanged / EmitChanged).
__global__ __launch_bounds__(256) void k_ix_update(const uint2* __restrict__ chg, const unsigned long long* __restrict__ n_dev, const Slot* __restrict__ slots,
                                                   uint32_t field, const uint32_t* __restrict__ slot_pos, int64_t* __restrict__ v64, int32_t* __restrict__ v32,
                                                   uint32_t* wide /* wide[1]: set when a value in the index really changed */, uint32_t track,
                                                   uint32_t* __restrict__ cl_pos, int64_t* __restrict__ cl_old, uint64_t cl_cap) {
  const uint64_t n = *n_dev;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
    const uint2 x = chg[i];
    uint32_t hole = POS_NONE; int64_t old = 0;
    if (x.y == field && !(x.x & CHG_CREATED)) {
      const uint32_t s = x.x;
      const uint32_t p = slot_pos[s];
      if (p != POS_NONE) {
        const uint4 hi = reinterpret_cast<const uint4*>(slots + s)[1];
        const int64_t v = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
        if (track == 2u) {
          old = (int64_t)atomicExch(reinterpret_cast<unsigned long long*>(v64 + p), (unsigned long long)v);
          if (old != v) { hole = p; wide[1] = 1u; }
        } else {
          if (track && v64[p] != v) wide[1] = 1u;     // (the value-ordered view of the index is stale only then)
          v64[p] = v;
        }
        v32[p] = v32_of(v);
        if (is_wide(v)) *wide = 1u;
      }
    }
    if (track == 2u && i < cl_cap) { cl_pos[i] = hole; cl_old[i] = old; }
  }
}
struct PredChanged {   // entries of the captured change run that are not holes
  static constexpr int E = 4;
  const uint32_t* cl_pos; const unsigned long long* n_dev;
  __device__ uint32_t mask(uint64_t first, uint64_t) const {
    const uint64_t n = *n_dev;
    uint32_t m = 0;
    if (first + 4 <= n) { const uint4 x = *reinterpret_cast<const uint4*>(cl_pos + first); m = (uint32_t)(x.x != POS_NONE) | ((uint32_t)(x.y != POS_NONE) << 1) | ((uint32_t)(x.z != POS_NONE) << 2) | ((uint32_t)(x.w != POS_NONE) << 3); }
    else for (int e = 0; e < 4 && first + e < n; e++) m |= (uint32_t)(cl_pos[first + e] != POS_NONE) << e;
    return m;
  }
};
struct EmitChanged {
  const uint32_t* cl_pos; const int64_t* cl_old; uint32_t* out_pos; int64_t* out_old;
  __device__ void operator()(uint64_t rank, uint64_t i) const { out_pos[rank] = cl_pos[i]; out_old[rank] = cl_old[i]; }
};
struct EmitRows {  // slot -> dumped row columns (bounded by cap)
  const Slot* slots; uint64_t cap; uint64_t* id; uint32_t* field; int64_t* ts; int64_t* val;
  __device__ void operator()(uint64_t pos, uint64_t s) const {
    if (pos >= cap) return;
    const Slot& sl = slots[s];
    id[pos] = sl.id; field[pos] = sl.field; ts[pos] = ts_value(sl.ts); val[pos] = sl.val;
  }
};

// ---- predicates over an index value column ----
// nt: the column is larger than the Infinity Cache and read once per scan: nontemporal loads (100M int32 rows: 70-72 -> 65 us, int64: 140 -> 124 us; a column that
// fits the cache is re-read from it by the next scan and is 0.8 us SLOWER with nt at 10M rows: profiles/r03_scan_nt_ab.log)
struct PredRange32 {  // lo <= v <= hi on an int32 column, 4 values (16 B) per lane
  static constexpr int E = 4;
  const int32_t* v; int32_t lo, hi; bool nt = false;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
    if (first + 4 <= n) {
      typedef int i32x4_t __attribute__((ext_vector_type(4)));
      int4 x;
      if (nt) { i32x4_t y = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t*>(v + first)); x = make_int4(y.x, y.y, y.z, y.w); }
      else x = *reinterpret_cast<const int4*>(v + first);
      m = (uint32_t)(x.x >= lo && x.x <= hi) | ((uint32_t)(x.y >= lo && x.y <= hi) << 1) |
          ((uint32_t)(x.z >= lo && x.z <= hi) << 2) | ((uint32_t)(x.w >= lo && x.w <= hi) << 3);
    } else {
      for (int e = 0; e < 4 && first + e < n; e++) { int32_t x = v[first + e]; m |= (uint32_t)(x >= lo && x <= hi) << e; }
    }
    return m;
  }
};
struct PredRange64 {  // int64 column, 2 values (16 B) per lane
  static constexpr int E = 2;
  const int64_t* v; int64_t lo, hi; bool nt = false;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
    if (first + 2 <= n) {
      typedef long long i64x2_t __attribute__((ext_vector_type(2)));
      longlong2 x;
      if (nt) { i64x2_t y = __builtin_nontemporal_load(reinterpret_cast<const i64x2_t*>(v + first)); x = make_longlong2(y.x, y.y); }
      else x = *reinterpret_cast<const longlong2*>(v + first);
      m = (uint32_t)(x.x >= lo && x.x <= hi) | ((uint32_t)(x.y >= lo && x.y <= hi) << 1);
    } else if (first < n) {
      int64_t x = v[first]; m = (uint32_t)(x >= lo && x <= hi);
    }
    return m;
  }
};

// Declarative filter (subset of filter(path, fn), src/bullet-query.js:270-283): term 0 on the index column,
// the remaining terms by probing the node's other field rows in the table.
constexpr int MAX_TERMS = 8;
struct PredFilter {
  static constexpr int E = 2;
  const int64_t* v; const uint64_t* ids; const Slot* slots; uint64_t nslots;
  uint32_t nterms; bmx_term t[MAX_TERMS];
  __device__ bool rest(uint64_t id) const {
    for (uint32_t k = 1; k < nterms; k++) {
      ProbeSeq<4> ps(id, t[k].field, nslots);
      bool ok = false;
      for (uint64_t p = 0; p < nslots; ++p) {
        const uint4* q = reinterpret_cast<const uint4*>(slots + ps.slot());
        uint4 lo = q[0];
        uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
        if (sid == EMPTY_ID) break;
        if (sid == id && lo.z == t[k].field) {
          uint4 hi = q[1];
          int64_t x = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
          ok = x >= t[k].lo && x <= t[k].hi;
          break;
        }
        ps.next();
      }
      if (!ok) return false;
    }
    return true;
  }
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      uint64_t i = first + e;
      if (i < n) { int64_t x = v[i]; if (x >= t[0].lo && x <= t[0].hi && rest(ids[i])) m |= 1u << e; }
    }
    return m;
  }
};

// Declarative filter through the value-ordered view: term 0 selects ONE run of the sorted columns (k_ordered_bounds), the other terms are probed for its ids only —
// O(log R + candidates) instead of one pass over the column. Survivors are appended as they are found (wave ballot + one atomic per wave): no order.
template <class PF>
__global__ __launch_bounds__(256) void k_ordered_filter(const uint64_t* __restrict__ s_ids, const unsigned long long* __restrict__ ab, PF P, uint64_t* __restrict__ out, uint64_t cap,
                                                        unsigned long long* __restrict__ n_out) {
  const uint64_t a = ab[0], m = ab[1] - a;
  const uint64_t rounds = (m + (uint64_t)gridDim.x * 256u - 1) / ((uint64_t)gridDim.x * 256u);
  for (uint64_t r = 0; r < rounds; r++) {          // (uniform trip count per wave: the ballot below wants every lane there)
    const uint64_t i = (r * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
    uint64_t id = 0; bool ok = false;
    if (i < m) { id = s_ids[a + i]; ok = P.rest(id); }
    const unsigned long long bal = __ballot(ok);
    if (bal) {
      unsigned long long base = 0;
      if ((threadIdx.x & 63u) == 0) base = atomicAdd(n_out, (unsigned long long)__popcll(bal));
      base = __shfl(base, 0);
      if (ok) { const uint64_t pos = base + (uint64_t)__popcll(bal & ((1ull << (threadIdx.x & 63u)) - 1ull)); if (out && pos < cap) out[pos] = id; }
    }
  }
}

struct EmitIds {  // index position -> node id (bounded by cap)
  static constexpr bool STREAMABLE = true;     // dense blocks read their part of the id column as a stream (select.h scan_emit_stream_block)
  const uint64_t* ids; uint64_t* out; uint64_t cap; bool nt = false;   // nt: the id column is larger than the Infinity Cache and read once per scan
  uint32_t stream_min = SCAN_STREAM_MIN;       // matches per 8192-row block from which the block streams (a launch argument so that one process can A/B it)
  uint32_t ntx = 0;                            // round 5 A/B (BMX_SCAN_NT): bit 0 = the gathered ids are loaded nontemporally too, bit 1 = the output is stored nontemporally
  __device__ __forceinline__ uint32_t stream_from() const { return stream_min; }
  __device__ __forceinline__ bool deep() const { return (ntx & 4u) != 0; }      // (uniform) streamed blocks keep 16 loads per lane in flight instead of 8
  __device__ __forceinline__ void st1(uint64_t pos, uint64_t id) const { if (ntx & 2u) __builtin_nontemporal_store((unsigned long long)id, reinterpret_cast<unsigned long long*>(out + pos)); else out[pos] = id; }
  __device__ void operator()(uint64_t pos, uint64_t i) const {
    if (out && pos < cap) st1(pos, (nt && (ntx & 1u)) ? (uint64_t)__builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(ids + i)) : ids[i]);
  }
  __device__ __forceinline__ void load2(uint64_t row, uint64_t& a, uint64_t& b) const {   // ids[row], ids[row + 1]; row is even: 16-byte aligned
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    u64x2_t v;
    if (nt) v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(ids + row)); else v = *reinterpret_cast<const u64x2_t*>(ids + row);
    a = v.x; b = v.y;
  }
  __device__ __forceinline__ uint64_t load1(uint64_t row) const { return ids[row]; }
  __device__ __forceinline__ void put(uint64_t pos, uint64_t id) const { if (out && pos < cap) st1(pos, id); }
  __device__ __forceinline__ uint32_t odd(uint64_t pos) const { return (uint32_t)((reinterpret_cast<uintptr_t>(out + pos) >> 3) & 1u); }
  __device__ __forceinline__ void put2(uint64_t pos, uint64_t a, uint64_t b) const {       // out[pos], out[pos + 1]; &out[pos] is 16-byte aligned (odd(pos) == 0): one store
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    if (!out) return;
    if (pos + 1 < cap) { u64x2_t v; v.x = a; v.y = b; if (ntx & 2u) __builtin_nontemporal_store(v, reinterpret_cast<u64x2_t*>(out + pos)); else *reinterpret_cast<u64x2_t*>(out + pos) = v; }
    else if (pos < cap) st1(pos, a);
  }
};
struct EmitPos {  // index position itself (u32, bounded by cap): no read of the id column — the caller maps positions to whatever it mirrors per index row
  static constexpr bool STREAMABLE = false;
  uint32_t* out; uint64_t cap;
  __device__ void operator()(uint64_t pos, uint64_t i) const { if (out && pos < cap) out[pos] = (uint32_t)i; }
  __device__ __forceinline__ uint32_t stream_from() const { return 0xFFFFFFFFu; }
  __device__ __forceinline__ bool deep() const { return false; }
  __device__ __forceinline__ void load2(uint64_t, uint64_t&, uint64_t&) const {}
  __device__ __forceinline__ uint64_t load1(uint64_t) const { return 0; }
  __device__ __forceinline__ void put(uint64_t, uint64_t) const {}
  __device__ __forceinline__ void put2(uint64_t, uint64_t, uint64_t) const {}
  __device__ __forceinline__ uint32_t odd(uint64_t) const { return 0; }
};
// ---- value-ordered view of an index (bmx.h bmx_index_set_ordered): columns sorted by (value, position) ----
// src/bullet-query.js keeps an index as a Map keyed by VALUE (:30-73): equals() is one lookup, range() walks the distinct values. The view gives the
// device index the same shape: a query is two k-ary searches on the sorted value column + one contiguous copy, O(log R + matches).
__global__ __launch_bounds__(256) void k_iota_u32(uint32_t* __restrict__ p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) p[i] = (uint32_t)i;
}
// smallest and largest value of a column, tombstones (the type's minimum) aside: mm[0] = min, mm[1] = max (caller: INT64_MAX / INT64_MIN before the launch)
template <class T>
__global__ __launch_bounds__(256) void k_col_minmax(const T* __restrict__ v, uint64_t n, long long* __restrict__ mm) {
  constexpr T TOMB = sizeof(T) == 4 ? (T)INT32_MIN : (T)INT64_MIN;
  constexpr int E = 16 / sizeof(T);                       // elements per 16-byte load
  typedef T vec_t __attribute__((ext_vector_type(E)));
  long long lo = INT64_MAX, hi = INT64_MIN;
  auto take = [&](T x) { if (x != TOMB) { lo = x < lo ? (long long)x : lo; hi = x > hi ? (long long)x : hi; } };
  const uint64_t nv = n / E;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nv; i += (uint64_t)gridDim.x * 256u) {
    const vec_t x = reinterpret_cast<const vec_t*>(v)[i];
#pragma unroll
    for (int e = 0; e < E; e++) take(x[e]);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nv * E) take(v[nv * E + threadIdx.x]);      // the ragged tail
  for (int d = 32; d >= 1; d >>= 1) { const long long l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
  if ((threadIdx.x & 63u) == 0) { if (lo != INT64_MAX) atomicMin(mm, lo); if (hi != INT64_MIN) atomicMax(mm + 1, hi); }
}
// ids in sorted order, and the sorted keys (rebased by the sort: csrc/ordered_sort.hip) turned back into values in place
template <class T, class U>
__global__ __launch_bounds__(256) void k_gather_ids(const uint64_t* __restrict__ ids, const uint32_t* __restrict__ pos, uint64_t* __restrict__ out, uint64_t n, U* keys, T lo) {
  constexpr T TOMB = sizeof(T) == 4 ? (T)INT32_MIN : (T)INT64_MIN;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
    out[i] = ids[pos[i]];
    const U k = keys[i];
    reinterpret_cast<T*>(keys)[i] = k == 0 ? TOMB : (T)((U)(k - 1) + (U)lo);
  }
}
// One wave per bound, 64-ary search: every round 63 lanes probe evenly spaced keys of [L, R) and the ballot says between which two the bound lies
// (a 100M-row column: five dependent rounds instead of the 27 of a binary search). UPPER = false: first index with v >= key; true: first with v > key.
template <class T, bool UPPER>
__device__ __forceinline__ uint64_t ordered_bound(const T* __restrict__ v, uint64_t n, T key) {
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t L = 0, R = n;                         // invariant: every index < L is in front of the bound, every index >= R behind it
  while (R - L > 64) {
    const uint64_t step = (R - L + 63) / 64;
    const uint64_t c = L + (uint64_t)lane * step;        // lane 0 probes nothing (c == L)
    bool before = false;
    if (lane > 0 && c < R) { const T x = v[c]; before = UPPER ? x <= key : x < key; }
    const uint32_t t = (uint32_t)__popcll(__ballot(before));        // sorted column: lanes 1..t are in front of the bound
    const uint64_t nl = t ? L + (uint64_t)t * step + 1 : L;
    const uint64_t cr = L + (uint64_t)(t + 1) * step;
    R = (t < 63 && cr < R) ? cr : R;
    L = nl;
  }
  bool before = false;
  if (L + lane < R) { const T x = v[L + lane]; before = UPPER ? x <= key : x < key; }
  return L + (uint64_t)__popcll(__ballot(before));
}
// ab[0] = first match, ab[1] = one past the last; *n_out = matches (optional). An empty range (lo > hi) matches nothing.
template <class T>
__global__ __launch_bounds__(128) void k_ordered_bounds(const T* __restrict__ v, uint64_t n, T lo, T hi, unsigned long long* __restrict__ ab, unsigned long long* __restrict__ n_out,
                                                        uint32_t zero_count /* the run is only a candidate list (k_ordered_filter counts the matches itself) */) {
  __shared__ unsigned long long sh[2];
  const uint32_t w = threadIdx.x >> 6;
  uint64_t r = 0;
  if (lo <= hi) r = w == 0 ? ordered_bound<T, false>(v, n, lo) : ordered_bound<T, true>(v, n, hi);
  if ((threadIdx.x & 63u) == 0) sh[w] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long a = sh[0], b = sh[1] < sh[0] ? sh[0] : sh[1];
    ab[0] = a; ab[1] = b;
    if (n_out) *n_out = zero_count ? 0ull : b - a;
  }
}
// out[k] = src[a + k] for k < min(b - a, cap): the matches are one contiguous run of the sorted id (or position) column
template <class OutT>
__global__ __launch_bounds__(256) void k_ordered_copy(const OutT* __restrict__ src, const unsigned long long* __restrict__ ab, OutT* __restrict__ out, uint64_t cap) {
  const uint64_t a = ab[0];
  uint64_t m = ab[1] - a;
  if (m > cap) m = cap;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256u) out[i] = __builtin_nontemporal_load(src + a + i);
}

struct FinishCount {  // total -> *n_out (device), optional
  unsigned long long* n_out;
  __device__ void operator()(uint64_t total, uint32_t*) const { if (n_out && threadIdx.x == 0) *n_out = total; }
};

// winner bytes -> bit mask: 16 flags (16 B) per lane (used by k_compact_winners)
struct PredWinner {
  static constexpr int E = 16;
  const uint8_t* w;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
    if (first + 16 <= n) {
      uint4 x = *reinterpret_cast<const uint4*>(w + first);
      uint32_t q[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // bytes are 0/1: gather bit 0 of each byte
        uint32_t b = q[k] & 0x01010101u;
        m |= (((b * 0x10204080u) >> 28) & 0xFu) << (4 * k);
      }
    } else {
      for (int e = 0; e < 16 && first + e < n; e++) m |= (uint32_t)(w[first + e] & 1u) << e;
    }
    return m;
  }
};
struct EmitApplied {
  uint32_t* out;
  __device__ void operator()(uint64_t pos, uint64_t j) const { if (out) out[pos] = (uint32_t)j; }
};
struct FinishMerge {  // totals -> caller; fold and clear the sharded per-batch counters
  unsigned long long* n_applied; bmx_merge_stats* stats;
  unsigned long long* shard_ctr; unsigned long long* row_count;
  // optional: {rows, batch sequence number} in host memory the GPU can write (the host bounds the row count from it without a sync)
  unsigned long long* host_mirror = nullptr; unsigned long long seq = 0;
  // optional (bmx_merge_notify): words, possibly in other GPUs' memory, that learn how many merges this context has finished — the origins of the
  // direct exchange reuse a receive slab set only after the merge that read it (this launch is the merge's last: the table and the slabs are done with)
  SeqPtrs notify{}; uint32_t n_notify = 0; unsigned long long notify_value = 0;
  __device__ void operator()(uint64_t total, uint32_t* lds4) const {
    static_assert(CTR_SHARDS == SEL_THREADS, "one counter shard per thread");
    unsigned long long* c = shard_ctr + (size_t)threadIdx.x * CTR_STRIDE;
    uint32_t rows = (uint32_t)c[0], conf = (uint32_t)c[1];
    c[0] = 0; c[1] = 0;
    uint32_t trows, tconf;
    block_excl_scan(rows, trows, lds4);
    block_excl_scan(conf, tconf, lds4);
    if (threadIdx.x == 0) {
      unsigned long long r = *row_count + trows;
      *row_count = r;
      if (host_mirror) {   // rows first, then the sequence number: a host that reads the number first never pairs it with an OLDER count
        __hip_atomic_store(host_mirror, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_mirror + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (n_applied) *n_applied = total;
      if (stats) { stats->n_applied = total; stats->n_conflicts = tconf; stats->n_rows = r; stats->reserved = 0; }
    }
    // (relaxed: nothing is published with it — it only says that the launches BEFORE this one are done with their input)
    if (threadIdx.x < n_notify && notify.p[threadIdx.x]) __hip_atomic_store(notify.p[threadIdx.x], notify_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
};

// K3 proper: ordered compaction of the winner bytes into applied_idx. The per-256-delta winner counts were left by
// k_probe_apply (and adjusted by k_resolve_lists), so a block gets its global rank from a prefix over counts written by
// EARLIER launches: no counting phase, no in-launch hand-off. One block = 4096 deltas = 16 count entries.
template <class Finish>
__global__ __launch_bounds__(SEL_THREADS) void k_compact_winners(const uint8_t* __restrict__ wflag, const uint32_t* __restrict__ blk_info, uint32_t n,
                                                                  uint32_t* __restrict__ applied, Finish Fin, ChgLog L, uint32_t mark_created = 0) {
  __shared__ uint32_t wsum[4];
  // both loads are issued before the first wait: this block's 16 winner bytes per lane and its share of the count prefix
  PredWinner P{wflag};
  const uint64_t first = (uint64_t)blockIdx.x * 4096u + (uint64_t)threadIdx.x * 16u;
  uint32_t m = first < n ? P.mask(first, n) : 0u;
  const uint32_t first_kb = blockIdx.x * 16u;
  uint32_t part = 0;
  if ((reinterpret_cast<uintptr_t>(blk_info) & 15u) == 0) {
    // 16-byte loads, four in flight per lane: a plain "load, add" loop waits one L2 round trip per iteration (16 of them for the
    // last block of a 1M-delta batch, which was most of this kernel's time)
    const uint4* bi4 = reinterpret_cast<const uint4*>(blk_info);
    const uint32_t nq = first_kb / 4;            // first_kb is a multiple of 16
    for (uint32_t q0 = 0; q0 < nq; q0 += 4 * SEL_THREADS) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t q = q0 + (uint32_t)u * SEL_THREADS + threadIdx.x;
        v[u] = q < nq ? bi4[q] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) part += (v[u].x & BLK_COUNT) + (v[u].y & BLK_COUNT) + (v[u].z & BLK_COUNT) + (v[u].w & BLK_COUNT);
    }
  } else {
    for (uint32_t k = threadIdx.x; k < first_kb; k += SEL_THREADS) part += blk_info[k] & BLK_COUNT;
  }
  uint32_t offset;
  block_excl_scan(part, offset, wsum);
  uint32_t tot;
  uint32_t lp = block_excl_scan((uint32_t)__popc(m), tot, wsum);
  if (applied || L.chg) {
    // winners of this block in order through LDS, then consecutive lanes store consecutive ranks (a lane's own run of up to 16
    // indices would be 16 scattered 4-byte stores per wave instruction)
    __shared__ uint32_t loc[4096];
    while (m) {
      int e = __ffs((int)m) - 1;
      m &= m - 1;
      loc[lp++] = (uint32_t)first + (uint32_t)e;
    }
    __syncthreads();
    if (applied) {
      if (mark_created)   // BMX_MERGE_MARK_CREATED: bit 31 on the winners that created their row
        for (uint32_t i = threadIdx.x; i < tot; i += SEL_THREADS) applied[offset + i] = loc[i] | ((wflag[loc[i]] & W_FIRSTWRITE) ? 0x80000000u : 0u);
      else
        for (uint32_t i = threadIdx.x; i < tot; i += SEL_THREADS) applied[offset + i] = loc[i];
    }
    if (L.chg) {   // index change log: (slot | created, field) of every winner, behind the entries of the batches before
      const unsigned long long base = *L.base + offset;
      for (uint32_t i = threadIdx.x; i < tot; i += SEL_THREADS) {
        const uint32_t j = loc[i];
        const uint32_t f = L.recs ? L.recs[j].field : L.field[j];
        if (base + i < L.cap) L.chg[base + i] = make_uint2(L.slot_of[j] | ((wflag[j] & W_CREATED) ? CHG_CREATED : 0u), f);
      }
    }
  }
  if (blockIdx.x == gridDim.x - 1) {
    if (L.chg && threadIdx.x == 0) *L.next = *L.base + offset + tot;
    Fin((uint64_t)offset + tot, wsum);
  }
}

// winners per 256-delta block (what k_compact_winners ranks from) out of a byte map: one 16-byte load per lane, 4096 deltas per workgroup.
// (The merge kernels leave these counts themselves; this is for byte maps filled otherwise: the winners of a host batch over several shards.)
__global__ __launch_bounds__(256) void k_count_winners(const uint8_t* __restrict__ wflag, uint32_t n, uint32_t* __restrict__ blk_info) {
  const uint64_t first = (uint64_t)blockIdx.x * 4096u + (uint64_t)threadIdx.x * 16u;
  uint32_t c = 0;
  if (first + 16 <= n) {
    const uint4 x = *reinterpret_cast<const uint4*>(wflag + first);
    c = __popc(x.x & 0x01010101u) + __popc(x.y & 0x01010101u) + __popc(x.z & 0x01010101u) + __popc(x.w & 0x01010101u);
  } else {
    for (uint32_t e = 0; e < 16 && first + e < n; e++) c += wflag[first + e] & 1u;
  }
  // sum over the 16 lanes that share a 256-delta block
  c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4); c += __shfl_xor(c, 8);
  const uint32_t kb = blockIdx.x * 16u + (threadIdx.x >> 4);
  if ((threadIdx.x & 15u) == 0 && (uint64_t)kb * 256u < n) blk_info[kb] = c;
}

__global__ void k_sum_counts(const uint32_t* block_counts, uint32_t nblocks, unsigned long long* n_out) {
  __shared__ uint32_t wsum[4];
  uint32_t part = strided_partial_sum(block_counts, nblocks);
  uint32_t tot;
  block_excl_scan(part, tot, wsum);
  if (threadIdx.x == 0) *n_out = tot;
}

}  // namespace bmx
