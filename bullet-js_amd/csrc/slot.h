// slot.h — resident-table slot layout, hashing and device status bits (gfx950 only).
//
// The resident graph is ONE open-addressed hash table of 32-byte AoS slots in HBM:
//     { id u64 | field u32 | head u32 | ts i64 | val i64 }
// A probe touches exactly one 128-byte L2 line whatever it reads (measured: profiles/r01_micro_probe_*.log:
// 16 B, 32 B or a whole line per probe all cost ~20 us per 1M probes; four SoA columns cost 3.4x), so the
// row's key, clock and value share a 32-byte sector and the (ts,val) pair is one aligned 16-byte access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmx {

constexpr uint64_t EMPTY_ID = ~0ull;            // reserved node id: empty slot
constexpr uint32_t FIELD_PENDING = 0xFFFFFFFFu; // reserved field: slot claimed, field not yet published
constexpr int64_t TS_NEW = INT64_MIN;           // ts of an empty slot / of a row created in the running batch
constexpr int64_t TS_MAX = (1ll << 53) - 1;     // JS safe-integer range
constexpr int64_t VAL_MAX = (1ll << 53) - 1;
constexpr int64_t VAL_DELETED = INT64_MIN;      // tombstone (== BMX_VAL_DELETED): below every legal value, so any delta at the tombstone's ts or later wins against it
constexpr uint32_t IDX_BITS = 24;               // batch index bits in a head / next tag
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1;
constexpr uint32_t MAX_BATCH = 1u << IDX_BITS;
constexpr uint32_t EPOCH_MAX = 255;             // 8-bit batch epoch in the tag; heads are swept when it wraps
// A row created in batch `epoch` stores ts | (epoch << 53) until the next epoch sweep: bits 53..60 of the stored
// clock are free because ts <= 2^53-1. A later delta of the same key in the same batch sees, in ONE aligned
// 16-byte load, both the value and the fact that the row has no pre-batch state.
constexpr int TS_MARK_SHIFT = 53;
constexpr int64_t TS_VALUE_MASK = (1ll << 53) - 1;
__host__ __device__ inline int64_t ts_value(int64_t w) { return w & TS_VALUE_MASK; }
__host__ __device__ inline uint32_t ts_mark(int64_t w) { return (uint32_t)((uint64_t)w >> TS_MARK_SHIFT) & 0xFFu; }

// device status word bits (sticky until read by the host)
constexpr uint32_t ST_RANGE = 1, ST_FULL = 2, ST_SPIN = 4, ST_SLAB = 8;

struct alignas(32) Slot {
  uint64_t id;
  uint32_t field;
  uint32_t head;  // (epoch << 24) | index of the last delta that claimed this row in batch `epoch`
  int64_t ts;
  int64_t val;
};
static_assert(sizeof(Slot) == 32, "slot must be one 32-byte sector");

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// owner shard of a node id; independent of node_hash (below) so every shard sees uniformly hashed lines
__host__ __device__ inline uint64_t owner_hash(uint64_t id) { return mix64(id * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL); }

// Probe sequence of a key. A 128-byte line is a bucket (SPL slots: 4 rows of 32 B, or 2 vector-clock rows of 64 B): the line comes from
// the NODE id alone, the start inside the line from the field, and the line's slots are tried cyclically before the next line is.
// So the fields of one node share a line while it has room — a sync chunk that carries several fields of a node costs one fill, and one
// coalesced request when the rows sit on neighbouring lanes — and a lookup leaves its home line only if that line is full (random
// probing crossed a line boundary on 8 % of the probes at load factor 0.3). No deletions exist, so "reached an empty slot" still means absent.
__host__ __device__ inline uint64_t node_hash(uint64_t id) { return mix64(id * 0x9FB21C651E98DF25ULL + 0x632BE59BD9B4E019ULL); }
template <int SPL>
struct ProbeSeq {
  uint64_t line, nlines; uint32_t c, k;
  __device__ __forceinline__ ProbeSeq(uint64_t id, uint32_t field, uint64_t nslots) {
    nlines = nslots / SPL;                                   // nslots is a multiple of SPL
    const uint64_t h = node_hash(id);
    line = __umul64hi(h, nlines);
    // start inside the line: field-dependent, rotated per node (low hash bits, which the line index barely depends on) so that a
    // single-field graph does not put every row in the same 32-byte sector of its line
    c = (((field * 0x9E3779B9u) >> (SPL == 4 ? 30 : 31)) + (uint32_t)h) & (SPL - 1);
    k = 0;
  }
  __device__ __forceinline__ uint64_t slot() const { return line * SPL + ((c + k) & (SPL - 1)); }
  __device__ __forceinline__ void next() { if ((++k & (SPL - 1)) == 0) line = (line + 1 == nlines) ? 0 : line + 1; }
};

// all kernels here use 1-D blocks whose size is a multiple of 64, so the lane is the low 6 bits of threadIdx.x
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// wave64 inclusive prefix sum in 7 DPP adds (row_shr 1,2,3,4,8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3); disabled / out-of-row source lanes contribute 0 (old = 0, bound_ctrl)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  uint32_t x = v;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);
  return x;
}

// lexicographic compare of (ts,val) pairs: -1, 0, +1   (reference: clock compare then default value compare,
// src/bullet-crt.js:68-95 scalar form, :11-15)
__device__ __forceinline__ int lexcmp(int64_t ta, int64_t va, int64_t tb, int64_t vb) {
  if (ta != tb) return ta < tb ? -1 : 1;
  if (va != vb) return va < vb ? -1 : 1;
  return 0;
}

}  // namespace bmx
