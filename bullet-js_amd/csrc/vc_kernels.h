// vc_kernels.h — N4: K-writer vector-clock rows, gfx950. 64-byte slots:
//   { id u64 | field u32 | head u32 | val i64 | state u32 | keyset u32 | clock u32[8] }
// keyset = WHICH of the K writers the clock object names and in which order (eight 4-bit writer indices, 0xF = end): the reference's clocks are JS
// objects, a missing key counts as 0 when clocks are compared (src/bullet-crt.js:76-79) but two clocks are "identical" only if their JSON texts are
// (:200-203) — same keys, same order —, and a merged clock lists the incoming clock's keys first, then the stored clock's other keys (:103-114).
// Two launches per batch, no optimistic writes (the outcome for concurrent clocks depends on the order of the deltas):
//   k_vc_link    every delta finds/creates its row, claims it (one atomicExch) and links into the row's list
//   k_vc_resolve the LAST claimer of a row applies the row's deltas in index order with the reference's resolve()
//                (src/bullet-crt.js:164-279, general clocks), writes every delta's flags, the row's final state and the
//                last updating delta; rows are written by exactly one lane, after every read of the launch pair.
//                The list is in claim order, not index order: a lane orders a short list (<= VC_SHORT deltas) by repeated
//                selection; longer lists are queued for
//   k_vc_resolve_long  one workgroup per long row: one walk gathers the member indices, a bitmap over the batch puts them in
//                ascending order (O(n/32 + m), no comparison sort), the deltas are loaded 256 at a time by the whole workgroup
//                and applied in order by one lane. Linear in the list length: 10^5 deltas on one key take tens of milliseconds
//                (the dependent walk), not the hours a quadratic selection would.
#pragma once
#include "slot.h"
#include "merge_kernels.h"
#include "select.h"

namespace bmx {

constexpr int VC_MAXK = 8;
constexpr uint32_t VC_ABSENT = 0, VC_DENSE = 1, VC_SPARSE = 2;
constexpr uint32_t VC_SHORT = 16;        // lists up to this length are ordered by selection inside k_vc_resolve (<= 256 hops)
constexpr uint32_t VC_LONG_WGS = 64;     // workgroups of k_vc_resolve_long (each owns one bitmap over the batch)
constexpr uint32_t VC_QUEUED = 0xFFFFFFFEu;

constexpr uint32_t VC_KS_NONE = 0xFFFFFFFFu;     // the clock {}
__host__ __device__ __forceinline__ uint32_t vc_ks_dense(uint32_t K) {      // all K writers, in the table's order
  uint32_t ks = VC_KS_NONE;
  for (uint32_t k = 0; k < K; k++) ks = (ks & ~(0xFu << (4 * k))) | (k << (4 * k));
  return ks;
}
__host__ __device__ __forceinline__ uint32_t vc_ks_single(uint32_t w) { return 0xFFFFFFF0u | w; }
// key order of mergeVectorClocks(in, cur) = {...in} followed by cur's keys that `in` does not have (src/bullet-crt.js:103-114)
__device__ __forceinline__ uint32_t vc_ks_merge(uint32_t in, uint32_t cur) {
  uint32_t have = 0, len = 0;
#pragma unroll
  for (int i = 0; i < VC_MAXK; i++) { const uint32_t w = (in >> (4 * i)) & 0xFu; if (w != 0xFu) { have |= 1u << w; len = (uint32_t)i + 1; } }
  uint32_t out = in;
#pragma unroll
  for (int i = 0; i < VC_MAXK; i++) {
    const uint32_t w = (cur >> (4 * i)) & 0xFu;
    if (w != 0xFu && !((have >> w) & 1u) && len < (uint32_t)VC_MAXK) { out = (out & ~(0xFu << (4 * len))) | (w << (4 * len)); have |= 1u << w; len++; }
  }
  return out;
}
// a well-formed key set: writer indices < K, each at most once, nothing behind the first 0xF; components of writers it does not name are zero
__device__ __forceinline__ bool vc_ks_valid(uint32_t ks, const uint32_t* comps, uint32_t K) {
  uint32_t have = 0; bool ended = false, ok = true;
#pragma unroll
  for (int i = 0; i < VC_MAXK; i++) {
    const uint32_t w = (ks >> (4 * i)) & 0xFu;
    if (w == 0xFu) { ended = true; continue; }
    if (ended || w >= K || ((have >> w) & 1u)) ok = false;
    have |= 1u << (w & 7u);
  }
#pragma unroll
  for (int k = 0; k < VC_MAXK; k++) if ((uint32_t)k < K && !((have >> k) & 1u) && comps[k] != 0u) ok = false;
  return ok;
}

struct VcLongRow { uint32_t slot, head, m, base; };
struct VcLongCtl { uint32_t n_rows, cursor; };

struct alignas(64) VSlot {
  uint64_t id; uint32_t field; uint32_t head;
  int64_t val; uint32_t state; uint32_t keyset;
  uint32_t clock[VC_MAXK];
};
static_assert(sizeof(VSlot) == 64, "vc slot is 64 bytes");

struct VcArgs {
  VSlot* slots; uint64_t nslots;
  const uint64_t* id; const uint32_t* field; const uint32_t* clocks; const int64_t* val;
  const uint32_t* keysets;   // per delta, or null: every clock names all K writers in the table's order
  uint32_t n, K, local, epoch;
  uint32_t* next; uint32_t* slot_of; uint8_t* wflag; uint8_t* flags; uint32_t* blk_info;
  unsigned long long* row_count; uint32_t* status;
  int load;   // 1: bulk preload (the highest-index delta of a key overwrites the row, dense)
  VcLongCtl* lctl; VcLongRow* lrows; uint32_t lrows_cap;   // queue of rows whose list is longer than VC_SHORT
  uint32_t* ord;                                           // member indices of the queued rows (segments of m entries)
  uint32_t* bitmap; uint32_t bitmap_words;                 // VC_LONG_WGS bitmaps of ceil(n/32) words, all zero between uses
};

__global__ __launch_bounds__(256) void k_vc_init(VSlot* slots, uint64_t nslots) {
  for (uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * 256u) {
    uint4* q = reinterpret_cast<uint4*>(slots + s);
    q[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, FIELD_PENDING, 0u);
    q[1] = make_uint4(0u, 0u, VC_ABSENT, VC_KS_NONE);
    q[2] = make_uint4(0u, 0u, 0u, 0u); q[3] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// epoch wrap: forget every claim tag (a stale head of the same epoch value would splice an old index into a new list)
__global__ __launch_bounds__(256) void k_vc_sweep_heads(VSlot* slots, uint64_t nslots) {
  for (uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * 256u) slots[s].head = 0u;
}

// growth: re-insert every row of the old table into a larger, initialised one (claim tags are dropped: head = 0)
__global__ __launch_bounds__(256) void k_vc_rehash(const VSlot* old_slots, uint64_t old_n, VSlot* slots, uint64_t nslots, uint32_t* status) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < old_n; i += (uint64_t)gridDim.x * 256u) {
    const uint4* q = reinterpret_cast<const uint4*>(old_slots + i);
    const uint4 lo = q[0];
    const uint64_t id = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    if (id == EMPTY_ID) continue;
    ProbeSeq<2> ps(id, lo.z, nslots);
    bool placed = false;
    for (uint64_t p = 0; p < nslots; ++p) {
      VSlot* sl = slots + ps.slot();
      if (atomicCAS(reinterpret_cast<unsigned long long*>(&sl->id), (unsigned long long)EMPTY_ID, (unsigned long long)id) == EMPTY_ID) {
        uint4* w = reinterpret_cast<uint4*>(sl);
        sl->field = lo.z; sl->head = 0u;
        w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
        placed = true;
        break;
      }
      ps.next();
    }
    if (!placed) atomicOr(status, ST_FULL);
  }
}

__global__ __launch_bounds__(256) void k_vc_link(VcArgs A) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const bool active = j < A.n;
  uint32_t slot = 0xFFFFFFFFu;
  bool created = false;
  if (active) {
    const uint64_t id = A.id[j]; const uint32_t field = A.field[j]; const int64_t v = A.val[j];
    bool valid = id != EMPTY_ID && field != FIELD_PENDING && v >= -VAL_MAX && v <= VAL_MAX;
    if (valid && A.keysets) {
      uint32_t in[VC_MAXK];
#pragma unroll
      for (int k = 0; k < VC_MAXK; k++) in[k] = ((uint32_t)k < A.K) ? A.clocks[(size_t)j * A.K + k] : 0u;
      valid = vc_ks_valid(A.keysets[j], in, A.K);
    }
    if (!valid) atomicOr(A.status, ST_RANGE);
    if (valid) {
      const uint32_t tag = (A.epoch << IDX_BITS) | j;
      ProbeSeq<2> ps(id, field, A.nslots);
      bool found = false, failed = false;
      uint64_t p = 0;
      // probe rounds: a lane that meets a slot of its node whose field is still unpublished leaves the probe loop and waits BEHIND the
      // loop's exit, where the wave has reconverged and any sibling lane that created a slot has issued its publishing store
      // (same structure and argument as probe_or_insert in merge_kernels.h)
      for (uint32_t round = 0; round < 256 && !found && !failed; ++round) {
        VSlot* wait_on = nullptr;
        uint64_t s = 0;
        for (; p < A.nslots; ++p) {
          s = ps.slot();
          VSlot* sl = A.slots + s;
          uint4 lo = reinterpret_cast<const uint4*>(sl)[0];
          uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
          uint32_t sf = lo.z;
          if (sid == EMPTY_ID) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&sl->id), (unsigned long long)EMPTY_ID, (unsigned long long)id);
            if (old == EMPTY_ID) {   // created: one 8-byte store publishes the field and claims the head (see merge_kernels.h)
              __hip_atomic_store(reinterpret_cast<unsigned long long*>(&sl->field), (unsigned long long)field | ((unsigned long long)tag << 32),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              created = true; found = true; slot = (uint32_t)s;
              break;
            }
            sid = old; sf = FIELD_PENDING;
          }
          if (sid == id) {
            if (sf == FIELD_PENDING) { wait_on = sl; break; }
            if (sf == field) {
              uint32_t prev = atomicExch(&sl->head, tag);
              if ((prev >> IDX_BITS) == A.epoch) A.next[j] = (A.epoch << IDX_BITS) | (prev & IDX_MASK);
              found = true; slot = (uint32_t)s;
              break;
            }
          }
          ps.next();
        }
        if (found) break;
        if (!wait_on) { failed = true; break; }
        __builtin_amdgcn_wave_barrier();
        uint32_t sf = FIELD_PENDING, spins = 0;
        do { sf = __hip_atomic_load(&wait_on->field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (sf != FIELD_PENDING) break; __builtin_amdgcn_s_sleep(1); } while (++spins < (1u << 22));
        if (sf == FIELD_PENDING) { atomicOr(A.status, ST_SPIN); failed = true; slot = 0xFFFFFFFDu; break; }
        if (sf == field) {
          uint32_t prev = atomicExch(&wait_on->head, tag);
          if ((prev >> IDX_BITS) == A.epoch) A.next[j] = (A.epoch << IDX_BITS) | (prev & IDX_MASK);
          found = true; slot = (uint32_t)s;
          break;
        }
        ps.next(); ++p;
      }
      if (slot == 0xFFFFFFFDu) slot = 0xFFFFFFFFu;
      else
      if (!found && slot == 0xFFFFFFFFu) atomicOr(A.status, ST_FULL);
    }
    A.slot_of[j] = slot;
    A.wflag[j] = W_NONE;
    if (A.flags) A.flags[j] = 0;
  }
  // rows created by this block: one (non-returning) add per block, not per lane or wave (a single hot word is slow)
  __shared__ uint32_t s_c[4];
  unsigned long long mc = __ballot(created);
  if (lane_id() == 0) s_c[threadIdx.x >> 6] = (uint32_t)__popcll(mc);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (c) atomicAdd(A.row_count, (unsigned long long)c);
    A.blk_info[blockIdx.x] = 0;
    if (blockIdx.x == 0) { A.lctl->n_rows = 0; A.lctl->cursor = 0; }   // k_vc_resolve (the next launch) fills the queue
  }
}

// resolve() for one delta against the running row state held in registers (static indices only)
struct VcState { uint32_t c[VC_MAXK]; int64_t val; uint32_t state; uint32_t ks; };
__device__ __forceinline__ uint32_t vc_apply(VcState& R, const uint32_t* in, uint32_t in_ks, int64_t v, uint32_t K, uint32_t local) {
  if (R.state == VC_ABSENT) {                    // "no current state": clock {local: 2}, incoming clock dropped (:172-185)
#pragma unroll
    for (int k = 0; k < VC_MAXK; k++) R.c[k] = ((uint32_t)k == local) ? 2u : 0u;
    R.val = v; R.state = VC_SPARSE; R.ks = vc_ks_single(local);
    return BMX_FLAG_INCOMING;
  }
  bool in_ahead = false, cur_ahead = false, equal = true;
#pragma unroll
  for (int k = 0; k < VC_MAXK; k++) if ((uint32_t)k < K) {     // a writer a clock does not name holds 0 there: "|| 0" of :76-79
    if (in[k] > R.c[k]) in_ahead = true; else if (R.c[k] > in[k]) cur_ahead = true;
    if (in[k] != R.c[k]) equal = false;
  }
  const int cmp = (in_ahead && cur_ahead) ? 0 : (in_ahead ? 1 : (cur_ahead ? -1 : 0));
  const bool json_equal = equal && in_ks == R.ks;            // JSON.stringify equality (:200-203): same keys in the same order, same counters
  if (cmp == 0 && json_equal) {                  // identical clocks: value comparison (:200-233)
    if (v == R.val) return 0u;
    if (v > R.val) { R.val = v; return BMX_FLAG_INCOMING; }
    return BMX_FLAG_CURRENT;
  }
  if (cmp < 0) return BMX_FLAG_CURRENT | BMX_FLAG_HISTORICAL;    // :251-263
#pragma unroll
  for (int k = 0; k < VC_MAXK; k++) if ((uint32_t)k < K && in[k] > R.c[k]) R.c[k] = in[k];   // merged clock stored with the update
  R.ks = vc_ks_merge(in_ks, R.ks);
  R.state = VC_DENSE;
  if (cmp > 0) { R.val = v; return BMX_FLAG_INCOMING; }          // :236-248
  if (v >= R.val) R.val = v;                                      // concurrent: mergeValues on non-objects (:266-278, :133-135)
  return BMX_FLAG_CONCURRENT;
}

// one row's work for its last claimer; returns the index of the last delta that updated the row (or ~0u)
__device__ __forceinline__ uint32_t vc_resolve_row(const VcArgs& A, VSlot* sl, uint32_t head) {
  const uint4* q = reinterpret_cast<const uint4*>(sl);
  VcState R;
  {
    const uint4 mid = q[1], c0 = q[2], c1 = q[3];
    R.val = (int64_t)((uint64_t)mid.x | ((uint64_t)mid.y << 32)); R.state = mid.z; R.ks = mid.w;
    R.c[0] = c0.x; R.c[1] = c0.y; R.c[2] = c0.z; R.c[3] = c0.w; R.c[4] = c1.x; R.c[5] = c1.y; R.c[6] = c1.z; R.c[7] = c1.w;
  }
  const uint32_t dense_ks = vc_ks_dense(A.K);
  uint32_t last_upd = ~0u;
  if (A.load) {
    // bulk preload: the highest-index delta of the key overwrites the row (dense)
    uint32_t best = head, idx = head, steps = 0;
    for (;;) { if (idx > best) best = idx; uint32_t nx = A.next[idx]; if ((nx >> IDX_BITS) != A.epoch) break; idx = nx & IDX_MASK; if (++steps > A.n) break; }
#pragma unroll
    for (int k = 0; k < VC_MAXK; k++) R.c[k] = ((uint32_t)k < A.K) ? A.clocks[(size_t)best * A.K + k] : 0u;
    R.val = A.val[best]; R.state = VC_DENSE; R.ks = A.keysets ? A.keysets[best] : dense_ks; last_upd = best;
  } else {
    // how long is the list? (one walk)
    uint32_t m = 1;
    { uint32_t idx = head; for (;;) { uint32_t nx = A.next[idx]; if ((nx >> IDX_BITS) != A.epoch) break; idx = nx & IDX_MASK; if (++m > A.n) { atomicOr(A.status, ST_SPIN); return ~0u; } } }
    if (m > VC_SHORT) {
      // long list: a workgroup of k_vc_resolve_long orders and applies it (selection by one lane would be quadratic)
      const uint32_t e = atomicAdd(&A.lctl->n_rows, 1u);
      if (e >= A.lrows_cap) { atomicOr(A.status, ST_SPIN); return ~0u; }   // cannot happen: a batch of n deltas has fewer than n/VC_SHORT long lists
      const uint32_t base = atomicAdd(&A.lctl->cursor, m);
      A.lrows[e] = VcLongRow{(uint32_t)(sl - A.slots), head, m, base};
      return VC_QUEUED;
    }
    // apply the row's deltas in index order: each round walks the list for the smallest index above the last one applied
    uint32_t done = 0, prev_idx = 0; bool first_round = true;
    for (uint32_t guard = 0; guard <= A.n; guard++) {
      uint32_t pick = ~0u, idx = head, steps = 0, members = 0;
      for (;;) {
        members++;
        if ((first_round || idx > prev_idx) && idx < pick) pick = idx;
        uint32_t nx = A.next[idx];
        if ((nx >> IDX_BITS) != A.epoch) break;
        idx = nx & IDX_MASK;
        if (++steps > A.n) { atomicOr(A.status, ST_SPIN); return ~0u; }
      }
      if (pick == ~0u) break;
      uint32_t in[VC_MAXK];
#pragma unroll
      for (int k = 0; k < VC_MAXK; k++) in[k] = ((uint32_t)k < A.K) ? A.clocks[(size_t)pick * A.K + k] : 0u;
      const uint32_t fl = vc_apply(R, in, A.keysets ? A.keysets[pick] : dense_ks, A.val[pick], A.K, A.local);
      if (A.flags) A.flags[pick] = (uint8_t)fl;
      if (fl & (BMX_FLAG_INCOMING | BMX_FLAG_CONCURRENT)) last_upd = pick;
      prev_idx = pick; first_round = false;
      if (++done >= members) break;
    }
  }
  // one writer per row, after all reads of this launch pair
  uint4* w = reinterpret_cast<uint4*>(sl);
  w[1] = make_uint4((uint32_t)(uint64_t)R.val, (uint32_t)((uint64_t)R.val >> 32), R.state, R.ks);
  w[2] = make_uint4(R.c[0], R.c[1], R.c[2], R.c[3]);
  w[3] = make_uint4(R.c[4], R.c[5], R.c[6], R.c[7]);
  return last_upd;
}

__global__ __launch_bounds__(256) void k_vc_resolve(VcArgs A) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  uint32_t last_upd = ~0u;
  if (j < A.n) {
    const uint32_t slot = A.slot_of[j];
    if (slot != 0xFFFFFFFFu) {
      VSlot* sl = A.slots + slot;
      const uint32_t head = __builtin_nontemporal_load(&sl->head) & IDX_MASK;
      if (head == j) last_upd = vc_resolve_row(A, sl, head);     // only the last claimer of a row works
      if (last_upd == VC_QUEUED) last_upd = ~0u;                  // its winner mark comes from k_vc_resolve_long
    }
  }
  // winner marks + the per-256-delta counts k_compact_winners ranks with. Winners inside this block's own 256 deltas
  // (every singleton row) are counted with one add per wave; only a winner that sits in another block pays its own atomic.
  if (last_upd != ~0u) A.wflag[last_upd] = W_WINNER;
  const bool own = last_upd != ~0u && (last_upd >> 8) == blockIdx.x;
  if (last_upd != ~0u && !own) atomicAdd(&A.blk_info[last_upd >> 8], 1u);
  const unsigned long long m = __ballot(own);
  if (lane_id() == 0 && m) atomicAdd(&A.blk_info[blockIdx.x], (uint32_t)__popcll(m));
}

// Long lists, one workgroup per row (fixed grid; workgroup w takes queue entries w, w + VC_LONG_WGS, ...).
__global__ __launch_bounds__(256) void k_vc_resolve_long(VcArgs A) {
  __shared__ uint32_t s_clk[256][VC_MAXK];
  __shared__ int64_t s_val[256];
  __shared__ uint32_t s_idx[256];
  __shared__ uint8_t s_fl[256];
  __shared__ uint32_t s_ks[256];
  __shared__ uint32_t wsum[4];
  const uint32_t nrows = min(A.lctl->n_rows, A.lrows_cap);
  uint32_t* bm = A.bitmap + (size_t)blockIdx.x * A.bitmap_words;
  for (uint32_t e = blockIdx.x; e < nrows; e += gridDim.x) {
    const VcLongRow r = A.lrows[e];
    uint32_t* ord = A.ord + r.base;
    // 1. one walk: member indices in claim order
    if (threadIdx.x == 0) {
      uint32_t idx = r.head;
      for (uint32_t i = 0; i < r.m; i++) { ord[i] = idx; const uint32_t nx = A.next[idx]; if ((nx >> IDX_BITS) != A.epoch) break; idx = nx & IDX_MASK; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 2. ascending order through a bitmap over the batch (atomics execute at the memory side; the scan below reads past L1)
    for (uint32_t i = threadIdx.x; i < r.m; i += 256) { const uint32_t idx = __builtin_nontemporal_load(&ord[i]); atomicOr(&bm[idx >> 5], 1u << (idx & 31u)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t per = (A.bitmap_words + 255u) / 256u;
    const uint32_t w0 = min(A.bitmap_words, threadIdx.x * per), w1 = min(A.bitmap_words, w0 + per);
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; w++) cnt += __popc(__builtin_nontemporal_load(&bm[w]));
    uint32_t tot;
    uint32_t pos = block_excl_scan(cnt, tot, wsum);
    for (uint32_t w = w0; w < w1; w++) {
      uint32_t bits = __builtin_nontemporal_load(&bm[w]);
      if (bits) atomicAnd(&bm[w], 0u);          // leave the bitmap zero for the next row
      while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; ord[pos++] = (w << 5) + (uint32_t)b; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 3. apply in index order: the workgroup loads 256 deltas at a time, lane 0 runs resolve() over them
    VcState R; uint32_t last_upd = ~0u;
    VSlot* sl = A.slots + r.slot;
    if (threadIdx.x == 0) {
      const uint4* q = reinterpret_cast<const uint4*>(sl);
      const uint4 mid = q[1], c0 = q[2], c1 = q[3];
      R.val = (int64_t)((uint64_t)mid.x | ((uint64_t)mid.y << 32)); R.state = mid.z; R.ks = mid.w;
      R.c[0] = c0.x; R.c[1] = c0.y; R.c[2] = c0.z; R.c[3] = c0.w; R.c[4] = c1.x; R.c[5] = c1.y; R.c[6] = c1.z; R.c[7] = c1.w;
    }
    for (uint32_t c0 = 0; c0 < tot; c0 += 256) {
      const uint32_t i = c0 + threadIdx.x;
      if (i < tot) {
        const uint32_t idx = __builtin_nontemporal_load(&ord[i]);
        s_idx[threadIdx.x] = idx; s_val[threadIdx.x] = A.val[idx]; s_ks[threadIdx.x] = A.keysets ? A.keysets[idx] : vc_ks_dense(A.K);
#pragma unroll
        for (int k = 0; k < VC_MAXK; k++) s_clk[threadIdx.x][k] = ((uint32_t)k < A.K) ? A.clocks[(size_t)idx * A.K + k] : 0u;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        const uint32_t lim = min(256u, tot - c0);
        for (uint32_t x = 0; x < lim; x++) {
          uint32_t in[VC_MAXK];
#pragma unroll
          for (int k = 0; k < VC_MAXK; k++) in[k] = s_clk[x][k];
          const uint32_t fl = vc_apply(R, in, s_ks[x], s_val[x], A.K, A.local);
          s_fl[x] = (uint8_t)fl;
          if (fl & (BMX_FLAG_INCOMING | BMX_FLAG_CONCURRENT)) last_upd = s_idx[x];
        }
      }
      __syncthreads();
      if (i < tot && A.flags) A.flags[s_idx[threadIdx.x]] = s_fl[threadIdx.x];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      uint4* w = reinterpret_cast<uint4*>(sl);
      w[1] = make_uint4((uint32_t)(uint64_t)R.val, (uint32_t)((uint64_t)R.val >> 32), R.state, R.ks);
      w[2] = make_uint4(R.c[0], R.c[1], R.c[2], R.c[3]);
      w[3] = make_uint4(R.c[4], R.c[5], R.c[6], R.c[7]);
      if (last_upd != ~0u) { A.wflag[last_upd] = W_WINNER; atomicAdd(&A.blk_info[last_upd >> 8], 1u); }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_vc_get(const VSlot* slots, uint64_t nslots, uint32_t n, uint32_t K, const uint64_t* id, const uint32_t* field,
                                                uint32_t* clocks, int64_t* val, uint8_t* state, uint32_t* keysets) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  const uint64_t kid = id[j]; const uint32_t kf = field[j];
  ProbeSeq<2> ps(kid, kf, nslots);
  uint8_t st = VC_ABSENT; int64_t v = 0; uint32_t ks = VC_KS_NONE; uint32_t c[VC_MAXK] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint64_t p = 0; p < nslots; ++p) {
    const uint4* q = reinterpret_cast<const uint4*>(slots + ps.slot());
    uint4 lo = q[0];
    uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    if (sid == EMPTY_ID) break;
    if (sid == kid && lo.z == kf) {
      uint4 mid = q[1], c0 = q[2], c1 = q[3];
      v = (int64_t)((uint64_t)mid.x | ((uint64_t)mid.y << 32)); st = (uint8_t)mid.z; ks = mid.w;
      c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
      break;
    }
    ps.next();
  }
  val[j] = v; state[j] = st;
  if (keysets) keysets[j] = ks;
#pragma unroll
  for (int k = 0; k < VC_MAXK; k++) if ((uint32_t)k < K) clocks[(size_t)j * K + k] = c[k];
}

// scan of the vector-clock table itself (range()/equals()/count() of src/bullet-query.js:186-313 over K-writer rows): rows of `field` with
// lo <= val <= hi, in slot order. No dense index column here: 64 bytes are read per slot (a 10M-row table at load 0.5: 1.3 GB, ~0.25 ms).
struct PredVSlotRange {
  static constexpr int E = 2;
  const VSlot* slots; uint32_t field; int64_t lo, hi;
  __device__ uint32_t mask(uint64_t first, uint64_t n) const {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const uint64_t s = first + e;
      if (s < n) {
        const uint4* q = reinterpret_cast<const uint4*>(slots + s);
        const uint4 a = q[0];
        if (!(a.x == 0xFFFFFFFFu && a.y == 0xFFFFFFFFu) && a.z == field) {
          const uint4 b = q[1];
          const int64_t v = (int64_t)((uint64_t)b.x | ((uint64_t)b.y << 32));
          if (b.z != VC_ABSENT && v >= lo && v <= hi) m |= 1u << e;
        }
      }
    }
    return m;
  }
};
struct EmitVIds {
  const VSlot* slots; uint64_t* out; uint64_t cap;
  __device__ void operator()(uint64_t pos, uint64_t s) const { if (out && pos < cap) out[pos] = slots[s].id; }
};

}  // namespace bmx
