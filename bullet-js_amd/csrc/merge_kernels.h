// merge_kernels.h — delta-vs-resident join and conflict resolution (K1/K2/K4 of SURVEY §2.1), gfx950.
//
// Replaces, for scalar clocks {w: ts}, the per-entry loop
//     for (entry of entries) bullet.setData(...)  ->  crt.handleUpdate -> crt.resolve
// (reference src/bullet-network-sync.js:551-569, src/bullet-crt.js:329-385, :164-279) by three launches
// whose joint effect on the resident rows and on the reported winners equals that loop's:
//
//   k_probe_apply    one lane per delta: probe the row's slot (ONE 128-B line), decide against the
//                    snapshot it saw, claim the row with ONE atomicExch on slot.head, and — if it is the
//                    first claimer of the batch — store (ts,val) right away (one aligned 16-B store).
//                    An absent key is created with a CAS on slot.id plus one 8-byte store that publishes the
//                    field and claims the head together.
//                    Later claimers of the same row (duplicate keys) link themselves BEHIND the previous claimer
//                    (next[previous] = me: a forward list that starts at the first claimer) instead of writing.
//   k_resolve_lists  first claimers that got followers only (none in a unique-key batch): each walks its row's list
//                    forward and applies the reference's sequential outcome (lexmax of (ts,val), ties to the
//                    smaller index, first write of an absent key stored with ts := 2). Followers do nothing there:
//                    who has to walk is known from next[own index], a coalesced load — no row is read to find out.
//   select (select.h) ordered compaction of the per-delta winner bytes -> applied_idx.
//
// Why this shape (measured, profiles/r01_micro_probe_v2.log): a random probe costs one 128-B line
// (~50 G lines/s), a dirty line costs its write-back, and global atomics run at ~25 G/s whatever the
// table size — so each delta gets exactly one line read, at most one atomic, at most one 16-B store,
// and nothing ever revisits the table unless two deltas of one batch share a key.
#pragma once
#include "slot.h"
#include "../../include/bmx.h"

namespace bmx {

// How long a device-side wait (a sequence word, arrival words, the deferred compaction's hand-off) polls before it gives up, sets the context's sticky
// ST_SPIN status and returns: 100 MHz ticks, ~60 s by default, bmx_set_wait_limit() changes it for the device. A wait also gives up AT ONCE when the
// context's status already carries ST_SPIN: after one hand-off has failed every wait still queued behind it drains immediately instead of taking its
// own full limit (round 5: a rank that raised in the middle of the direct exchange left its peer six queued waits = six minutes).
__device__ unsigned long long g_wait_ticks = 6000000000ull;
__device__ __forceinline__ bool wait_gave_up(unsigned long long t0, const uint32_t* status) {
  return wall_clock64() - t0 > g_wait_ticks || (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_SPIN) != 0;
}
constexpr int PART_MAX_SHARDS = 16;
struct SeqPtrs { unsigned long long* p[PART_MAX_SHARDS]; };

struct MergeArgs {
  Slot* slots;
  uint64_t nslots;
  const uint64_t* id;
  const uint32_t* field;
  const int64_t* ts;
  const int64_t* val;
  const bmx_delta_rec* recs;  // AoS input (id/field/ts/val unused then)
  uint32_t n;
  uint32_t epoch;             // 1..EPOCH_MAX
  uint32_t* next;             // per delta: (epoch<<24)|the claimer that came right after this one on the same row; stale epoch = end of list
                              // (strict mode links backward instead: next[j] = previous claimer)
  uint32_t* blk_follow;       // per 256-delta block: == epoch if a delta of the block got a follower (k_resolve_lists skips the others)
  uint8_t* wflag;             // per delta: W_WINNER = this delta's value is the row's final value,
                              //            W_PENDING = needs k_resolve_lists (duplicate key / reference-mode insert)
  uint8_t* flags;             // optional decision flags
  uint32_t* slot_of;          // per delta: the row's slot — of a first claimer that got a follower (written by that follower: k_resolve_lists starts from it) and,
                              // while an index change log is kept (log_slots), of every first claimer (the compaction logs the winners' rows)
  uint32_t* blk_info;         // per 256-delta block: winners in the block (this batch's half of the double buffer, zero when the batch starts)
  unsigned long long* shard_ctr;  // CTR_SHARDS x CTR_STRIDE counters: [s][0] rows created, [s][1] conflicts
  uint32_t* status;
  uint32_t* blk_next;         // the block summaries of the NEXT batch (the other half of the double buffer): zeroed by k_probe_apply
  uint32_t blk_ents;          // entries per half
  uint32_t force;             // bmx_put_rows: store every delta as given (unique keys, BMX_INSERT_DELTA); BMX_VAL_DELETED is a legal value then
  // deferred compaction (bmx.hip, "K3 under the next K1"): the compaction of batch b runs on a second stream while batch b + 1 is probed
  unsigned long long* started = nullptr;          // k_probe_apply's block 0 stores started_val here when it begins: every launch in front of it on the stream is done
  unsigned long long started_val = 0;
  const unsigned long long* k3_done = nullptr;    // k_resolve_lists' block 0 returns only once *k3_done >= k3_wait: the compaction that last read the
  unsigned long long k3_wait = 0;                 // workspace the NEXT batch writes has finished (it was released when this batch's probe kernel started)
  uint32_t* fld_out = nullptr;                    // per delta: its field hash, kept for a change log that is written after the caller's columns may be gone
  uint32_t log_slots = 0;                         // every first claimer records its row's slot (the compaction writes an index change log)
  // bmx_merge_notify with a deferred compaction: the batch BEFORE this one is done with its receive slabs as soon as this launch starts (its probe and
  // resolve kernels are in front of it on the stream), so block 0 tells the origins here instead of the compaction that runs under this kernel
  SeqPtrs notify{}; uint32_t n_notify = 0; unsigned long long notify_value = 0;
  // bmx_merge_tail_wait: k_resolve_lists' block 0 returns only once tail_n words (this GPU's memory, written by peers) are >= tail_at_least — the
  // arrival words of the NEXT batch's slabs, so that its probe kernel follows without a wait launch in between
  const unsigned long long* tail_words = nullptr; uint32_t tail_n = 0; unsigned long long tail_at_least = 0; unsigned long long* tail_diag = nullptr;
};

constexpr uint8_t W_NONE = 0, W_WINNER = 1, W_PENDING = 2, W_FIRST = 4;   // W_FIRST (bit): first claimer of its row in this batch; bit 0 is what the compaction reads
constexpr uint8_t W_CREATED = 8;   // (bit, on a winner) its row did not exist before this batch: the index change log appends it instead of updating it
constexpr uint8_t W_FIRSTWRITE = 16;   // (bit, on a winner) it is the delta that CREATED its row: what the row stores is the insert rule's clock (2 in reference mode), not the delta's own
constexpr uint32_t BLK_COUNT = 0x7FFFFFFFu;
// A single hot word takes only ~88 atomics/us (MI355X_MICROARCH.md "dequeue"), so per-batch counters are
// spread over 256 words on separate 128-B lines and folded once per batch by the last compaction block.
constexpr uint32_t CTR_SHARDS = 256, CTR_STRIDE = 16;

template <bool AOS>
__device__ __forceinline__ void load_delta(const MergeArgs& A, uint32_t j, uint64_t& id, uint32_t& field, int64_t& ts, int64_t& val) {
  if (AOS) {
    const uint4* p = reinterpret_cast<const uint4*>(A.recs + j);
    uint4 lo = p[0], hi = p[1];
    id = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    field = lo.z;
    ts = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
    val = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
  } else {
    id = A.id[j]; field = A.field[j]; ts = A.ts[j]; val = A.val[j];
  }
}
template <bool AOS>
__device__ __forceinline__ void load_delta_tv(const MergeArgs& A, uint32_t j, int64_t& ts, int64_t& val) {
  if (AOS) {
    uint4 hi = reinterpret_cast<const uint4*>(A.recs + j)[1];
    ts = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
    val = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
  } else {
    ts = A.ts[j]; val = A.val[j];
  }
}

__device__ __forceinline__ void store_tv(Slot* sl, int64_t ts, int64_t val) {
  uint4 v = make_uint4((uint32_t)(uint64_t)ts, (uint32_t)((uint64_t)ts >> 32), (uint32_t)(uint64_t)val, (uint32_t)((uint64_t)val >> 32));
  reinterpret_cast<uint4*>(sl)[1] = v;  // one aligned global_store_dwordx4: (ts,val) never tears
}

// Locate (or create) the slot of key (id, field). Returns false if the table is full / protocol fault.
// is_new: the row has no resident value visible to this lane (created in this batch, by anyone).
// created: this lane created the row AND (with the same 64-bit store that publishes the field) claimed it;
// prev_head is then the old head word.
//
// A lane that meets a slot of its own node whose field is not published yet (another lane created it a moment ago) must wait for
// that field. The wait is NOT inside the probe loop: the lane leaves the loop, and the wait sits behind the loop's exit, where
// the wave has reconverged — every sibling lane of the same wave that created a slot in this round has executed its publishing
// store by then (it is in the creating lane's path through the loop body, in front of that lane's exit). So a waiter can never
// spin in front of the store it is waiting for, whatever order the compiler gives the blocks inside the loop.
template <bool UNIQUE>
__device__ __forceinline__ bool probe_or_insert(const MergeArgs& A, uint32_t tag, uint64_t id, uint32_t field, uint64_t& slot_out,
                                                bool& is_new, bool& created, uint32_t& prev_head, int64_t& cts, int64_t& cval) {
  created = false;
  ProbeSeq<4> ps(id, field, A.nslots);
  uint64_t p = 0;
  for (uint32_t round = 0; round < 256; ++round) {
    Slot* wait_on = nullptr;
    uint64_t s = 0;
    int done = 0;
    for (; p < A.nslots; ++p) {
      s = ps.slot();
      Slot* sl = A.slots + s;
      const uint4* q = reinterpret_cast<const uint4*>(sl);
      uint4 lo = q[0], hi = q[1];
      uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
      uint32_t sf = lo.z;
      bool fresh = false;
      if (sid == EMPTY_ID) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&sl->id), (unsigned long long)EMPTY_ID, (unsigned long long)id);
        if (old == EMPTY_ID) {  // this lane created the row
          // ONE aligned 8-byte agent-scope store publishes the field and claims the head together. A store (not an
          // exchange) is enough: other lanes of this key wait for the field before they touch the head, so nobody
          // can have claimed it earlier and every later claimer's exchange returns this tag.
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(&sl->field), (unsigned long long)field | ((unsigned long long)(UNIQUE ? 0u : tag) << 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          prev_head = 0;
          created = true;
          slot_out = s; is_new = true; cts = TS_NEW; cval = 0;
          done = 1;
          break;
        }
        sid = old;            // somebody claimed it while we looked (or our L1 copy was stale)
        sf = FIELD_PENDING;   // read the field through L2
        fresh = true;         // whatever (ts,val) we loaded is not a resident value
      }
      if (sid == id) {
        if (sf == FIELD_PENDING) { wait_on = sl; break; }     // leave the loop; wait behind its exit
        if (sf == field) {
          int64_t t = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
          slot_out = s;
          // no pre-batch state: slot just claimed, still unwritten, or created earlier in this very batch
          is_new = fresh || t == TS_NEW || ts_mark(t) == (tag >> IDX_BITS);
          cts = is_new ? TS_NEW : ts_value(t);
          cval = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
          done = 1;
          break;
        }
      }
      ps.next();
    }
    if (done) return true;
    if (!wait_on) { atomicOr(A.status, ST_FULL); return false; }
    __builtin_amdgcn_wave_barrier();     // nothing of the wait moves in front of the loop exit
    uint32_t sf = FIELD_PENDING, spins = 0;
    do {
      sf = __hip_atomic_load(&wait_on->field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sf != FIELD_PENDING) break;
      __builtin_amdgcn_s_sleep(1);
    } while (++spins < (1u << 22));
    if (sf == FIELD_PENDING) { atomicOr(A.status, ST_SPIN); return false; }
    if (sf == field) {   // the row is being created in this very batch: no pre-batch state
      slot_out = s; is_new = true; cts = TS_NEW; cval = 0;
      return true;
    }
    ps.next(); ++p;      // another field of the same node: keep probing behind it
  }
  atomicOr(A.status, ST_SPIN);
  return false;
}

// What a delta does once its row is located (resolve(): src/bullet-crt.js:164-279, scalar clocks): decide against the snapshot the lane
// saw, claim the row, store if it is the first claimer of the batch, link behind the previous claimer otherwise.
// `created`: this lane created the row (it claimed the head with the publishing store: prev is the old head word).
template <int MODE, bool UNIQUE>
__device__ __forceinline__ void decide_and_claim(const MergeArgs& A, const uint32_t j, const uint64_t s, const bool is_new, const bool created, uint32_t prev,
                                                 const int64_t cts, const int64_t cval, const int64_t a, const int64_t v,
                                                 uint32_t& fl, uint32_t& wf, bool& conflict) {
  const uint32_t tag = (A.epoch << IDX_BITS) | j;
  const int c = (is_new || A.force) ? 1 : lexcmp(a, v, cts, cval);
  if (c < 0) {
    // strictly below a pre-batch value or a value stored in this batch by a delta of a resident row:
    // it can never be the final value
    fl = BMX_FLAG_CURRENT | (a < cts ? BMX_FLAG_HISTORICAL : 0u);
    return;
  }
  Slot* sl = A.slots + s;
  if (!created && !UNIQUE) prev = atomicExch(&sl->head, tag);   // UNIQUE: caller-guaranteed single claimer (prev stays 0)
  if ((prev >> IDX_BITS) != A.epoch) {
    // first claimer of this row in this batch: its snapshot is the pre-batch row
    // it walks the row's list in k_resolve_lists if anybody follows; its slot is recorded by that follower (a unique-key batch then writes no
    // slot_of[] at all: 0.8M scattered 4-byte stores = 4 MB of dirty sectors less per 1M-delta launch), or here for the index change log
    if (!UNIQUE) wf = W_FIRST;
    if (A.log_slots) A.slot_of[j] = (uint32_t)s;
    if (is_new) {
      // first write of an absent key: the reference stores clock {id:2} (src/bullet-crt.js:172-185);
      // the creation mark keeps later deltas of this key from comparing against this provisional value
      const int64_t t0 = (MODE == BMX_INSERT_REFERENCE) ? 2 : a;
      store_tv(sl, t0 | ((int64_t)A.epoch << TS_MARK_SHIFT), v);
      wf |= W_WINNER | W_CREATED | W_FIRSTWRITE; fl = BMX_FLAG_INCOMING;    // (a first claimer with followers is corrected by k_resolve_lists: the creating delta is the smallest index)
    } else if (c > 0) {
      store_tv(sl, a, v); wf |= W_WINNER; fl = BMX_FLAG_INCOMING;
    }  // c == 0: identical clock and value: no-op, all flags false
  } else {
    // duplicate key inside the batch: link behind the previous claimer; the row's first claimer resolves the list in k_resolve_lists
    A.next[prev & IDX_MASK] = tag;
    A.slot_of[prev & IDX_MASK] = (uint32_t)s;      // same row as the previous claimer's: this is where the first claimer of the list finds its row
    A.blk_follow[(prev & IDX_MASK) >> 8] = A.epoch;
    wf = W_PENDING; conflict = true;
    fl = c > 0 ? BMX_FLAG_INCOMING : 0u;
  }
}

// K1. One lane per delta, NO workgroup barrier and no LDS: a wave adds its winner count to its 256-delta block's summary with one global atomic
// that returns nothing; the summaries are double-buffered and this launch zeroes the half the NEXT batch will add into (its last readers, the
// compaction of the batch before, are long done). Waves retire on their own — NT = 64 makes every wave its own workgroup — and the CU takes new
// ones earlier. Measured against the rounds 1-2 kernel (block summary behind a __syncthreads) and against three ways of moving the row creations
// out of the probing waves (LDS-compacted behind a barrier, by the last wave, by a second launch = SURVEY §2.1 K4): profiles/r03_ab_inserts.log —
// the barrier cost 2-7 us per 1M-delta launch, every split of the creations cost more than it saved.
template <bool AOS, int MODE, bool UNIQUE, int NT>
__device__ __forceinline__ void probe_apply_body(const MergeArgs& A) {
  const uint32_t j = blockIdx.x * (uint32_t)NT + threadIdx.x;
  if (A.started && j == 0) __hip_atomic_store(A.started, A.started_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (A.n_notify && j < A.n_notify && A.notify.p[j]) __hip_atomic_store(A.notify.p[j], A.notify_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (uint32_t t = j; t < A.blk_ents; t += gridDim.x * (uint32_t)NT) A.blk_next[t] = 0u;
  const bool active = j < A.n;
  uint64_t id = EMPTY_ID; uint32_t field = 0; int64_t a = 0, v = 0;
  if (active) load_delta<AOS>(A, j, id, field, a, v);
  if (A.fld_out && active) A.fld_out[j] = field;
  const bool pad = AOS && id == EMPTY_ID;
  const bool valid = active && id != EMPTY_ID && field != FIELD_PENDING && a >= 0 && a <= TS_MAX && ((v >= -VAL_MAX && v <= VAL_MAX) || (A.force && v == VAL_DELETED));
  if (active && !valid && !pad) atomicOr(A.status, ST_RANGE);
  uint32_t fl = 0, wf = W_NONE;
  bool conflict = false, created = false;
  if (valid) {
    bool is_new; int64_t cts, cval; uint64_t s; uint32_t prev = 0;
    if (probe_or_insert<UNIQUE>(A, (A.epoch << IDX_BITS) | j, id, field, s, is_new, created, prev, cts, cval))
      decide_and_claim<MODE, UNIQUE>(A, j, s, is_new, created, prev, cts, cval, a, v, fl, wf, conflict);
  }
  if (active) {
    A.wflag[j] = (uint8_t)wf;
    if (A.flags) A.flags[j] = (uint8_t)fl;
  }
  const uint32_t nc = (uint32_t)__popcll(__ballot(created)), nx = (uint32_t)__popcll(__ballot(conflict));
  const uint32_t nw = (uint32_t)__popcll(__ballot((wf & W_WINNER) != 0));
  if (lane_id() == 0) {
    if (nw) atomicAdd(&A.blk_info[j >> 8], nw);       // lane 0's 64 deltas lie inside one 256-delta block
    if (nc | nx) {
      unsigned long long* ctr = A.shard_ctr + (size_t)((j >> 6) & (CTR_SHARDS - 1)) * CTR_STRIDE;
      if (nc) atomicAdd(ctr + 0, (unsigned long long)nc);
      if (nx) atomicAdd(ctr + 1, (unsigned long long)nx);
    }
  }
}

template <bool AOS, int MODE, bool UNIQUE, int NT>
__global__ __launch_bounds__(NT) void k_probe_apply(MergeArgs A) { probe_apply_body<AOS, MODE, UNIQUE, NT>(A); }
// The same kernel held to at most SIX (FIVE) resident waves per SIMD instead of eight. The probe kernel is bound by memory-side requests in flight (a CU's 64-entry
// miss queue is full with far fewer waves), so it loses nothing — and the kernels that are meant to run BESIDE it (the deferred compaction, the owner partition of
// the next batch on the exchange stream: 256-thread workgroups) find wave slots on every CU at once instead of waiting for four of this kernel's one-wave
// workgroups to retire on the same CU (kernel trace of the sharded rehearsal: k_part_count 58-72 us and the deferred k_compact_winners 75 us beside the full-occupancy
// kernel, i.e. they finished when it did).
template <bool AOS, int MODE, bool UNIQUE, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 6))) void k_probe_apply_w6(MergeArgs A) { probe_apply_body<AOS, MODE, UNIQUE, NT>(A); }
template <bool AOS, int MODE, bool UNIQUE, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 5))) void k_probe_apply_w5(MergeArgs A) { probe_apply_body<AOS, MODE, UNIQUE, NT>(A); }
template <bool AOS, int MODE, bool UNIQUE, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 4))) void k_probe_apply_w4(MergeArgs A) { probe_apply_body<AOS, MODE, UNIQUE, NT>(A); }
template <bool AOS, int MODE, bool UNIQUE, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 3))) void k_probe_apply_w3(MergeArgs A) { probe_apply_body<AOS, MODE, UNIQUE, NT>(A); }

// Pending pass (duplicate keys only): one lane per delta, so every walker starts at once (the pass is latency bound:
// few walkers, each a chain of dependent reads). Only the FIRST claimer of a row that got followers does anything:
// it walks next[] forward to the last claimer and applies the sequential outcome for that key.

// top-2 tracker: best (ts,val) with its smallest index, and the best among the rest
struct Top2 {
  int64_t t1 = INT64_MIN, v1 = INT64_MIN, t2 = INT64_MIN, v2 = INT64_MIN;
  uint32_t o1 = ~0u, o2 = ~0u;
  __device__ __forceinline__ void add(uint32_t idx, int64_t t, int64_t v) {
    int c1 = o1 == ~0u ? 1 : lexcmp(t, v, t1, v1);
    if (c1 > 0 || (c1 == 0 && idx < o1)) {
      if (o1 != ~0u) { int c2 = o2 == ~0u ? 1 : lexcmp(t1, v1, t2, v2); if (c2 > 0 || (c2 == 0 && o1 < o2)) { t2 = t1; v2 = v1; o2 = o1; } }
      t1 = t; v1 = v; o1 = idx;
    } else {
      int c2 = o2 == ~0u ? 1 : lexcmp(t, v, t2, v2);
      if (c2 > 0 || (c2 == 0 && idx < o2)) { t2 = t; v2 = v; o2 = idx; }
    }
  }
  // best entry whose index is not `skip`
  __device__ __forceinline__ bool best_except(uint32_t skip, int64_t& t, int64_t& v, uint32_t& o) const {
    const bool use2 = o1 == skip;
    o = use2 ? o2 : o1;
    if (o == ~0u) return false;
    t = use2 ? t2 : t1; v = use2 ? v2 : v1;
    return true;
  }
};

// Single-pass walk of a row's list by its FIRST claimer: tracks the smallest index (it creates an absent row, stored with ts := 2),
// the best (ts,val) with its smallest index and the best among the other deltas, so that the creating delta can be excluded
// afterwards without a second walk. One dependent L2 round trip per list node (the node's link and its (ts,val) are loaded together).
template <bool AOS, int MODE>
__device__ __forceinline__ void resolve_one(const MergeArgs& A, const uint32_t j) {
  if (j >= A.n) return;
  const uint8_t wf = A.wflag[j];
  if (!(wf & W_FIRST)) return;                      // followers do nothing here
  uint32_t nx = A.next[j];
  if ((nx >> IDX_BITS) != A.epoch) return;          // nobody followed: the row already holds the outcome
  // independent loads: this delta's value and the row (one line)
  int64_t t, v; load_delta_tv<AOS>(A, j, t, v);
  Slot* sl = A.slots + A.slot_of[j];
  const uint4 hi = reinterpret_cast<const uint4*>(sl)[1];
  const int64_t tsw = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
  const int64_t cval = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
  const bool is_new = tsw == TS_NEW || ts_mark(tsw) == A.epoch;  // row created in this batch: no pre-batch state
  const uint32_t base_owner = (wf & W_WINNER) ? j : ~0u;   // the first claimer stored iff it beat the pre-batch row
  int64_t bt, bv;
  uint32_t owner, created_by = ~0u;     // created_by: the delta whose write creates the row (the smallest index of the list)
  if (!is_new) {
    // Resident row (every hot key of a streaming replay): lexmax over the list with ties to the smaller index, starting from what the row
    // holds now — the pre-batch value, or the first claimer's if it won. A dozen instructions per hop: with one active lane per wave the walk
    // is paced by instruction issue as much as by the dependent load, so the general tracker below (second-best, smallest index) is kept
    // for rows created in this batch only.
    bt = ts_value(tsw); bv = cval; owner = base_owner;
    uint32_t idx = j, steps = 0;
    for (;;) {
      if (idx != j) {       // the first claimer's own value is what the row already holds (or it lost against the pre-batch row)
        const bool gt = t > bt || (t == bt && v > bv);
        const bool eq = t == bt && v == bv;
        if (gt) { bt = t; bv = v; owner = idx; }
        else if (eq && owner != ~0u && idx < owner) owner = idx;
      }
      if ((nx >> IDX_BITS) != A.epoch) break;
      idx = nx & IDX_MASK;
      nx = A.next[idx]; load_delta_tv<AOS>(A, idx, t, v);   // the next node's link and value: independent loads, one round trip
      if (++steps > A.n) { atomicOr(A.status, ST_SPIN); return; }
    }
  } else {
    uint32_t j0 = j;
    int64_t v_j0 = v, t_j0 = t;
    Top2 top;
    {
      uint32_t idx = j, steps = 0;
      for (;;) {
        if (idx <= j0) { j0 = idx; t_j0 = t; v_j0 = v; }
        top.add(idx, t, v);
        if ((nx >> IDX_BITS) != A.epoch) break;
        idx = nx & IDX_MASK;
        nx = A.next[idx]; load_delta_tv<AOS>(A, idx, t, v);
        if (++steps > A.n) { atomicOr(A.status, ST_SPIN); return; }
      }
    }
    // the row starts as (2 or t_j0, v_j0) owned by j0 (src/bullet-crt.js:172-185); the other deltas then compete against it
    bt = (MODE == BMX_INSERT_REFERENCE) ? 2 : t_j0; bv = v_j0; owner = j0; created_by = j0;
    int64_t tm, vm; uint32_t om;
    if (top.best_except(j0, tm, vm, om) && lexcmp(tm, vm, bt, bv) > 0) { bt = tm; bv = vm; owner = om; }   // a tie keeps j0 (it has the smaller index)
  }
  store_tv(sl, is_new ? (bt | ((int64_t)A.epoch << TS_MARK_SHIFT)) : bt, bv);
  // move the winner mark (and the per-block winner counts the compaction relies on) from the first claimer to the owner
  if (base_owner != owner) {
    if (base_owner != ~0u) { A.wflag[base_owner] = W_NONE; atomicSub(&A.blk_info[base_owner >> 8], 1u); }
    if (owner != ~0u) {
      A.wflag[owner] = is_new ? (uint8_t)(W_WINNER | W_CREATED | (owner == created_by ? W_FIRSTWRITE : 0)) : W_WINNER; atomicAdd(&A.blk_info[owner >> 8], 1u);
      A.slot_of[owner] = A.slot_of[j];    // the compaction's index change log names the winner's row
    }
  } else if (is_new && owner != ~0u) {
    A.wflag[owner] = (uint8_t)(W_FIRST | W_WINNER | W_CREATED | (owner == created_by ? W_FIRSTWRITE : 0));   // the first claimer stays the owner, but is it the creating delta?
  }
  if (owner != ~0u && A.flags) A.flags[owner] = (uint8_t)BMX_FLAG_INCOMING;
}

// One workgroup per 256-delta block: every walker of the batch starts at once. (Handling several blocks per workgroup to dispatch
// fewer workgroups when nothing is flagged was measured: no gain — the 4-5 us this launch costs on a unique-key batch are the kernel
// boundary behind k_probe_apply, which leaves ~27 MB of dirty lines to write back, not the dispatch.)
template <bool AOS, int MODE>
__global__ __launch_bounds__(256) void k_resolve_lists(MergeArgs A) {
  // 256-delta blocks none of whose deltas got a follower return after one load
  if (A.blk_follow[blockIdx.x] == A.epoch) resolve_one<AOS, MODE>(A, blockIdx.x * 256u + threadIdx.x);
  // Deferred compaction: this launch ends only once the compaction of the batch BEFORE this one is done (it was released when this batch's
  // probe kernel started and takes a tenth of that kernel's time, so the word is there already): the next probe kernel may then overwrite
  // the workspace half that compaction read. One lane of one block polls; every other block is gone, nothing can starve the compaction.
  if (A.k3_done && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();            // 100 MHz
    while (__hip_atomic_load(A.k3_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.k3_wait) {
      __builtin_amdgcn_s_sleep(8);
      if (wait_gave_up(t0, A.status)) { atomicOr(A.status, ST_SPIN); break; }   // ~60 s (g_wait_ticks): report instead of hanging
    }
  }
  if (A.tail_n && blockIdx.x == 0 && threadIdx.x < A.tail_n) {     // lane k polls word k (relaxed, cache-bypassing: the acquire is the boundary behind this launch)
    const unsigned long long t0 = wall_clock64();
    unsigned long long seen;
    while ((seen = __hip_atomic_load(A.tail_words + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < A.tail_at_least) {
      __builtin_amdgcn_s_sleep(8);
      if (wait_gave_up(t0, A.status)) {
        if (A.tail_diag && !A.tail_diag[0]) { A.tail_diag[0] = (unsigned long long)(uintptr_t)(A.tail_words + threadIdx.x); A.tail_diag[1] = A.tail_at_least; A.tail_diag[2] = seen; }
        atomicOr(A.status, ST_SPIN);
        break;
      }
    }
  }
}

// ---- strict mode (BMX_MERGE_STRICT_FLAGS): exact sequential per-delta flags for any batch ----
// k_probe_link_strict: every valid delta finds/creates its row, claims it and links into the row's list; nobody compares,
// drops or writes. k_resolve_strict: every delta walks its row's whole list once. Because the state after applying any SET
// of deltas in index order is order independent — lexmax of the resident row and the deltas, an absent row starting as
// (2, value of its smallest-index delta) — the state delta j meets is computed from the members with a smaller index,
// which gives resolve()'s flags for j exactly (src/bullet-crt.js:164-279). The last claimer also writes the final state.
constexpr uint32_t STRICT_NO_ROW = 0xFFFFFFFFu;

template <bool AOS>
__global__ __launch_bounds__(256) void k_probe_link_strict(MergeArgs A) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const bool active = j < A.n;
  uint64_t id = EMPTY_ID; uint32_t field = 0; int64_t a = 0, v = 0;
  if (active) load_delta<AOS>(A, j, id, field, a, v);
  const bool pad = AOS && id == EMPTY_ID;
  const bool valid = active && id != EMPTY_ID && field != FIELD_PENDING && a >= 0 && a <= TS_MAX && ((v >= -VAL_MAX && v <= VAL_MAX) || (A.force && v == VAL_DELETED));
  if (active && !valid && !pad) atomicOr(A.status, ST_RANGE);
  uint32_t slot = STRICT_NO_ROW;
  bool conflict = false, created = false;
  if (valid) {
    bool is_new; int64_t cts, cval; uint64_t s; uint32_t prev = 0;
    const uint32_t tag = (A.epoch << IDX_BITS) | j;
    if (probe_or_insert<false>(A, tag, id, field, s, is_new, created, prev, cts, cval)) {
      if (!created) prev = atomicExch(&(A.slots + s)->head, tag);
      if ((prev >> IDX_BITS) == A.epoch) { A.next[j] = (A.epoch << IDX_BITS) | (prev & IDX_MASK); conflict = true; }
      slot = (uint32_t)s;
    }
  }
  {
    unsigned long long mc = __ballot(created), mx = __ballot(conflict);
    if (lane_id() == 0 && (mc | mx)) {
      unsigned long long* ctr = A.shard_ctr + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & (CTR_SHARDS - 1)) * CTR_STRIDE;
      if (mc) atomicAdd(ctr + 0, (unsigned long long)__popcll(mc));
      if (mx) atomicAdd(ctr + 1, (unsigned long long)__popcll(mx));
    }
  }
  if (active) {
    A.slot_of[j] = slot;            // STRICT_NO_ROW for deltas that take no part (invalid, padding)
    A.wflag[j] = W_NONE;            // only the last claimer of a row ever writes W_WINNER (in k_resolve_strict)
    if (A.flags) A.flags[j] = 0;
  }
  if (threadIdx.x == 0) A.blk_info[blockIdx.x] = 0;   // winners are counted by k_resolve_strict
}

// APPLY = false: every delta computes its own sequential flags (rows are only READ: they still hold the pre-batch state).
// APPLY = true : launched afterwards; only the last claimer of each row proceeds, writes the final state, names the winner.
template <bool AOS, int MODE, bool APPLY>
__global__ __launch_bounds__(256) void k_resolve_strict(MergeArgs A) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= A.n) return;
  const uint32_t slot = A.slot_of[j];
  if (slot == STRICT_NO_ROW) return;
  Slot* sl = A.slots + slot;
  const uint4* q = reinterpret_cast<const uint4*>(sl);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t head = lo.w & IDX_MASK;
  if (APPLY && head != j) return;
  int64_t aj, vj; load_delta_tv<AOS>(A, j, aj, vj);
  const int64_t tsw = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
  const int64_t rval = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
  const bool is_new = tsw == TS_NEW;            // unwritten = absent before the batch (strict mode writes rows only in the APPLY launch)
  // one walk over the whole list, from the last claimer to the first
  uint32_t j0 = ~0u; int64_t t_j0 = 0, v_j0 = 0;   // smallest index of the list and its delta
  Top2 top;                                       // APPLY: over all members; else: over the members with index < j
  {
    uint32_t idx = head, steps = 0;
    for (;;) {
      int64_t t, v; load_delta_tv<AOS>(A, idx, t, v);
      uint32_t nx = A.next[idx];
      if (idx < j0) { j0 = idx; t_j0 = t; v_j0 = v; }
      if (APPLY || idx < j) top.add(idx, t, v);
      if ((nx >> IDX_BITS) != A.epoch) break;
      idx = nx & IDX_MASK;
      if (++steps > A.n) { atomicOr(A.status, ST_SPIN); return; }
    }
  }
  // state that (APPLY) the whole list leaves / (else) delta j meets: the resident row or the row created by j0, raised by `top`
  int64_t bt, bv; uint32_t owner;
  if (is_new) {
    bt = (MODE == BMX_INSERT_REFERENCE) ? 2 : t_j0; bv = v_j0; owner = j0;
    int64_t tm, vm; uint32_t om;
    if (top.best_except(j0, tm, vm, om) && lexcmp(tm, vm, bt, bv) > 0) { bt = tm; bv = vm; owner = om; }   // a tie keeps j0 (smaller index)
  } else {
    bt = ts_value(tsw); bv = rval; owner = ~0u;
    if (top.o1 != ~0u && lexcmp(top.t1, top.v1, bt, bv) > 0) { bt = top.t1; bv = top.v1; owner = top.o1; }
  }
  if (APPLY) {
    if (owner != ~0u) {
      store_tv(sl, is_new ? (bt | ((int64_t)A.epoch << TS_MARK_SHIFT)) : bt, bv);
      A.wflag[owner] = (uint8_t)(W_WINNER | (is_new ? W_CREATED | (owner == j0 ? W_FIRSTWRITE : 0) : 0));
      atomicAdd(&A.blk_info[owner >> 8], 1u);
    }
  } else if (A.flags) {
    uint32_t fl;
    if (is_new && j == j0) {
      fl = BMX_FLAG_INCOMING;                       // "no current state": src/bullet-crt.js:172-185
    } else {
      int c = lexcmp(aj, vj, bt, bv);
      fl = c > 0 ? BMX_FLAG_INCOMING : (c == 0 ? 0u : (BMX_FLAG_CURRENT | (aj < bt ? BMX_FLAG_HISTORICAL : 0u)));
    }
    A.flags[j] = (uint8_t)fl;
  }
}

// Self-check of the ONE hardware assumption k_probe_apply's exactness rests on: an aligned 16-byte load never observes half of an
// aligned 16-byte store to the same address. Even workgroups store (x, ~x ^ K) pairs into the (ts,val) half of random slots of a small table,
// odd ones load them — alternately with the plain loads the probe uses and with loads that go to L2 every time — and count pairs that do not
// belong together. SPLIT writes the halves with two 8-byte stores instead: the control, which must show torn pairs for the check to mean anything.
// Run once per device and process by bmx_create (bmx_selfcheck); a few milliseconds.
constexpr uint64_t TEAR_K = 0x5DEECE66DA5A5A5Aull;
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_selfcheck_tear(uint4* slots, uint32_t nslots, uint32_t iters, unsigned long long* torn, unsigned long long* reads) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const bool writer = (blockIdx.x & 1u) == 0;
  uint64_t x = (uint64_t)gid * 0x9E3779B97F4A7C15ull + 1;
  unsigned long long bad = 0, n = 0;
  for (uint32_t i = 0; i < iters; i++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t s = (uint32_t)(x >> 40) % nslots;
    uint4* p = slots + 2 * (size_t)s + 1;                       // second half of a 32-byte slot: where (ts,val) lives
    if (writer) {
      const uint64_t a = x, b = ~x ^ TEAR_K;
      if (SPLIT) { reinterpret_cast<volatile uint64_t*>(p)[0] = a; reinterpret_cast<volatile uint64_t*>(p)[1] = b; }
      else *p = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    } else {
      u32x4 v;
      if (blockIdx.x & 2u) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
      else { v = *reinterpret_cast<const u32x4*>(p); asm volatile("" ::: "memory"); }     // the probe's own form: one plain global_load_dwordx4
      const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32), b = (uint64_t)v.z | ((uint64_t)v.w << 32);
      if (!(a == 0 && b == 0) && b != (~a ^ TEAR_K)) bad++;
      n++;
    }
  }
  if (bad) atomicAdd(torn, bad);
  if (n) atomicAdd(reads, n);
}

// Placement probe (bmx_create): the probe kernel's request mix — one random 32-byte slot read, one atomic exchange on its head word, one 16-byte store of its
// (ts,val) half — over an UNINITIALISED table allocation, n threads. Used to rank candidate allocations of the same size (see tune_table_placement in bmx.hip);
// k_init_slots runs afterwards, so what it scribbles does not matter.
__global__ __launch_bounds__(64) void k_placement_probe(Slot* slots, uint64_t nslots, uint32_t n, uint32_t salt) {
  const uint32_t j = blockIdx.x * 64u + threadIdx.x;
  if (j >= n) return;
  const uint64_t h = mix64(((uint64_t)salt << 32) | j);
  Slot* sl = slots + __umul64hi(h, nslots);
  const uint4* q = reinterpret_cast<const uint4*>(sl);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t prev = atomicExch(&sl->head, j);
  if (((lo.x ^ hi.x ^ prev) & 7u) != 5u)     // (data dependent: the loads and the exchange cannot be dropped; true for seven slots in eight)
    reinterpret_cast<uint4*>(sl)[1] = make_uint4(j, lo.y, hi.z, prev);
}

// epoch wrap: forget every claim tag and every creation mark
__global__ __launch_bounds__(256) void k_sweep_heads(Slot* slots, uint64_t nslots) {
  for (uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * 256u) {
    slots[s].head = 0;
    int64_t t = slots[s].ts;
    if (t != TS_NEW && ts_mark(t)) slots[s].ts = ts_value(t);
  }
}

__global__ __launch_bounds__(256) void k_init_slots(Slot* slots, uint64_t nslots) {
  const uint4 lo = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, FIELD_PENDING, 0u);
  const uint4 hi = make_uint4(0u, 0x80000000u, 0u, 0u);  // ts = INT64_MIN, val = 0
  for (uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * 256u) {
    uint4* q = reinterpret_cast<uint4*>(slots + s);
    q[0] = lo; q[1] = hi;
  }
}

// growth: re-insert every row of the old table into the new (empty) one. Keys are unique, so a lane only competes
// with lanes of OTHER keys for an empty slot: CAS on id, then plain stores of the rest (nobody reads it in this launch).
__global__ __launch_bounds__(256) void k_rehash(const Slot* old_slots, uint64_t old_n, Slot* slots, uint64_t nslots, uint32_t* status) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < old_n; i += (uint64_t)gridDim.x * 256u) {
    const uint4* q = reinterpret_cast<const uint4*>(old_slots + i);
    uint4 lo = q[0];
    uint64_t id = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    if (id == EMPTY_ID) continue;
    uint4 hi = q[1];
    int64_t t = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
    if (t != TS_NEW) t = ts_value(t);          // creation marks do not survive a rehash
    ProbeSeq<4> ps(id, lo.z, nslots);
    bool done = false;
    for (uint64_t p = 0; p < nslots && !done; ++p) {
      Slot* sl = slots + ps.slot();
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&sl->id), (unsigned long long)EMPTY_ID, (unsigned long long)id);
      if (old == EMPTY_ID) {
        sl->field = lo.z; sl->head = 0;
        reinterpret_cast<uint4*>(sl)[1] = make_uint4((uint32_t)(uint64_t)t, (uint32_t)((uint64_t)t >> 32), hi.z, hi.w);
        done = true;
      } else {
        ps.next();
      }
    }
    if (!done) atomicOr(status, ST_FULL);
  }
}

// read-only lookup of n keys
__global__ __launch_bounds__(256) void k_get_rows(const Slot* slots, uint64_t nslots, uint32_t n, const uint64_t* id,
                                                  const uint32_t* field, int64_t* ts, int64_t* val, uint8_t* found) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  uint64_t kid = id[j]; uint32_t kf = field[j];
  ProbeSeq<4> ps(kid, kf, nslots);
  uint8_t f = 0; int64_t t = 0, v = 0;
  for (uint64_t p = 0; p < nslots; ++p) {
    const uint4* q = reinterpret_cast<const uint4*>(slots + ps.slot());
    uint4 lo = q[0];
    uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    if (sid == EMPTY_ID) break;
    if (sid == kid && lo.z == kf) {
      uint4 hi = q[1];
      t = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32));
      v = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
      f = t != TS_NEW;
      t = ts_value(t);
      break;
    }
    ps.next();
  }
  ts[j] = t; val[j] = v; found[j] = f;
}

// Cross-stream sequencing through a word in device memory (bmx_seq_signal / bmx_seq_wait): one wave each.
__global__ void k_seq_signal(unsigned long long* seq, unsigned long long value) {
  if (threadIdx.x == 0) __hip_atomic_store(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// `diag` (3 words, optional): on expiry {address of the word, value waited for, value last seen} so that the host can say WHICH
// hand-off never came (a peer that died or fell behind shows up as "waited for 17, saw 16", not as an anonymous timeout)
__global__ void k_seq_wait(const unsigned long long* seq, unsigned long long at_least, uint32_t* status, unsigned long long* diag) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();            // 100 MHz
  unsigned long long seen;
  while ((seen = __hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < at_least) {
    __builtin_amdgcn_s_sleep(8);
    if (wait_gave_up(t0, status)) {                        // ~60 s (g_wait_ticks), or an earlier wait of this context failed already: report instead of hanging
      if (diag && !diag[0]) { diag[0] = (unsigned long long)(uintptr_t)seq; diag[1] = at_least; diag[2] = seen; }
      atomicOr(status, ST_SPIN);
      return;
    }
  }
}

// Wait until EVERY one of n words (this GPU's memory, written by peers: system-scope loads) has reached `at_least`: lane k polls word k.
__global__ void k_seq_wait_all(const unsigned long long* words, uint32_t n, unsigned long long at_least, uint32_t* status, unsigned long long* diag) {
  const uint32_t k = threadIdx.x;
  if (k >= n) return;
  const unsigned long long t0 = wall_clock64();            // 100 MHz
  unsigned long long seen;
  // relaxed loads that bypass the caches: the ACQUIRE is the kernel boundary behind this launch (an acquire fence at system scope here would
  // invalidate the L2 under everything else that runs; measured: +100 us per step with fences inside the kernels)
  while ((seen = __hip_atomic_load(words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < at_least) {
    __builtin_amdgcn_s_sleep(8);
    if (wait_gave_up(t0, status)) {                        // ~60 s (g_wait_ticks), or an earlier wait of this context failed already
      if (diag && !diag[0]) { diag[0] = (unsigned long long)(uintptr_t)(words + k); diag[1] = at_least; diag[2] = seen; }
      atomicOr(status, ST_SPIN);
      return;
    }
  }
}
// One store of `value` into each of n words that may live in other GPUs' memory (lane k -> word k), after everything enqueued before.
// The RELEASE is the kernel boundary in front of this launch: everything enqueued before it on the stream is complete and written back.
__global__ void k_seq_signal_multi(SeqPtrs w, uint32_t n, unsigned long long value) {
  if (threadIdx.x < n && w.p[threadIdx.x]) __hip_atomic_store(w.p[threadIdx.x], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// K7: stable partition of a delta batch by owner shard into 32-byte records (two launches: count, scatter).
// Counting and ranking are wave-ballot based (one __ballot per shard per 64 deltas): no LDS or global atomics.

constexpr int PART_BLOCKS = 1024;
constexpr uint32_t PART_TILE = 1024;   // deltas staged in LDS per step of k_part_scatter; per_block is a multiple of it
// Where the records of each shard go. Default: one buffer, shard g's run (or slab) at its offset. With `split` set, shard g's slab
// starts at base[g] instead — a pointer into the RECEIVE slabs of the shard that owns g, possibly peer-mapped memory of another GPU:
// the owner partition then scatters straight into its peers' receive buffers and no copy kernel runs afterwards (bmx_comm_*).
struct PartOut {
  bmx_delta_rec* base[PART_MAX_SHARDS];
  uint32_t split;
  uint32_t aux_base;     // added to the origin index carried in `aux` (offset of this originator's slice in a global batch)
  // round 5: the wait for the destination slabs to be free again (bmx_partition_scatter's wait words) folded into the scatter pass itself — every workgroup polls
  // the (<= 16) words before its first store — instead of a one-wave launch of its own in front of the count pass: that launch was 5-6 us of the exchange
  // stream's 90-us chain per step (profiles/r04_sharded_timeline.log), and the count pass, which writes nothing into the slabs, had to wait behind it
  const unsigned long long* wait_words = nullptr; uint32_t n_wait = 0; unsigned long long wait_at_least = 0; unsigned long long* wait_diag = nullptr;
};

__device__ __forceinline__ uint32_t owner_of_dev(uint64_t id, uint32_t nshards) { return (uint32_t)__umul64hi(owner_hash(id), (uint64_t)nshards); }

__global__ __launch_bounds__(256) void k_part_count(const uint64_t* id, uint32_t n, uint32_t nshards, uint32_t per_block,
                                                    uint32_t* counts /*[nshards][PART_BLOCKS]*/, uint8_t* owner_out /*[n]*/) {
  __shared__ uint32_t wtot[4][PART_MAX_SHARDS];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t lo = blockIdx.x * per_block, hi = min(n, lo + per_block);
  uint32_t mine = 0;  // lane g (< nshards) of each wave accumulates shard g's count
  for (uint32_t t0 = lo; t0 < hi; t0 += 256) {
    uint32_t j = t0 + threadIdx.x;
    uint32_t g = j < hi ? owner_of_dev(id[j], nshards) : 0xFFFFFFFFu;
    if (j < hi) owner_out[j] = (uint8_t)g;          // the scatter pass reads this byte instead of hashing again
    for (uint32_t gg = 0; gg < nshards; gg++) {
      uint32_t c = (uint32_t)__popcll(__ballot(g == gg));
      if (lane == gg) mine += c;
    }
  }
  if (lane < PART_MAX_SHARDS) wtot[w][lane] = mine;
  __syncthreads();
  if (threadIdx.x < nshards) counts[threadIdx.x * PART_BLOCKS + blockIdx.x] = wtot[0][threadIdx.x] + wtot[1][threadIdx.x] + wtot[2][threadIdx.x] + wtot[3][threadIdx.x];
}

// The scatter pass is VALU-issue bound, not bandwidth bound (every block is resident at once, 4 waves per SIMD; PMC: 1900 VALU
// instructions per wave in the first version), so it is written to execute few instructions: owners come as bytes from the count
// pass, ranks from packed 8-bit one-hot counters scanned with DPP (4 shards per 32-bit word), all prefix tables stay in LDS and
// are read with computed addresses, and the copy-out walks one shard's run at a time.
__global__ __launch_bounds__(256) void k_part_scatter(const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                                                      const uint8_t* owner, uint32_t n, uint32_t nshards, uint32_t per_block, const uint32_t* counts,
                                                      bmx_delta_rec* out, unsigned long long* totals, uint32_t slab, uint32_t* status, PartOut po) {
  __shared__ uint32_t base[PART_MAX_SHARDS];       // running output cursor of this block per shard
  __shared__ uint32_t tot[PART_MAX_SHARDS];        // shard totals over the whole batch
  __shared__ uint32_t red[PART_MAX_SHARDS][2];
  __shared__ uint4 stage[PART_TILE * 2];           // 32 KB: one tile of records grouped by shard
  __shared__ __attribute__((aligned(16))) uint32_t F[PART_MAX_SHARDS * 16 + 4];   // F[g*16+c]: count of shard g in sub-chunk c, then the
                                                                                   // exclusive prefix of the flattened table; F[256] = records in the tile
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t lo = blockIdx.x * per_block, hi = min(n, lo + per_block);
  uint64_t kid[4], rt[4], rv[4]; uint32_t rf[4], g[4];
  auto load_tile = [&](uint32_t t0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t j = t0 + (uint32_t)i * 256u + threadIdx.x;
      const bool act = j < hi;
      kid[i] = act ? id[j] : EMPTY_ID; rf[i] = act ? field[j] : 0u; rt[i] = act ? (uint64_t)ts[j] : 0ull; rv[i] = act ? (uint64_t)val[j] : 0ull;
      g[i] = act ? (uint32_t)owner[j] : 0xFFu;
    }
  };
  if (lo < hi) load_tile(lo);                      // in flight under the prologue
  if (po.n_wait) {                                   // the slabs this launch stores into are free once every word has reached wait_at_least (lane k polls word k)
    __shared__ uint32_t gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    if (threadIdx.x < po.n_wait) {
      const unsigned long long t0 = wall_clock64();
      unsigned long long seen;
      while ((seen = __hip_atomic_load(po.wait_words + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < po.wait_at_least) {
        __builtin_amdgcn_s_sleep(8);
        if (wait_gave_up(t0, status)) {
          if (po.wait_diag && !po.wait_diag[0]) { po.wait_diag[0] = (unsigned long long)(uintptr_t)(po.wait_words + threadIdx.x); po.wait_diag[1] = po.wait_at_least; po.wait_diag[2] = seen; }
          atomicOr(status, ST_SPIN); gave_up = 1u;
          break;
        }
      }
    }
    __syncthreads();
    if (gave_up) return;                             // (uniform) never store into slabs somebody may still read: the sticky error says the batch was not routed
  }
  // prologue: per shard, before = sum of counts[g][b] over the blocks before this one, all = sum over every block.
  // Wave w reduces shards w, w+4, ...; a lane reads 4 consecutive blocks per 16-byte load.
  for (uint32_t gs = w; gs < nshards; gs += 4) {
    const uint4* cg = reinterpret_cast<const uint4*>(counts + gs * PART_BLOCKS);
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < PART_BLOCKS / 256; k++) {
      const uint32_t b4 = k * 64 + lane;
      const uint4 c = cg[b4];
      const uint32_t b0 = b4 * 4;
      all += c.x + c.y + c.z + c.w;
      before += (b0 < blockIdx.x ? c.x : 0u) + (b0 + 1 < blockIdx.x ? c.y : 0u) + (b0 + 2 < blockIdx.x ? c.z : 0u) + (b0 + 3 < blockIdx.x ? c.w : 0u);
    }
    before = wave_incl_scan_u32(before); all = wave_incl_scan_u32(all);
    if (lane == 63) { red[gs][0] = before; red[gs][1] = all; }
  }
  __syncthreads();
  if (threadIdx.x < nshards) {
    tot[threadIdx.x] = red[threadIdx.x][1];
    if (blockIdx.x == 0) {
      totals[threadIdx.x] = red[threadIdx.x][1];
      if (slab && red[threadIdx.x][1] > slab) atomicOr(status, ST_SLAB);   // records were dropped: sticky, reported by bmx_sync
    }
  }
  __syncthreads();
  if (threadIdx.x < nshards) {
    uint32_t gg = threadIdx.x, start = 0;
    if (slab) start = gg * slab;                             // fixed-size slabs: shard g starts at g*slab
    else for (uint32_t x = 0; x < gg; x++) start += tot[x];
    base[gg] = start + red[gg][0];
  }
  // (the barriers inside the loop order base[] before its first use)
  const uint32_t nwords = (nshards + 3) >> 2;
  for (uint32_t t0 = lo; t0 < hi; t0 += PART_TILE) {
    if (t0 != lo) load_tile(t0);
    // rank of every delta among the deltas of its shard inside its 64-delta sub-chunk (index order: sub-chunk c = i*4 + w)
    uint32_t rk[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      rk[i] = 0;
      const uint32_t c = (uint32_t)i * 4u + w;
      for (uint32_t k = 0; k < nwords; k++) {
        const uint32_t sh = (g[i] & 3u) * 8u;
        const bool mine = (g[i] >> 2) == k;
        const uint32_t incl = wave_incl_scan_u32(mine ? (1u << sh) : 0u);
        if (mine) rk[i] = ((incl >> sh) & 0xFFu) - 1u;
        const uint32_t totw = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);     // at most 64 per field: no carry between fields
        if (lane < 4 && 4 * k + lane < nshards) F[(4 * k + lane) * 16 + c] = (totw >> (8u * lane)) & 0xFFu;
      }
    }
    __syncthreads();
    if (w == 0) {   // exclusive prefix of the flattened table (shard-major, sub-chunk-minor) = tile-local position of every run
      const bool live = lane * 4 < nshards * 16;
      uint4 x = live ? reinterpret_cast<const uint4*>(F)[lane] : make_uint4(0u, 0u, 0u, 0u);
      const uint32_t sum = x.x + x.y + x.z + x.w;
      const uint32_t incl = wave_incl_scan_u32(sum);
      const uint32_t e = incl - sum;
      reinterpret_cast<uint4*>(F)[lane] = make_uint4(e, e + x.x, e + x.x + x.y, e + x.x + x.y + x.z);
      if (lane == 63) F[PART_MAX_SHARDS * 16] = incl;
    }
    __syncthreads();
    uint32_t grow = 0;                             // this tile's contribution to the block's cursor of shard threadIdx.x
    if (threadIdx.x < nshards) grow = F[(threadIdx.x + 1) * 16] - F[threadIdx.x * 16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (g[i] != 0xFFu) {
        const uint32_t p = F[g[i] * 16 + (uint32_t)i * 4u + w] + rk[i];
        stage[2 * p] = make_uint4((uint32_t)kid[i], (uint32_t)(kid[i] >> 32), rf[i], po.aux_base + t0 + (uint32_t)i * 256u + threadIdx.x);
        stage[2 * p + 1] = make_uint4((uint32_t)rt[i], (uint32_t)(rt[i] >> 32), (uint32_t)rv[i], (uint32_t)(rv[i] >> 32));
      }
    }
    __syncthreads();
    for (uint32_t gg = 0; gg < nshards; gg++) {    // one shard's run at a time: consecutive lanes on consecutive 16-byte halves
      const uint32_t s0 = F[gg * 16], e0 = F[(gg + 1) * 16], b = base[gg];
      uint4* out16 = po.split ? reinterpret_cast<uint4*>(po.base[gg]) - 2 * (size_t)gg * slab : reinterpret_cast<uint4*>(out);
      for (uint32_t q = 2 * s0 + threadIdx.x; q < 2 * e0; q += 256) {
        const uint32_t pos = b + ((q >> 1) - s0);
        if (!slab || pos - gg * slab < slab)        // a slab overflow drops the record; totals[] tells the caller
          out16[2 * (size_t)pos + (q & 1)] = stage[q];
      }
    }
    __syncthreads();
    if (threadIdx.x < nshards) base[threadIdx.x] += grow;
  }
  // fixed-size slabs: the unused tail of every slab becomes padding (reserved id), striped over all blocks
  if (slab) {
    const uint4 padlo = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, FIELD_PENDING, 0u), padhi = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t g = 0; g < nshards; g++) {
      uint32_t used = min(tot[g], slab);
      for (uint32_t p = used + blockIdx.x * 256u + threadIdx.x; p < slab; p += PART_BLOCKS * 256u) {
        uint4* q = reinterpret_cast<uint4*>((po.split ? po.base[g] : out + (size_t)g * slab) + p);
        q[0] = padlo; q[1] = padhi;
      }
    }
  }
}

}  // namespace bmx
