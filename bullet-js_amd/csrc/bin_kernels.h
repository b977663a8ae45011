// bin_kernels.h — the bucketed merge path (opt-in, BMX_MERGE_BUCKETED): bucket the batch, then merge every bucket inside ONE workgroup (gfx950).
//
// Replaces, for scalar clocks {w: ts}, the reference's per-entry loop
//     for (entry of entries) bullet.setData(...) -> crt.handleUpdate -> crt.resolve
// (src/bullet-network-sync.js:551-569, src/bullet-crt.js:329-385, :164-279) with the same final rows and the same winners.
//
// Why (measured, profiles/r02_micro_*.log): on this chip a random row access costs a fixed price per memory-side request,
// whatever it moves — 1M random 128-B line reads 21.5 us, 0.84M dirty 32-B sectors written back 14-15 us, 0.87M global atomics
// 17-20 us, and the three ADD UP. The round-1 kernel paid all three per delta (the atomic claimed the row against other deltas
// of the same key running on other CUs). Here every key of the batch is handled by exactly one workgroup, so duplicates meet in
// LDS and the table sees one line read and at most one sector write-back per KEY and no atomic (except the CAS that creates a row):
//
//   k_bucket      one pass over the batch: a tile of 4096 deltas is regrouped by bin = top bits of the node hash (the same hash
//                 that picks the row's 128-B line, so a bin owns a contiguous range of lines) into a tile-local image + a table
//                 of 1025 offsets. No atomics on global memory, no counting pass, output order inside a tile-bin is arbitrary.
//   k_merge_bins  workgroup b gathers bin b's run from every tile (runs of one tile after the other = index order between
//                 tiles), finds duplicate keys with an LDS hash table, folds each key's deltas into the reference's sequential
//                 outcome (lexmax of (ts,val), ties to the smaller index, first write of an absent key stored with ts := 2),
//                 probes the row once and stores the final (ts,val). A bin larger than the LDS image is processed in several
//                 chunks of whole tile-runs: exact, because the chunks are in index order and a chunk is equivalent to applying
//                 its deltas one by one to the table state the previous chunk left.
//   k_count_winners + k_compact_winners (scan_kernels.h)  ordered compaction of the winner bytes -> applied_idx.
//
// Nothing here waits for another workgroup; the only cross-workgroup operation is the CAS on slot.id that claims an empty slot.
#pragma once
#include "slot.h"
#include "merge_kernels.h"
#include "../../include/bmx.h"

namespace bmx {

constexpr uint32_t NB = 1024;             // bins = workgroups of k_merge_bins
constexpr uint32_t BK_TILE = 4096;        // deltas per bucketing tile
constexpr uint32_t BK_THREADS = 512;      // 8 deltas per thread
constexpr uint32_t BK_RPT = BK_TILE / BK_THREADS;
constexpr uint32_t TOFF_STRIDE = NB + 2;  // u16 offsets per tile (NB + 1 used; even, so that rows stay 4-byte aligned)
constexpr uint32_t MB_THREADS = 512;
constexpr uint32_t MB_CAP = 1536;         // records of one chunk (LDS image); mean bin of a 1M batch: 977, sigma 31
constexpr uint32_t MB_HS = 2048;          // LDS hash entries (power of two)
constexpr uint32_t MB_TILES = 512;        // tile runs examined per chunk (one per thread)
constexpr uint32_t MB_SUB = 4;            // an over-long run of ONE tile is split by index quarter: BK_TILE / MB_SUB <= MB_CAP
constexpr uint32_t L_EMPTY = 0xFFFFFFFFu;
constexpr uint16_t NXT_END = 0xFFFFu, NXT_REP = 0xFFFEu;
static_assert(BK_TILE / MB_SUB <= MB_CAP, "a sub-range of one tile must fit the LDS image");
static_assert(MB_CAP < NXT_REP, "record indices are 16-bit");

struct BinArgs {
  Slot* slots; uint64_t nslots;
  const uint64_t* id; const uint32_t* field; const int64_t* ts; const int64_t* val; const bmx_delta_rec* recs;
  uint32_t n, epoch, ntiles;
  uint4* stage;        // ntiles * BK_TILE records of 32 B: {id lo, id hi, field, index} {ts lo, ts hi, val lo, val hi}, grouped by bin inside each tile
  uint16_t* toff;      // ntiles * TOFF_STRIDE: exclusive offsets of the bins inside their tile (entry NB = valid deltas of the tile)
  uint8_t* wflag;      // per delta: 1 = this delta's value is the row's final value
  uint8_t* flags;      // optional decision flags
  unsigned long long* shard_ctr; uint32_t* status;
};

__device__ __forceinline__ uint32_t bin_of(uint64_t id) { return (uint32_t)__umul64hi(node_hash(id), (uint64_t)NB); }

// ---- K_bucket ----------------------------------------------------------------------------------------------------------
template <bool AOS>
__global__ __launch_bounds__(BK_THREADS) void k_bucket(BinArgs A) {
  __shared__ uint32_t hist[NB];
  __shared__ uint32_t wtot[BK_THREADS / 64];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * BK_TILE;
  for (uint32_t i = tid; i < NB; i += BK_THREADS) hist[i] = 0;
  __syncthreads();
  uint64_t kid[BK_RPT]; uint32_t kf[BK_RPT], bin[BK_RPT], rk[BK_RPT]; int64_t ka[BK_RPT], kv[BK_RPT];
  bool bad = false;
#pragma unroll
  for (uint32_t k = 0; k < BK_RPT; k++) {
    const uint32_t j = base + k * BK_THREADS + tid;
    bin[k] = NB;
    if (j < A.n) {
      if (AOS) {
        const uint4* p = reinterpret_cast<const uint4*>(A.recs + j);
        const uint4 lo = p[0], hi = p[1];
        kid[k] = (uint64_t)lo.x | ((uint64_t)lo.y << 32); kf[k] = lo.z;
        ka[k] = (int64_t)((uint64_t)hi.x | ((uint64_t)hi.y << 32)); kv[k] = (int64_t)((uint64_t)hi.z | ((uint64_t)hi.w << 32));
      } else {
        kid[k] = A.id[j]; kf[k] = A.field[j]; ka[k] = A.ts[j]; kv[k] = A.val[j];
      }
      A.wflag[j] = 0;
      if (A.flags) A.flags[j] = 0;
      const bool pad = AOS && kid[k] == EMPTY_ID;   // padding record of a fixed-size exchange slab
      const bool valid = kid[k] != EMPTY_ID && kf[k] != FIELD_PENDING && ka[k] >= 0 && ka[k] <= TS_MAX && kv[k] >= -VAL_MAX && kv[k] <= VAL_MAX;
      if (!valid && !pad) bad = true;
      if (valid) { bin[k] = bin_of(kid[k]); rk[k] = atomicAdd(&hist[bin[k]], 1u); }
    }
  }
  if (bad) atomicOr(A.status, ST_RANGE);
  __syncthreads();
  // exclusive scan of the NB counts (two per thread)
  const uint32_t c0 = hist[2 * tid], c1 = hist[2 * tid + 1], s = c0 + c1;
  const uint32_t incl = wave_incl_scan_u32(s);
  if ((tid & 63u) == 63u) wtot[tid >> 6] = incl;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (uint32_t i = 0; i < BK_THREADS / 64; i++) { const uint32_t x = wtot[i]; if (i < (tid >> 6)) woff += x; tot += x; }
  const uint32_t e0 = woff + incl - s, e1 = e0 + c0;
  hist[2 * tid] = e0; hist[2 * tid + 1] = e1;
  uint32_t* trow = reinterpret_cast<uint32_t*>(A.toff + (size_t)blockIdx.x * TOFF_STRIDE);
  trow[tid] = e0 | (e1 << 16);
  if (tid == 0) trow[NB / 2] = tot;   // entry NB (and the unused NB + 1)
  __syncthreads();
  uint4* st = A.stage + (size_t)base * 2;
#pragma unroll
  for (uint32_t k = 0; k < BK_RPT; k++) {
    if (bin[k] != NB) {
      const uint32_t p = hist[bin[k]] + rk[k];
      st[2 * p] = make_uint4((uint32_t)kid[k], (uint32_t)(kid[k] >> 32), kf[k], base + k * BK_THREADS + tid);
      st[2 * p + 1] = make_uint4((uint32_t)(uint64_t)ka[k], (uint32_t)((uint64_t)ka[k] >> 32), (uint32_t)(uint64_t)kv[k], (uint32_t)((uint64_t)kv[k] >> 32));
    }
  }
}

// ---- K_merge_bins ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 ld16_nt(const void* p) {   // 16-byte load served by L2 (never by this CU's L1)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int64_t i64_of(uint32_t lo, uint32_t hi) { return (int64_t)((uint64_t)lo | ((uint64_t)hi << 32)); }

// block-wide inclusive scan (MB_THREADS threads); total to every thread
__device__ __forceinline__ uint32_t mb_incl_scan(uint32_t v, uint32_t& total, uint32_t* wsum /* [MB_THREADS/64] */) {
  const uint32_t x = wave_incl_scan_u32(v);
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (uint32_t i = 0; i < MB_THREADS / 64; i++) { const uint32_t s = wsum[i]; if (i < (threadIdx.x >> 6)) woff += s; tot += s; }
  __syncthreads();
  total = tot;
  return woff + x;
}

template <int MODE>
__global__ __launch_bounds__(MB_THREADS) void k_merge_bins(BinArgs A) {
  __shared__ uint4 r_lo[MB_CAP], r_hi[MB_CAP];   // chunk image: {id lo, id hi, field, index}, {ts, val}
  __shared__ uint32_t H[MB_HS];                  // key -> representative record
  __shared__ uint32_t lhead[MB_CAP];             // representative -> last linked duplicate
  __shared__ uint16_t nxt[MB_CAP];               // duplicate -> previously linked duplicate; NXT_REP marks a representative
  __shared__ uint32_t s_incl[MB_TILES];          // inclusive prefix of the run lengths of the tiles examined for this chunk
  __shared__ uint16_t s_off[MB_TILES];           // start of each run inside its tile
  __shared__ uint32_t wsum[MB_THREADS / 64];
  __shared__ uint32_t s_m;
  const uint32_t tid = threadIdx.x, b = blockIdx.x;
  uint32_t created = 0, conflicts = 0;
  uint32_t t0 = 0, sub = 0;                      // next tile; next index quarter of an over-long run (sub-range mode)
  while (t0 < A.ntiles) {
    // ---- 1. which tile runs form this chunk ----
    uint32_t len = 0, off = 0;
    const uint32_t tt = t0 + tid;
    if (tid < MB_TILES && tt < A.ntiles) {
      const uint16_t* row = A.toff + (size_t)tt * TOFF_STRIDE + b;
      off = row[0]; len = (uint32_t)row[1] - off;
    }
    uint32_t tot_all;
    const uint32_t incl = mb_incl_scan(len, tot_all, wsum);
    s_incl[tid] = incl; s_off[tid] = (uint16_t)off;
    const uint32_t ntile_here = min(MB_TILES, A.ntiles - t0);
    const uint32_t cut = (uint32_t)__syncthreads_count(tid < ntile_here && incl <= MB_CAP);   // leading runs that fit together (incl is monotone)
    uint32_t m;
    if (cut > 0) {
      m = s_incl[cut - 1];
      // records of the chunk, one per thread and step: find the run by binary search over the prefix
      for (uint32_t i = tid; i < m; i += MB_THREADS) {
        uint32_t lo = 0, hi = cut - 1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_incl[mid] > i) hi = mid; else lo = mid + 1; }
        const uint32_t start = lo ? s_incl[lo - 1] : 0u;
        const uint4* src = A.stage + ((size_t)(t0 + lo) * BK_TILE + s_off[lo] + (i - start)) * 2;
        r_lo[i] = src[0]; r_hi[i] = src[1];
      }
    } else {
      // the first run alone is longer than the image (one tile put > MB_CAP deltas into this bin): take its deltas one index quarter
      // at a time (quarters are disjoint index ranges in ascending order, so the chunks are still in index order)
      const uint32_t run = s_incl[0], o0 = s_off[0];
      const uint32_t qlo = t0 * BK_TILE + sub * (BK_TILE / MB_SUB), qhi = qlo + BK_TILE / MB_SUB;
      uint32_t filled = 0;
      for (uint32_t q0 = 0; q0 < run; q0 += MB_THREADS) {
        const uint32_t q = q0 + tid;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        bool take = false;
        if (q < run) {
          const uint4* src = A.stage + ((size_t)t0 * BK_TILE + o0 + q) * 2;
          a = src[0];
          take = a.w >= qlo && a.w < qhi;
          if (take) c = src[1];
        }
        uint32_t tk;
        const uint32_t pos = filled + mb_incl_scan(take ? 1u : 0u, tk, wsum) - (take ? 1u : 0u);
        if (take) { r_lo[pos] = a; r_hi[pos] = c; }
        filled += tk;
      }
      m = filled;
    }
    // ---- 2. duplicate keys meet in LDS ----
    for (uint32_t i = tid; i < MB_HS; i += MB_THREADS) H[i] = L_EMPTY;
    for (uint32_t i = tid; i < m; i += MB_THREADS) lhead[i] = L_EMPTY;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += MB_THREADS) {
      const uint4 k = r_lo[i];
      uint32_t h = (k.x * 0x9E3779B1u ^ k.y * 0x85EBCA77u ^ k.z * 0xC2B2AE3Du);
      h = (h ^ (h >> 15)) & (MB_HS - 1);
      uint16_t link = NXT_REP;
      for (;;) {
        const uint32_t old = atomicCAS(&H[h], L_EMPTY, i);
        if (old == L_EMPTY) break;                                     // first record of its key in this chunk: representative
        const uint4 ko = r_lo[old];
        if (ko.x == k.x && ko.y == k.y && ko.z == k.z) { link = (uint16_t)atomicExch(&lhead[old], i); break; }   // L_EMPTY truncates to NXT_END
        h = (h + 1) & (MB_HS - 1);
      }
      nxt[i] = link;
    }
    __syncthreads();
    // ---- 3. one thread per key: fold the key's deltas, probe the row once, store the outcome ----
    // Creating rows may need a second look at a slot whose field another thread of THIS workgroup is about to publish (same node id
    // means same bin, so it is never another workgroup's): such keys are retried after a barrier instead of spinning.
    uint32_t pending_mask = 0;   // bit k: my k-th record still has to be processed
    for (uint32_t k = 0; k * MB_THREADS + tid < m; k++) if (nxt[k * MB_THREADS + tid] == NXT_REP) pending_mask |= 1u << k;
    for (uint32_t round = 0;; round++) {
      for (uint32_t k = 0; k * MB_THREADS + tid < m; k++) {
        if (!(pending_mask & (1u << k))) continue;
        const uint32_t i = k * MB_THREADS + tid;
        const uint4 key = r_lo[i];
        const uint64_t id = (uint64_t)key.x | ((uint64_t)key.y << 32);
        const uint32_t field = key.z;
        ProbeSeq<4> ps(id, field, A.nslots);
        uint4 lo = ld16_nt(A.slots + ps.slot()), hi = ld16_nt(reinterpret_cast<const uint4*>(A.slots + ps.slot()) + 1);   // in flight under the fold
        // fold: smallest index (it creates an absent row), best (ts,val) with its smallest index, best of the rest
        uint32_t j0 = key.w, cnt = 1; int64_t t_j0, v_j0;
        Top2 top;
        { const uint4 d = r_hi[i]; t_j0 = i64_of(d.x, d.y); v_j0 = i64_of(d.z, d.w); top.add(key.w, t_j0, v_j0); }
        for (uint32_t p = lhead[i] & 0xFFFFu; p != NXT_END; p = nxt[p]) {
          const uint32_t jx = r_lo[p].w; const uint4 d = r_hi[p];
          const int64_t t = i64_of(d.x, d.y), v = i64_of(d.z, d.w);
          if (jx < j0) { j0 = jx; t_j0 = t; v_j0 = v; }
          top.add(jx, t, v);
          cnt++;
        }
        // probe
        bool found = false, blocked = false, is_new = false, full = true;
        uint64_t s = 0; int64_t cts = 0, cval = 0; uint32_t head = 0;
        for (uint64_t p = 0; p < A.nslots; ++p) {
          s = ps.slot();
          if (p) { lo = ld16_nt(A.slots + s); hi = ld16_nt(reinterpret_cast<const uint4*>(A.slots + s) + 1); }
          uint64_t sid = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
          uint32_t sf = lo.z;
          if (sid == EMPTY_ID) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&A.slots[s].id), (unsigned long long)EMPTY_ID, (unsigned long long)id);
            if (old == EMPTY_ID) { found = true; is_new = true; full = false; break; }
            sid = old; sf = FIELD_PENDING;     // somebody claimed it meanwhile: look at it again below
            if (sid == id) { lo = ld16_nt(A.slots + s); sf = lo.z; hi = ld16_nt(reinterpret_cast<const uint4*>(A.slots + s) + 1); }
          }
          if (sid == id) {
            if (sf == FIELD_PENDING) { blocked = true; full = false; break; }   // a sibling field of this node is being created right now
            if (sf == field) {
              const int64_t t = i64_of(hi.x, hi.y);
              is_new = t == TS_NEW; cts = ts_value(t); cval = i64_of(hi.z, hi.w); head = lo.w;
              found = true; full = false; break;
            }
          }
          ps.next();
        }
        if (blocked) continue;                 // stays pending: retried after the barrier
        pending_mask &= ~(1u << k);
        if (full || !found) { atomicOr(A.status, ST_FULL); continue; }
        conflicts += cnt - 1 + ((!is_new && (head >> IDX_BITS) == A.epoch) ? 1u : 0u);   // the row was already written by an earlier chunk of this batch
        // the reference's sequential outcome for this key (src/bullet-crt.js:164-279, scalar clocks)
        int64_t bt, bv; uint32_t owner = ~0u;
        bool created_here = false;
        if (is_new) {
          created_here = true; created++;
          bt = (MODE == BMX_INSERT_REFERENCE) ? 2 : t_j0; bv = v_j0; owner = j0;        // :172-185: first write stores clock {local: 2}
          int64_t tm, vm; uint32_t om;
          if (top.best_except(j0, tm, vm, om) && lexcmp(tm, vm, bt, bv) > 0) { bt = tm; bv = vm; owner = om; }   // a tie keeps j0 (smaller index)
        } else {
          bt = cts; bv = cval;
          if (lexcmp(top.t1, top.v1, bt, bv) > 0) { bt = top.t1; bv = top.v1; owner = top.o1; }
        }
        Slot* sl = A.slots + s;
        const uint32_t tag = (A.epoch << IDX_BITS) | (owner & IDX_MASK);
        if (created_here) {
          // field + winner tag in one 8-byte store, then the pair: same 32-byte sector, one write-back
          *reinterpret_cast<unsigned long long*>(&sl->field) = (unsigned long long)field | ((unsigned long long)tag << 32);
          store_tv(sl, bt, bv);
          A.wflag[owner] = 1;
        } else if (owner != ~0u) {
          if ((head >> IDX_BITS) == A.epoch) A.wflag[head & IDX_MASK] = 0;   // an earlier chunk of this batch had named a winner for this row
          sl->head = tag;
          store_tv(sl, bt, bv);
          A.wflag[owner] = 1;
        }
        if (A.flags) {
          // decision flags relative to the row as this chunk found it (exact whenever the key occurs once in the batch)
          for (uint32_t p = i;;) {
            const uint32_t jx = r_lo[p].w; const uint4 d = r_hi[p];
            const int64_t t = i64_of(d.x, d.y), v = i64_of(d.z, d.w);
            uint32_t fl;
            if (jx == owner || created_here) fl = BMX_FLAG_INCOMING;
            else { const int c = lexcmp(t, v, cts, cval); fl = c > 0 ? BMX_FLAG_INCOMING : (c == 0 ? 0u : (BMX_FLAG_CURRENT | (t < cts ? BMX_FLAG_HISTORICAL : 0u))); }
            A.flags[jx] = (uint8_t)fl;
            p = (p == i) ? (lhead[i] & 0xFFFFu) : nxt[p];
            if (p == NXT_END) break;
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's stores have reached L2 before anybody looks again
      if (!__syncthreads_or(pending_mask != 0)) break;
      if (round > MB_CAP) { if (pending_mask) atomicOr(A.status, ST_SPIN); break; }   // cannot happen: every round publishes at least one field
    }
    // ---- 4. next chunk ----
    if (cut > 0) t0 += cut;
    else if (++sub == MB_SUB) { sub = 0; t0 += 1; }
    __syncthreads();
  }
  // per-workgroup counts -> sharded counters (folded by the compaction's last block)
  {
    uint32_t tc, tx;
    mb_incl_scan(created, tc, wsum);
    mb_incl_scan(conflicts, tx, wsum);
    if (tid == 0 && (tc | tx)) {
      unsigned long long* ctr = A.shard_ctr + (size_t)(b & (CTR_SHARDS - 1)) * CTR_STRIDE;
      if (tc) atomicAdd(ctr + 0, (unsigned long long)tc);
      if (tx) atomicAdd(ctr + 1, (unsigned long long)tx);
    }
  }
}

// winners per 256-delta block (what k_compact_winners ranks from): one 16-byte load per lane, 4096 deltas per workgroup
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

}  // namespace bmx
