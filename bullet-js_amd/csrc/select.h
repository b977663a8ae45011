// select.h — ordered (stable) stream compaction skeletons. Output order = element order (deterministic). Workgroups never
// talk to each other inside a launch: ranks always come from counts an EARLIER launch left behind (a single-launch
// look-back variant was built and measured: its cross-CU hand-off costs ~4 us per launch and forces a second read of
// large columns).
//
// k_sel_count + k_sel_write : generic two-launch select over any predicate (index build, row dump); the predicate is
//                              evaluated twice.
// k_scan_mask + k_scan_emit  : index scans; the column is read once, pass 2 works from a 1-bit-per-row mask.
//
// Pred  : struct { static constexpr int E; __device__ uint32_t mask(uint64_t first, uint64_t n) const; }
//         thread owns E consecutive elements [first, first+E); bit e set <=> element first+e selected (and < n)
// Emit  : __device__ void operator()(uint64_t pos, uint64_t elem) const
// Finish: __device__ void operator()(uint64_t total, uint32_t* lds4) const   (every thread of the last block)
#pragma once
#include "slot.h"

namespace bmx {

constexpr int SEL_THREADS = 256;
constexpr int SEL_MAX_BLOCKS = 1024;          // tiles whose masks a block keeps in registers between counting and writing

struct SelGeom {
  uint32_t blocks;
  uint32_t tiles_per_block;
};
template <int E>
inline SelGeom sel_geom(uint64_t n) {
  uint64_t tile = (uint64_t)SEL_THREADS * E;
  uint64_t tiles = (n + tile - 1) / tile;
  if (tiles == 0) tiles = 1;
  uint64_t blocks = tiles < (uint64_t)SEL_MAX_BLOCKS ? tiles : (uint64_t)SEL_MAX_BLOCKS;
  uint64_t tpb = (tiles + blocks - 1) / blocks;
  blocks = (tiles + tpb - 1) / tpb;
  return SelGeom{(uint32_t)blocks, (uint32_t)tpb};
}

// exclusive scan of one value per thread over a 256-thread block; total returned to every thread
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t cnt, uint32_t& total, uint32_t* wsum /*[4] LDS*/) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t x = wave_incl_scan_u32(cnt);   // DPP: no LDS round trips (callers run it with every lane active)
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SEL_THREADS / 64; i++) {
    uint32_t s = wsum[i];
    if (i < (int)w) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + x - cnt;
}

// this thread's share of sum(c[0..n)), eight loads in flight: a plain "load, add" loop pays one L2 round trip per iteration
__device__ __forceinline__ uint32_t strided_partial_sum(const uint32_t* __restrict__ c, uint32_t n) {
  uint32_t part = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 8 * SEL_THREADS) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t b = b0 + (uint32_t)u * SEL_THREADS + threadIdx.x;
      v[u] = b < n ? c[b] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) part += v[u];
  }
  return part;
}

template <class Pred>
__global__ __launch_bounds__(SEL_THREADS) void k_sel_count(Pred P, uint64_t n, uint32_t tiles_per_block,
                                                            uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[4];
  constexpr int E = Pred::E;
  const uint64_t tile = (uint64_t)SEL_THREADS * E;
  uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_block;
  uint32_t cnt = 0;
  for (uint32_t k = 0; k < tiles_per_block; k++) {
    uint64_t first = (t0 + k) * tile + (uint64_t)threadIdx.x * E;
    if (first < n) cnt += __popc(P.mask(first, n));
  }
  uint32_t total;
  block_excl_scan(cnt, total, wsum);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

template <class Pred, class Emit, class Finish>
__global__ __launch_bounds__(SEL_THREADS) void k_sel_write(Pred P, Emit Em, Finish Fin, uint64_t n, uint32_t tiles_per_block,
                                                            const uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[4];
  constexpr int E = Pred::E;
  const uint64_t tile = (uint64_t)SEL_THREADS * E;
  // global rank of this block's first selected element
  uint32_t part = strided_partial_sum(block_counts, blockIdx.x);
  uint32_t offset;
  block_excl_scan(part, offset, wsum);
  uint64_t running = offset;
  uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_block;
  for (uint32_t k = 0; k < tiles_per_block; k++) {
    uint64_t first = (t0 + k) * tile + (uint64_t)threadIdx.x * E;
    uint32_t m = first < n ? P.mask(first, n) : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan((uint32_t)__popc(m), tot, wsum);
    uint64_t pos = running + ex;
    while (m) {
      int e = __ffs((int)m) - 1;
      m &= m - 1;
      Em(pos++, first + (uint64_t)e);
    }
    running += tot;
  }
  if (blockIdx.x == gridDim.x - 1) Fin(running, wsum);
}


// ---- index scans: match mask + counts (pass 1), emit from the mask (pass 2) ----
// Pass 1 streams the value column ONCE (every lane has 32/E x 16 B in flight), packs the per-lane match bits into one
// bit per element with wave shuffles and leaves one count per 8192-element block. Pass 2 reads only the mask (1/32 of an
// int32 column) and writes the matching ids at ranks derived from the counts of the PREVIOUS launch: no in-launch
// communication between workgroups, any column size, deterministic element order.
constexpr uint32_t SCAN_BLOCK_ELEMS = 8192;   // elements per block in both passes = 256 mask words
constexpr uint32_t SCAN_SUB8_BLOCKS = 2048;   // beyond this many blocks (16.78M rows) the emit pass takes eight blocks per workgroup
// From this many matches in an 8192-row block (29 %) the id output STREAMS the block's 64 KB of the id column (16 bytes per lane, coalesced) instead of
// gathering one id per match, packs the matches in LDS and writes them as 16-byte stores. Measured in one process on one index
// (bench_micro/scan_stream_ab.py, profiles/r04_scan_stream_ab.log, 100M rows, whole scan, two boxes): 20 % of the rows 266 / 286 us streamed against
// 262 / 275 gathered, 50 % 315 / 328 against 338 / 362, 100 % 372 against 514-531 — but 10 % 245-269 against 217-220: the gather touches most of the lines
// from 10 % on (81 %, profiles/traffic_scan.json) yet still moves less than the whole column, so the switch sits where the stream wins, not where the
// traffic curves cross. (The first form of the round stored each lane's matches with 8-byte stores straight from registers: 334 us at 50 %, 418 at
// 100 %, and won only from 40 %.)
constexpr uint32_t SCAN_STREAM_MIN = 2400;

// One 8192-row block of the id output, streamed (Emit::STREAMABLE): wave w takes rows [2048 w, 2048 (w + 1)) = the 64 mask words its own lanes
// hold (thread t owns word t), 128 rows per step, two consecutive ids (16 B) per lane. The four mask words of a step and the rank in front of
// them come from the lanes that own them (readlane: no barrier); a lane's rank = that base + the set bits below its two. `r` = exclusive
// rank of the thread's own word inside the block, `pos0` = rank of the block's first match in the whole answer. Every 512 rows the wave's matches,
// packed in rank order in ITS 4 KB of LDS (`stg`; nobody else touches it: wave-level ordering only), go out as 16-byte stores.
template <int U, class Emit>
__device__ __forceinline__ void scan_emit_stream_block(const Emit& Em, uint32_t mk, uint32_t r, uint64_t pos0, uint64_t block_row0, uint64_t n, uint64_t* stg) {
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint64_t row_w = block_row0 + (uint64_t)w * 2048u + 2u * lane;
  const uint32_t sub = lane >> 4, sh = 2u * (lane & 15u);          // which of the step's four words holds this lane's two rows, and where
  const bool whole = block_row0 + SCAN_BLOCK_ELEMS <= n;             // (uniform) every row of the block exists: loads need no guard
  // U 16-byte loads in flight per lane before the first is used (8: round 4; 16: the wave's whole 2048 rows — A/B switch BMX_SCAN_NT bit 2)
  constexpr int F = 4;                                                // steps (of 128 rows) per flush of the wave's LDS
  const uint32_t r_end = (uint32_t)__builtin_amdgcn_readlane((int)r, 63) + (uint32_t)__popc((uint32_t)__builtin_amdgcn_readlane((int)mk, 63));   // rank behind the wave's last row
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < 16; i0 += U) {
    uint64_t a[U], b[U];
    if (whole) {
#pragma unroll
      for (int u = 0; u < U; u++) Em.load2(row_w + (uint64_t)(i0 + u) * 128u, a[u], b[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint64_t row = row_w + (uint64_t)(i0 + u) * 128u;
        a[u] = 0; b[u] = 0;
        if (row + 2 <= n) Em.load2(row, a[u], b[u]); else if (row < n) a[u] = Em.load1(row);
      }
    }
    // the matches of 512 rows at a time are packed into the wave's 4 KB of LDS in rank order ...
#pragma unroll
    for (int h = 0; h < U; h += F) {
      const uint32_t ih = i0 + (uint32_t)h;
      const uint32_t rb0 = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)(4 * ih));
#pragma unroll
      for (int u = h; u < h + F; u++) {
        const uint32_t i = i0 + (uint32_t)u;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mk, (int)(4 * i)), w1 = (uint32_t)__builtin_amdgcn_readlane((int)mk, (int)(4 * i + 1));
        const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)mk, (int)(4 * i + 2)), w3 = (uint32_t)__builtin_amdgcn_readlane((int)mk, (int)(4 * i + 3));
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)(4 * i));
        const uint32_t mine = sub == 0 ? w0 : (sub == 1 ? w1 : (sub == 2 ? w2 : w3));
        const uint32_t before = (sub > 0 ? __popc(w0) : 0u) + (sub > 1 ? __popc(w1) : 0u) + (sub > 2 ? __popc(w2) : 0u) + __popc(mine & ((1u << sh) - 1u));
        const uint32_t bits = (mine >> sh) & 3u;
        const uint32_t loc = rb - rb0 + before;
        if (bits & 1u) stg[loc] = a[u];
        if (bits & 2u) stg[loc + (bits & 1u)] = b[u];
      }
      // ... and leave it as 16-byte stores, consecutive lanes on consecutive pairs (a first id in the upper half of a 16-byte unit goes on its own)
      const uint32_t T = (ih + F < 16 ? (uint32_t)__builtin_amdgcn_readlane((int)r, (int)(4 * ((ih + F) & 15u))) : r_end) - rb0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint64_t p0 = pos0 + rb0;
      const uint32_t head = T ? Em.odd(p0) : 0u;
      if (head && lane == 0) Em.put(p0, stg[0]);
      const uint32_t rest = T - head, npairs = rest >> 1;
      for (uint32_t q = lane; q < npairs; q += 64) { const uint32_t e = head + 2u * q; Em.put2(p0 + e, stg[e], stg[e + 1]); }
      if ((rest & 1u) && lane == 1) Em.put(p0 + T - 1, stg[T - 1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <class Pred, bool WRITE_MASK>
__global__ __launch_bounds__(SEL_THREADS) void k_scan_mask(Pred P, uint64_t n, uint32_t* __restrict__ mask_words, uint32_t* __restrict__ block_counts) {
  constexpr int E = Pred::E;
  constexpr int TILES = 32 / E;   // tiles of 256*E elements per block
  constexpr int LPW = 32 / E;     // lanes that share one 32-bit mask word
  __shared__ uint32_t wsum[4];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK_ELEMS;
  uint32_t m[TILES];
#pragma unroll
  for (int k = 0; k < TILES; k++) {
    uint64_t first = base + (uint64_t)k * SEL_THREADS * E + (uint64_t)threadIdx.x * E;
    m[k] = first < n ? P.mask(first, n) : 0u;
  }
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < TILES; k++) {
    cnt += __popc(m[k]);
    if (WRITE_MASK) {
      uint32_t v = m[k] << (E * (threadIdx.x & (LPW - 1)));
#pragma unroll
      for (int d = 1; d < LPW; d <<= 1) v |= __shfl_xor(v, d);
      if ((threadIdx.x & (LPW - 1)) == 0)
        mask_words[(base + (uint64_t)k * SEL_THREADS * E + (uint64_t)threadIdx.x * E) >> 5] = v;
    }
  }
  uint32_t total;
  block_excl_scan(cnt, total, wsum);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// SUB consecutive 8192-row blocks per workgroup; the rank of its first match = the counts of all blocks before its first block, summed by the
// workgroup itself (counts written by the PREVIOUS launch). SUB = 1 for columns up to 2048 blocks; SUB = 8 for large ones: an eighth of the
// workgroups, each sums <= 48 KB of counts from L2 once (100M-row column) and carries the rank through its blocks. (Round 1 ran a one-block
// prefix kernel between the two passes for large columns: 7 us of launch + boundary for a 100M-row column; dropped.)
template <class Emit, class Finish, int SUB>
__global__ __launch_bounds__(SEL_THREADS) void k_scan_emit(const uint32_t* __restrict__ mask_words, const uint32_t* __restrict__ counts,
                                                            uint64_t n, uint32_t nblocks, Emit Em, Finish Fin) {
  __shared__ uint32_t wsum[4];
  // block-local row offsets of the matches, in rank order (16 KB) — or, while a dense block streams, the packed ids of 512 rows per wave (4 x 4 KB)
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * SCAN_BLOCK_ELEMS];
  uint16_t* loc = reinterpret_cast<uint16_t*>(lds_raw);
  const uint32_t b0 = blockIdx.x * (uint32_t)SUB;
  uint32_t offset;
  {
    uint32_t part = strided_partial_sum(counts, b0);
    block_excl_scan(part, offset, wsum);
  }
  uint64_t running = offset;
  if (SUB > 1) {
    // fast path (<= 8192 matches in the workgroup's SUB blocks, i.e. selectivity up to 1/SUB): a thread owns SUB consecutive mask words
    // (SUB*32 consecutive rows), so ONE block scan ranks every match of the SUB blocks; otherwise block by block below
    uint32_t mk[SUB];
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < SUB; q++) {
      const uint64_t w = (uint64_t)b0 * SEL_THREADS + (uint64_t)threadIdx.x * SUB + q;
      mk[q] = (w << 5) < n ? mask_words[w] : 0u;
      cnt += __popc(mk[q]);
    }
    uint32_t tot;
    uint32_t r = block_excl_scan(cnt, tot, wsum);
    // (dense id output goes block by block below, where every block streams its part of the id column)
    if (tot <= SCAN_BLOCK_ELEMS && !(Emit::STREAMABLE && tot >= (uint32_t)SUB * Em.stream_from())) {
#pragma unroll
      for (int q = 0; q < SUB; q++) {
        uint32_t m = mk[q];
        while (m) { const int e = __ffs((int)m) - 1; m &= m - 1; loc[r++] = (uint16_t)((threadIdx.x * SUB + q) * 32u + (uint32_t)e); }
      }
      __syncthreads();
      const uint64_t base = (uint64_t)b0 * SCAN_BLOCK_ELEMS;
      for (uint32_t q = threadIdx.x; q < tot; q += SEL_THREADS) Em(running + q, base + loc[q]);
      if (blockIdx.x == gridDim.x - 1) Fin(running + tot, wsum);
      return;
    }
  }
#pragma unroll 1
  for (uint32_t k = 0; k < (uint32_t)SUB; k++) {
    const uint32_t blk = b0 + k;
    if (blk >= nblocks) break;
    const uint64_t w = (uint64_t)blk * SEL_THREADS + threadIdx.x;
    uint32_t mk = (w << 5) < n ? mask_words[w] : 0u;
    uint32_t tot;
    uint32_t r = block_excl_scan((uint32_t)__popc(mk), tot, wsum);
    if (Emit::STREAMABLE && tot >= Em.stream_from()) {      // (uniform over the workgroup)
      if (Em.deep()) scan_emit_stream_block<16>(Em, mk, r, running, (uint64_t)blk * SCAN_BLOCK_ELEMS, n, reinterpret_cast<uint64_t*>(lds_raw) + (threadIdx.x >> 6) * 512u);
      else scan_emit_stream_block<8>(Em, mk, r, running, (uint64_t)blk * SCAN_BLOCK_ELEMS, n, reinterpret_cast<uint64_t*>(lds_raw) + (threadIdx.x >> 6) * 512u);
      running += tot;
      continue;
    }
    // transpose through LDS: a lane owns 32 consecutive rows, but the output wants consecutive lanes on consecutive ranks
    while (mk) {
      int e = __ffs((int)mk) - 1;
      mk &= mk - 1;
      loc[r++] = (uint16_t)(threadIdx.x * 32u + (uint32_t)e);
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blk * SCAN_BLOCK_ELEMS;
    for (uint32_t q = threadIdx.x; q < tot; q += SEL_THREADS) Em(running + q, base + loc[q]);   // coalesced stores, near-sequential gathers
    running += tot;
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1) Fin(running, wsum);
}

}  // namespace bmx
