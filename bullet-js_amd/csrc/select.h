// select.h — ordered (stable) stream compaction: the prefix-sum winner compaction K3 and the match
// compaction of the index scans K5/K6 share these skeletons. Output order = element order (deterministic).
//
// ONE launch (k_select / k_select_staged): block b counts the selected elements of its contiguous chunk, publishes
//   the count as an 8-byte {call sequence number, count} granule with ONE agent-scope atomic store
//   sums the granules of blocks 0..b-1 (relaxed agent-scope loads, all of a thread's polls in flight together),
//   then writes its chunk at the global rank. Every block publishes BEFORE it waits and the grid (<= 1024 blocks of
//   256 threads) is fully resident, so the waits terminate; spins are bounded anyway. This is the "data is the
//   flag" hand-off of cdna_hip_programming.md G16 (R2): no fence, no plain loads of handed-off bytes; the sequence
//   number makes re-initialisation unnecessary (granules are re-zeroed when the sequence wraps).
// TWO launches (k_sel_count + k_sel_write): used where the host needs the total before the output exists
//   (index build sizes its columns from it).
//
// Pred  : struct { static constexpr int E; __device__ uint32_t mask(uint64_t first, uint64_t n) const; }
//         thread owns E consecutive elements [first, first+E); bit e set <=> element first+e selected (and < n)
// Emit  : __device__ void operator()(uint64_t pos, uint64_t elem) const
// Finish: __device__ void operator()(uint64_t total, uint32_t* lds4) const   (every thread of the last block)
#pragma once
#include "slot.h"

namespace bmx {

constexpr int SEL_THREADS = 256;
constexpr int SEL_MAX_BLOCKS = 1024;   // <= 4 blocks of 256 threads per CU: the whole grid is resident
constexpr int SEL_STAGE = 12;          // tiles whose masks a block keeps in registers between counting and writing

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
  uint32_t x = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d);
    if ((int)lane >= d) x += y;
  }
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
  uint32_t part = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SEL_THREADS) part += block_counts[b];
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


// ---- look-back over per-block aggregates ----
// granule = call sequence number << 32 | this block's count, published with ONE agent-scope atomic store before the
// block waits for anything. Block b sums granules 0..b-1: each thread owns at most SEL_MAX_BLOCKS/256 of them and
// issues all its loads together, re-polling only those not yet published. There is no block-to-block chain (a
// chained inclusive-prefix look-back measured slower here: ~0.8 us per cross-CU hop x 32 hops).
__device__ __forceinline__ uint32_t lookback_exclusive(unsigned long long* granules, uint32_t seq, uint32_t mine, uint32_t* status,
                                                       uint32_t* wsum /* 4 LDS words */) {
  if (threadIdx.x == 0)
    __hip_atomic_store(granules + blockIdx.x, ((unsigned long long)seq << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  constexpr int NP = SEL_MAX_BLOCKS / SEL_THREADS;
  uint32_t part = 0, pending = 0;
#pragma unroll
  for (int k = 0; k < NP; k++) if (threadIdx.x + k * SEL_THREADS < blockIdx.x) pending |= 1u << k;
  uint32_t spins = 0;
  while (pending) {
    unsigned long long g[NP];
#pragma unroll
    for (int k = 0; k < NP; k++)
      g[k] = (pending >> k & 1u) ? __hip_atomic_load(granules + threadIdx.x + k * SEL_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
    for (int k = 0; k < NP; k++)
      if ((pending >> k & 1u) && (uint32_t)(g[k] >> 32) == seq) { part += (uint32_t)g[k]; pending &= ~(1u << k); }
    if (pending) {
      if (++spins > (1u << 22)) { atomicOr(status, ST_SPIN); break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  uint32_t total;
  block_excl_scan(part, total, wsum);
  return total;
}

// ---- single-launch variant with look-back granules ----
template <class Pred, class Emit, class Finish>
__global__ __launch_bounds__(SEL_THREADS) void k_select(Pred P, Emit Em, Finish Fin, uint64_t n, uint32_t tiles_per_block,
                                                         unsigned long long* granules, uint32_t seq, uint32_t* status) {
  __shared__ uint32_t wsum[4];
  constexpr int E = Pred::E;
  const uint64_t tile = (uint64_t)SEL_THREADS * E;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_block;
  // 1. count this block's chunk and publish it
  uint32_t cnt = 0;
  for (uint32_t k = 0; k < tiles_per_block; k++) {
    uint64_t first = (t0 + k) * tile + (uint64_t)threadIdx.x * E;
    if (first < n) cnt += __popc(P.mask(first, n));
  }
  uint32_t mine;
  block_excl_scan(cnt, mine, wsum);
  // 2. rank of the chunk's first selected element = sum of the predecessors' counts (decoupled look-back)
  const uint32_t offset = lookback_exclusive(granules, seq, mine, status, wsum);
  // 3. re-evaluate (the chunk is L2-hot) and write in order
  uint64_t running = offset;
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

// Register-staged variant for tiles_per_block <= SEL_STAGE: every load of the chunk is issued before the first
// wait (SEL_STAGE x 16 B in flight per lane), the predicate masks stay in registers, the input is read ONCE.
template <class Pred, class Emit, class Finish>
__global__ __launch_bounds__(SEL_THREADS) void k_select_staged(Pred P, Emit Em, Finish Fin, uint64_t n, uint32_t tiles_per_block,
                                                                unsigned long long* granules, uint32_t seq, uint32_t* status) {
  __shared__ uint32_t wsum[4];
  constexpr int E = Pred::E;
  const uint64_t tile = (uint64_t)SEL_THREADS * E;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_block;
  uint32_t m[SEL_STAGE];
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < SEL_STAGE; k++) {
    uint64_t first = (t0 + k) * tile + (uint64_t)threadIdx.x * E;
    m[k] = ((uint32_t)k < tiles_per_block && first < n) ? P.mask(first, n) : 0u;
  }
#pragma unroll
  for (int k = 0; k < SEL_STAGE; k++) cnt += __popc(m[k]);
  uint32_t mine;
  block_excl_scan(cnt, mine, wsum);
  const uint32_t offset = lookback_exclusive(granules, seq, mine, status, wsum);
  uint64_t running = offset;
#pragma unroll
  for (int k = 0; k < SEL_STAGE; k++) {
    if ((uint32_t)k < tiles_per_block) {   // uniform across the block
      uint64_t first = (t0 + k) * tile + (uint64_t)threadIdx.x * E;
      uint32_t mk = m[k], tot;
      uint32_t ex = block_excl_scan((uint32_t)__popc(mk), tot, wsum);
      uint64_t pos = running + ex;
      while (mk) {
        int e = __ffs((int)mk) - 1;
        mk &= mk - 1;
        Em(pos++, first + (uint64_t)e);
      }
      running += tot;
    }
  }
  if (blockIdx.x == gridDim.x - 1) Fin(running, wsum);
}

}  // namespace bmx
