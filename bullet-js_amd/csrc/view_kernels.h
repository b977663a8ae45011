// view_kernels.h — keeping the value-ordered view of an index CURRENT under writes (round 5), gfx950, hand-written.
//
// The reference maintains its value-keyed index on every write: _updateIndices moves the path from the bucket of the old value to the
// bucket of the new one (src/bullet-query.js:139-176 -> _removeFromIndex / _addToIndex :82-118). The device's value-ordered view
// (scan_kernels.h: the index columns sorted by (value, position)) used to be thrown away by any write to its field and sorted again from
// scratch. Here the merges' change log is turned into a CHANGE RUN — for every index row whose value really changed: the key it had,
// (old value, position), and the key it has now; for every appended row: its key — the run is sorted by these kernels (an LDS bitonic
// sort of 2048-key tiles + rank-merge passes: the run is at most a few million keys and lives in L2 / the Infinity Cache) and merged into the
// view in ONE streaming pass: every key of the old view that is not deleted moves to (its index - deleted keys in front of it + inserted
// keys in front of it), every inserted key to (its index + surviving keys in front of it). No atomics, no ordering between workgroups:
// every output element is computed and written by exactly one thread from ranks in sorted, read-only inputs.
//
// Keys are (value, position) pairs compared lexicographically; a position occurs at most once among the live keys, so keys are unique.
// Bytes per patched view of n rows, c changed and a appended rows: n * (w + 12) read + (n + a) * (w + 12) written (w = 4 or 8: the value
// width), + O((c + a) log(c + a)) cached traffic for the sort: HBM-stream bound.
#pragma once
#include "scan_kernels.h"

namespace bmx {

constexpr uint32_t VIEW_TILE = 2048;             // keys per LDS-sorted tile = keys per workgroup of the streaming merge
constexpr uint32_t VIEW_WIN = 1024;              // the deleted indices / inserted keys that fall into one tile of the view are searched in LDS up to this many

template <class T>
__device__ __forceinline__ bool vk_less(T av, uint32_t ap, T bv, uint32_t bp) { return av < bv || (av == bv && ap < bp); }

// first index in [lo, hi) whose key is not less than (kv, kp); UPPER: first index whose key is greater
template <class T, bool UPPER = false>
__device__ __forceinline__ uint64_t vk_bound(const T* __restrict__ v, const uint32_t* __restrict__ p, uint64_t lo, uint64_t hi, T kv, uint32_t kp) {
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    const T mv = v[mid]; const uint32_t mp = p[mid];
    const bool before = UPPER ? !vk_less<T>(kv, kp, mv, mp) : vk_less<T>(mv, mp, kv, kp);
    if (before) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t u32_lower_bound(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t key) {
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
// The same bounds found by a whole WAVE (every lane calls it with the same arguments and gets the answer): 64-ary search like ordered_bound — every round
// 63 lanes probe evenly spaced keys of [L, R) and a ballot says between which two the bound lies. A binary search over a run of a million keys is a chain of
// 20 dependent loads (10-15 us of latency that every workgroup of a merge pass paid before its first key moved); this is four rounds.
template <class T, bool UPPER = false>
__device__ __forceinline__ uint64_t vk_bound_wave(const T* __restrict__ v, const uint32_t* __restrict__ p, uint64_t L, uint64_t R, T kv, uint32_t kp) {
  const uint32_t lane = threadIdx.x & 63u;
  while (R - L > 64) {
    const uint64_t step = (R - L + 63) / 64;
    const uint64_t c = L + (uint64_t)lane * step;        // lane 0 probes nothing (c == L)
    bool before = false;
    if (lane > 0 && c < R) { const T x = v[c]; const uint32_t xp = p[c]; before = UPPER ? !vk_less<T>(kv, kp, x, xp) : vk_less<T>(x, xp, kv, kp); }
    const uint32_t t = (uint32_t)__popcll(__ballot(before));        // sorted keys: lanes 1..t are in front of the bound
    const uint64_t nl = t ? L + (uint64_t)t * step + 1 : L;
    const uint64_t cr = L + (uint64_t)(t + 1) * step;
    R = (t < 63 && cr < R) ? cr : R;
    L = nl;
  }
  bool before = false;
  if (L + lane < R) { const T x = v[L + lane]; const uint32_t xp = p[L + lane]; before = UPPER ? !vk_less<T>(kv, kp, x, xp) : vk_less<T>(x, xp, kv, kp); }
  return L + (uint64_t)__popcll(__ballot(before));
}
__device__ __forceinline__ uint32_t u32_lower_bound_wave(const uint32_t* __restrict__ a, uint32_t L, uint32_t R, uint32_t key) {
  const uint32_t lane = threadIdx.x & 63u;
  while (R - L > 64) {
    const uint32_t step = (R - L + 63) / 64;
    const uint32_t c = L + lane * step;
    const bool before = lane > 0 && c < R && a[c] < key;
    const uint32_t t = (uint32_t)__popcll(__ballot(before));
    const uint32_t nl = t ? L + t * step + 1 : L;
    const uint32_t cr = L + (t + 1) * step;
    R = (t < 63 && cr < R) ? cr : R;
    L = nl;
  }
  const bool before = L + lane < R && a[L + lane] < key;
  return L + (uint32_t)__popcll(__ballot(before));
}

// Up to two independent key arrays sorted by the same launches (the deleted keys and the inserted keys of one patch): segment s occupies
// [base[s], base[s] + len[s]) of the key arrays and blocks [blk0[s], blk0[s + 1]) of the grid.
struct ViewSegs {
  uint32_t base[2], len[2], blk0[3];
  __device__ __forceinline__ uint32_t seg_of(uint32_t block) const { return block >= blk0[1] ? 1u : 0u; }
};

// The change run of one refresh (k_ix_update, capture mode): positions whose value changed + the value they had. Turned into sort keys:
//   segment 0 [0, c)            deleted keys  (old value, position)
//   segment 1 [c, 2c)           inserted keys (current value, position) of the changed rows
//             [2c, 2c + added)  inserted keys of the rows appended to the index columns, positions n0 .. n0 + added - 1
template <class T>
__global__ __launch_bounds__(256) void k_view_keys(const uint32_t* __restrict__ cl_pos, const int64_t* __restrict__ cl_old, uint32_t c, const T* __restrict__ col, uint32_t n0, uint32_t added,
                                                   T* __restrict__ kv, uint32_t* __restrict__ kp) {
  constexpr T TOMB = sizeof(T) == 4 ? (T)INT32_MIN : (T)INT64_MIN;
  const uint32_t total = 2u * c + added;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    if (i < c) { const int64_t o = cl_old[i]; kv[i] = o == VAL_DELETED ? TOMB : (T)o; kp[i] = cl_pos[i]; }
    else if (i < 2u * c) { const uint32_t p = cl_pos[i - c]; kv[i] = col[p]; kp[i] = p; }
    else { const uint32_t p = n0 + (i - 2u * c); kv[i] = col[p]; kp[i] = p; }
  }
}

// One tile of 2048 keys sorted in LDS (bitonic network, 66 compare-exchange rounds of 1024 pairs over 256 threads). A ragged last tile is padded with
// keys behind every real one; only the real keys are written back.
template <class T>
__global__ __launch_bounds__(256) void k_view_tile_sort(const T* __restrict__ vin, const uint32_t* __restrict__ pin, T* __restrict__ vout, uint32_t* __restrict__ pout, ViewSegs S) {
  constexpr T VMAX = sizeof(T) == 4 ? (T)INT32_MAX : (T)INT64_MAX;
  __shared__ T sv[VIEW_TILE];
  __shared__ uint32_t sp[VIEW_TILE];
  const uint32_t s = S.seg_of(blockIdx.x);
  const uint32_t t0 = (blockIdx.x - S.blk0[s]) * VIEW_TILE;          // first key of the tile inside its segment
  const uint32_t len = S.len[s], base = S.base[s];
#pragma unroll
  for (uint32_t u = 0; u < VIEW_TILE / 256u; u++) {
    const uint32_t e = u * 256u + threadIdx.x;
    const bool ok = t0 + e < len;
    sv[e] = ok ? vin[base + t0 + e] : VMAX; sp[e] = ok ? pin[base + t0 + e] : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= VIEW_TILE; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (uint32_t u = 0; u < VIEW_TILE / 512u; u++) {
        const uint32_t t = u * 256u + threadIdx.x;                   // pair number
        const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));  // lower element of the pair: bit j clear
        const uint32_t q = i | j;
        const T av = sv[i], bv = sv[q]; const uint32_t ap = sp[i], bp = sp[q];
        const bool up = (i & k) == 0;                                 // this stretch is sorted ascending
        if (vk_less<T>(bv, bp, av, ap) == up) { sv[i] = bv; sp[i] = bp; sv[q] = av; sp[q] = ap; }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < VIEW_TILE / 256u; u++) {
    const uint32_t e = u * 256u + threadIdx.x;
    if (t0 + e < len) { vout[base + t0 + e] = sv[e]; pout[base + t0 + e] = sp[e]; }
  }
}

// One pass of the merge sort: inside every segment, sorted runs of L keys (L a multiple of 2048) are merged pairwise into runs of 2L. Rank merge: a key's
// place in the merged run = its index in its own run + the keys of the partner run in front of it (strictly smaller for the left run, smaller or equal
// for the right one: stable), found by a binary search that the workgroup's first and last key bound for all 256 (the keys of a workgroup are
// consecutive in one sorted run, so their ranks lie between those two).
template <class T>
__global__ __launch_bounds__(256) void k_view_merge_pass(const T* __restrict__ vin, const uint32_t* __restrict__ pin, T* __restrict__ vout, uint32_t* __restrict__ pout, ViewSegs S, uint32_t L) {
  __shared__ uint32_t win[2];
  const uint32_t s = S.seg_of(blockIdx.x);
  const uint32_t len = S.len[s], base = S.base[s];
  const uint32_t e0 = (blockIdx.x - S.blk0[s]) * 256u;             // first key of this workgroup inside the segment
  if (e0 >= len) return;
  const uint32_t r = e0 / L, q = r ^ 1u;                             // own run, partner run
  const bool left = (r & 1u) == 0;
  const uint32_t q0 = q * L < len ? q * L : len, q1 = (q + 1u) * L < len ? (q + 1u) * L : len;       // partner run [q0, q1): may be empty (odd run count)
  const uint32_t e = e0 + threadIdx.x;
  const uint32_t last = (e0 + 255u < len ? e0 + 255u : len - 1u);
  const T* V = vin + base; const uint32_t* P = pin + base;
  const uint32_t w = threadIdx.x >> 6;
  if (w < 2) {           // wave 0: where the workgroup's first key stands in the partner run, wave 1: where its last key does
    const uint32_t k = w == 0 ? e0 : last;
    const uint32_t b = q0 == q1 ? q0 : (uint32_t)(left ? vk_bound_wave<T, false>(V, P, q0, q1, V[k], P[k]) : vk_bound_wave<T, true>(V, P, q0, q1, V[k], P[k]));
    if ((threadIdx.x & 63u) == 0) win[w] = b;
  }
  __syncthreads();
  if (e >= len) return;
  const T kv = V[e]; const uint32_t kp = P[e];
  const uint32_t rank = (uint32_t)(left ? vk_bound<T, false>(V, P, win[0], win[1], kv, kp) : vk_bound<T, true>(V, P, win[0], win[1], kv, kp)) - q0;
  const uint32_t out = (r >> 1) * 2u * L + (e - r * L) + rank;
  vout[base + out] = kv; pout[base + out] = kp;
}

// Where the deleted keys stand in the view: dx[i] = index of sorted deleted key i in (xv, xp), ascending because both are sorted. A key that is not there
// means the view and the index columns have drifted apart: *err is set and the caller sorts the view from scratch.
template <class T>
__global__ __launch_bounds__(256) void k_view_find(const T* __restrict__ xv, const uint32_t* __restrict__ xp, uint32_t nx, const T* __restrict__ dv, const uint32_t* __restrict__ dp, uint32_t nd,
                                                   uint32_t* __restrict__ dx, uint32_t* __restrict__ err) {
  __shared__ uint32_t win[2];
  const uint32_t i0 = blockIdx.x * 256u;
  if (i0 >= nd) return;
  const uint32_t last = i0 + 255u < nd ? i0 + 255u : nd - 1u;
  const uint32_t w = threadIdx.x >> 6;
  if (w < 2) {
    const uint32_t k = w == 0 ? i0 : last;
    const uint32_t b = (uint32_t)vk_bound_wave<T>(xv, xp, 0, nx, dv[k], dp[k]);
    if ((threadIdx.x & 63u) == 0) win[w] = b;
  }
  __syncthreads();
  const uint32_t i = i0 + threadIdx.x;
  if (i >= nd) return;
  const T kv = dv[i]; const uint32_t kp = dp[i];
  const uint32_t at = (uint32_t)vk_bound<T>(xv, xp, win[0], win[1] < nx ? win[1] + 1u : nx, kv, kp);
  const bool found = at < nx && xv[at] == kv && xp[at] == kp;
  dx[i] = at;
  if (!found) *err = 1u;
}

template <class T>
struct ViewRun { T* v; uint32_t* p; uint64_t* ids; };

// The streaming merge: Z = (X without the keys at the sorted indices dx[0..ndx)) merged with the sorted keys Y; ids travel with the keys of X and are
// gathered from the index's id column for the keys of Y. Workgroups [0, nbx) take one 2048-key tile of X each, the others 256 keys of Y each.
//   key i of X (alive):  Z[i - (deleted indices < i) + (keys of Y < key)]      key j of Y:  Z[j + p - (deleted indices < p)],  p = keys of X < key
// A tile's deleted indices and inserted keys are a window of dx / Y that its first and last key bound; up to VIEW_WIN of each are searched in LDS
// (a patch of 1M keys into 10^8 puts ~20 into a tile), beyond that in global memory. Loads of X are issued eight deep per lane before the first use.
template <class T>
__global__ __launch_bounds__(256) void k_view_merge(ViewRun<T> X, uint32_t nx, const uint32_t* __restrict__ dx, uint32_t ndx, const T* __restrict__ yv, const uint32_t* __restrict__ yp, uint32_t ny,
                                                    const uint64_t* __restrict__ ix_ids, ViewRun<T> Z, uint32_t nbx) {
  __shared__ uint32_t win[4];
  __shared__ uint32_t s_dx[VIEW_WIN];
  __shared__ T s_yv[VIEW_WIN];
  __shared__ uint32_t s_yp[VIEW_WIN];
  if (blockIdx.x < nbx) {
    const uint32_t lo = blockIdx.x * VIEW_TILE, hi = lo + VIEW_TILE < nx ? lo + VIEW_TILE : nx;     // tile [lo, hi) of X, hi > lo
    constexpr uint32_t U = VIEW_TILE / 256u;
    T xv[U]; uint32_t xp[U]; uint64_t xi[U];
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t i = lo + u * 256u + threadIdx.x;
      if (i < hi) { xv[u] = __builtin_nontemporal_load(X.v + i); xp[u] = __builtin_nontemporal_load(X.p + i); xi[u] = __builtin_nontemporal_load(X.ids + i); }
    }
    {   // one wave per window end: the deleted indices in [lo, hi), the keys of Y between the tile's first and last key
      const uint32_t w = threadIdx.x >> 6;
      uint32_t b;
      if (w == 0) b = u32_lower_bound_wave(dx, 0, ndx, lo);
      else if (w == 1) b = u32_lower_bound_wave(dx, 0, ndx, hi);
      else if (w == 2) b = (uint32_t)vk_bound_wave<T>(yv, yp, 0, ny, X.v[lo], X.p[lo]);
      else b = (uint32_t)vk_bound_wave<T>(yv, yp, 0, ny, X.v[hi - 1], X.p[hi - 1]);
      if ((threadIdx.x & 63u) == 0) win[w] = b;
    }
    __syncthreads();
    const uint32_t d0 = win[0], d1 = win[1], y0 = win[2], y1 = win[3];
    const bool in_lds = d1 - d0 <= VIEW_WIN && y1 - y0 <= VIEW_WIN;       // (uniform)
    if (in_lds) {
      for (uint32_t k = threadIdx.x; k < d1 - d0; k += 256u) s_dx[k] = dx[d0 + k];
      for (uint32_t k = threadIdx.x; k < y1 - y0; k += 256u) { s_yv[k] = yv[y0 + k]; s_yp[k] = yp[y0 + k]; }
      __syncthreads();
    }
#pragma unroll
    for (uint32_t u = 0; u < U; u++) {
      const uint32_t i = lo + u * 256u + threadIdx.x;
      if (i >= hi) continue;
      uint32_t r, y; bool gone;
      if (in_lds) {
        const uint32_t rl = u32_lower_bound(s_dx, 0, d1 - d0, i);
        gone = rl < d1 - d0 && s_dx[rl] == i;
        r = d0 + rl;
        y = y0 + (uint32_t)vk_bound<T>(s_yv, s_yp, 0, y1 - y0, xv[u], xp[u]);
      } else {
        r = u32_lower_bound(dx, d0, d1, i);
        gone = r < d1 && dx[r] == i;
        y = (uint32_t)vk_bound<T>(yv, yp, y0, y1, xv[u], xp[u]);
      }
      if (gone) continue;
      const uint64_t o = (uint64_t)i - r + y;
      Z.v[o] = xv[u]; Z.p[o] = xp[u]; Z.ids[o] = xi[u];
    }
    return;
  }
  // keys of Y: 256 consecutive ones; their places in X lie between the places of the first and the last
  const uint32_t j0 = (blockIdx.x - nbx) * 256u;
  if (j0 >= ny) return;
  const uint32_t jl = j0 + 255u < ny ? j0 + 255u : ny - 1u;
  {
    const uint32_t w = threadIdx.x >> 6;
    if (w < 2) {
      const uint32_t k = w == 0 ? j0 : jl;
      const uint32_t b = (uint32_t)vk_bound_wave<T>(X.v, X.p, 0, nx, yv[k], yp[k]);
      if ((threadIdx.x & 63u) == 0) win[w] = b;
    }
  }
  __syncthreads();
  const uint32_t j = j0 + threadIdx.x;
  if (j >= ny) return;
  const T kv = yv[j]; const uint32_t kp = yp[j];
  const uint32_t p = (uint32_t)vk_bound<T>(X.v, X.p, win[0], win[1], kv, kp);
  const uint32_t r = u32_lower_bound(dx, 0, ndx, p);
  const uint64_t o = (uint64_t)j + p - r;
  Z.v[o] = kv; Z.p[o] = kp; Z.ids[o] = ix_ids[kp];
}

}  // namespace bmx
