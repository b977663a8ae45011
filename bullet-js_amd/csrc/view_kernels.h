// view_kernels.h — keeping the value-ordered view of an index CURRENT under writes (round 5), gfx950, hand-written.
//
// The reference maintains its value-keyed index on every write: _updateIndices moves the path from the bucket of the old value to the
// bucket of the new one (src/bullet-query.js:139-176 -> _removeFromIndex / _addToIndex :82-118). The device's value-ordered view
// (scan_kernels.h: the index columns sorted by (value, position)) used to be thrown away by any write to its field and sorted again from
// scratch. Here the merges' change log is turned into a CHANGE RUN — for every index row whose value really changed: the key it had,
// (old value, position), and the key it has now; for every appended row: its key — the run is sorted by these kernels (an LDS bitonic
// sort of 4096-key tiles, three stages per LDS round trip, + merge-path passes: the run is at most a few million keys and lives in L2 / the
// Infinity Cache). What happens to the sorted run (patch_view_t in bmx.hip):
//   * small views, or a run larger than a sixteenth of the view: it is merged into the view in ONE streaming pass (k_view_merge): every key of the
//     old view that is not deleted moves to (its index - deleted keys in front of it + inserted keys in front of it), every inserted key to
//     (its index + surviving keys in front of it);
//   * large views: it joins the view's PENDING patch (deleted keys PD, inserted keys PI; k_view_flag_in, the flag selects, k_view_merge2), queries
//     answer from main - PD + PI (k_ordered_bounds_p / k_ordered_copy_p / k_ordered_filter_p), and the streaming pass runs behind an answer once
//     the patch has grown.
// No atomics, no ordering between workgroups: every output element is computed and written by exactly one thread from ranks in sorted, read-only inputs.
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
  uint32_t dbg = 0;        // measurement switches of bench_micro/view_merge_micro.hip (1: no merge-path search, 2: no rank search in LDS): wrong output, never set by the engine
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

// One tile of VIEW_SORT_TILE keys sorted in LDS: a bitonic network whose compare-exchange stages are done THREE AT A TIME in registers. A thread takes 8 keys
// whose indices differ in three consecutive bits (stride s, 2s, 4s), runs the stages j = 4s, 2s, s on them and puts them back: one LDS round trip and one
// barrier per three stages — 28 rounds for 4096 keys instead of the 78 of the stage-per-round form (2048-key tiles in that form: 68 us per 2M keys; this one
// sorts twice the run length, which also saves a merge pass). A ragged last tile is padded with keys behind every real one; only real keys are written back.
constexpr uint32_t VIEW_SORT_TILE = 4096, VIEW_SORT_THREADS = VIEW_SORT_TILE / 8;
// Where key i of the tile lives in LDS. With A, B, C the three 3-bit fields of i from the bottom, the bank (low 6 bits) is (A ^ C, B ^ C): whichever two of the
// three fields a wave's lanes run through in a round (stride 1: B, C; stride 8: A, C; stride 64 and up: A, B), the 64 lanes fall on 64 different banks. The plain
// layout serialises the stride-1 and stride-8 rounds eight ways (21 of the 30 rounds).
__device__ __forceinline__ uint32_t vk_sw(uint32_t i) { return i ^ (((i >> 6) & 7u) * 9u); }
template <class T>
__device__ __forceinline__ void vk_cx(T& av, uint32_t& ap, T& bv, uint32_t& bp, bool up) {       // afterwards (a, b) ascending when up, descending otherwise
  const bool sw = vk_less<T>(bv, bp, av, ap) == up;
  const T tv = sw ? bv : av; const uint32_t tp = sw ? bp : ap;
  bv = sw ? av : bv; bp = sw ? ap : bp; av = tv; ap = tp;
}
// NST stages (j = s << (NST - 1) ... s) of the merge step that builds sorted stretches of k keys; every thread 8 keys = 8 >> NST groups of 1 << NST
template <class T, int NST>
__device__ __forceinline__ void vk_round(T* sv, uint32_t* sp, uint32_t s, uint32_t k) {
  constexpr uint32_t G = 1u << NST, NG = 8u / G;
#pragma unroll
  for (uint32_t q = 0; q < NG; q++) {
    const uint32_t t = q * VIEW_SORT_THREADS + threadIdx.x;                    // group number: consecutive lanes on consecutive groups (no bank conflicts)
    const uint32_t base = (t & (s - 1u)) | ((t & ~(s - 1u)) << NST);           // its lowest key: the NST bits above log2(s) are the group's own
    const bool up = (base & k) == 0;                                           // (bit k lies above the group's bits: one direction for the whole group)
    T v[G]; uint32_t p[G];
#pragma unroll
    for (uint32_t r = 0; r < G; r++) { v[r] = sv[vk_sw(base + r * s)]; p[r] = sp[vk_sw(base + r * s)]; }
#pragma unroll
    for (uint32_t h = G >> 1; h > 0; h >>= 1) {
#pragma unroll
      for (uint32_t r = 0; r < G; r++) if ((r & h) == 0) vk_cx<T>(v[r], p[r], v[r | h], p[r | h], up);
    }
#pragma unroll
    for (uint32_t r = 0; r < G; r++) { sv[vk_sw(base + r * s)] = v[r]; sp[vk_sw(base + r * s)] = p[r]; }
  }
  __syncthreads();
}
template <class T>
__global__ __launch_bounds__(VIEW_SORT_THREADS) void k_view_tile_sort(const T* __restrict__ vin, const uint32_t* __restrict__ pin, T* __restrict__ vout, uint32_t* __restrict__ pout, ViewSegs S) {
  constexpr T VMAX = sizeof(T) == 4 ? (T)INT32_MAX : (T)INT64_MAX;
  __shared__ T sv[VIEW_SORT_TILE];
  __shared__ uint32_t sp[VIEW_SORT_TILE];
  const uint32_t sg = S.seg_of(blockIdx.x);
  const uint32_t t0 = (blockIdx.x - S.blk0[sg]) * VIEW_SORT_TILE;     // first key of the tile inside its segment
  const uint32_t len = S.len[sg], base = S.base[sg];
#pragma unroll
  for (uint32_t u = 0; u < 8u; u++) {
    const uint32_t e = u * VIEW_SORT_THREADS + threadIdx.x;
    const bool ok = t0 + e < len;
    sv[vk_sw(e)] = ok ? vin[base + t0 + e] : VMAX; sp[vk_sw(e)] = ok ? pin[base + t0 + e] : 0xFFFFFFFFu;
  }
  __syncthreads();
  {   // the merge steps k = 2, 4, 8 touch only the 8 consecutive keys of a thread: one round trip for their six stages
    const uint32_t b8 = threadIdx.x * 8u;
    T v[8]; uint32_t p[8];
#pragma unroll
    for (uint32_t r = 0; r < 8u; r++) { v[r] = sv[vk_sw(b8 + r)]; p[r] = sp[vk_sw(b8 + r)]; }
#pragma unroll
    for (uint32_t r = 0; r < 8u; r += 2u) vk_cx<T>(v[r], p[r], v[r + 1u], p[r + 1u], (r & 2u) == 0);                          // k = 2
#pragma unroll
    for (uint32_t r = 0; r < 8u; r++) if ((r & 2u) == 0) vk_cx<T>(v[r], p[r], v[r + 2u], p[r + 2u], (r & 4u) == 0);            // k = 4, j = 2
#pragma unroll
    for (uint32_t r = 0; r < 8u; r += 2u) vk_cx<T>(v[r], p[r], v[r + 1u], p[r + 1u], (r & 4u) == 0);                          // k = 4, j = 1
    const bool up8 = (b8 & 8u) == 0;
#pragma unroll
    for (uint32_t h = 4u; h > 0; h >>= 1) {
#pragma unroll
      for (uint32_t r = 0; r < 8u; r++) if ((r & h) == 0) vk_cx<T>(v[r], p[r], v[r | h], p[r | h], up8);                       // k = 8
    }
#pragma unroll
    for (uint32_t r = 0; r < 8u; r++) { sv[vk_sw(b8 + r)] = v[r]; sp[vk_sw(b8 + r)] = p[r]; }
    __syncthreads();
  }
  for (uint32_t k = 16, n = 4; k <= VIEW_SORT_TILE; k <<= 1, n++) {    // n = log2(k) stages: j = k/2 ... 1
    uint32_t left = n, jt = k >> 1;
    while (left) {
      const uint32_t g = left % 3u ? left % 3u : 3u;                   // the odd group first, then threes
      const uint32_t s = jt >> (g - 1u);
      if (g == 3u) vk_round<T, 3>(sv, sp, s, k); else if (g == 2u) vk_round<T, 2>(sv, sp, s, k); else vk_round<T, 1>(sv, sp, s, k);
      jt >>= g; left -= g;
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < 8u; u++) {
    const uint32_t e = u * VIEW_SORT_THREADS + threadIdx.x;
    if (t0 + e < len) { vout[base + t0 + e] = sv[vk_sw(e)]; pout[base + t0 + e] = sp[vk_sw(e)]; }
  }
}

// One pass of the merge sort: inside every segment, sorted runs of L keys (L a multiple of 1024) are merged pairwise into runs of 2L — merge path: a
// workgroup OWNS 1024 consecutive keys of the OUTPUT. Two waves find, 64-ary, where the diagonals through its first and last output key cut the two input
// runs (a few rounds of dependent reads instead of a binary search's twenty); the two input pieces (1024 keys together) are staged in LDS, every key finds its
// rank in the other piece there, and the workgroup writes its own contiguous stretch. (The first version let every INPUT key compute its output place and
// store it there: two workgroups on different XCDs then fill every 32-byte sector of the output together, the memory side merges byte-masked partial writes,
// and a pass over 1.9M keys took 27 us however the searches were arranged.)
constexpr uint32_t VIEW_PASS_KEYS = 1024;
constexpr uint32_t MP_WINDOW = 64u * 63u;          // keys spanned by the first (guessed) round of a merge-path search
// keys of run A that the merged sequence holds in front of diagonal d (A: la keys at a0, B: lb keys at b0; ties: A first). Whole wave, same arguments.
template <class T>
__device__ __forceinline__ uint32_t merge_path_wave(const T* __restrict__ V, const uint32_t* __restrict__ P, uint32_t a0, uint32_t la, uint32_t b0, uint32_t lb, uint32_t d) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t Lo = d > lb ? d - lb : 0u, Hi = d < la ? d : la;          // the answer i lies in [Lo, Hi]; predicate pred(i) = "A[i] is in front of the diagonal" = !(B[d-1-i] < A[i]), true for i < answer
  if (Hi - Lo > 2u * MP_WINDOW) {
    // a first round around the place the cut has when both runs are samples of one distribution, d * la / (la + lb): a window of 64 probes 63 keys apart. The cut is
    // usually inside (the runs of a sort are, up to ~sqrt(L)) and one more round finishes; when it is not, the window has still cut the range on one side.
    const uint32_t g = (uint32_t)(((uint64_t)d * la) / (la + lb));
    uint32_t w0 = g > MP_WINDOW / 2u ? g - MP_WINDOW / 2u : 0u;
    if (w0 < Lo) w0 = Lo;
    if (w0 + MP_WINDOW > Hi) w0 = Hi - MP_WINDOW;                      // (Hi - Lo > 2 windows: stays >= Lo)
    const uint32_t i = w0 + lane * 63u;                                // < Hi
    const bool pr = !vk_less<T>(V[b0 + d - 1u - i], P[b0 + d - 1u - i], V[a0 + i], P[a0 + i]);
    const uint32_t tcount = (uint32_t)__popcll(__ballot(pr));
    if (tcount == 0) Hi = w0;                                          // the cut is at or in front of the window's first probe
    else { Lo = w0 + (tcount - 1u) * 63u + 1u; if (tcount < 64u) Hi = w0 + tcount * 63u; }
  }
  while (Hi - Lo > 63u) {
    const uint32_t step = (Hi - Lo + 63u) / 64u;
    const uint32_t i = Lo + lane * step;                              // probes Lo, Lo + step, ...; pred is monotone (true ... true false ... false)
    bool pr = false;
    if (i < Hi) pr = !vk_less<T>(V[b0 + d - 1u - i], P[b0 + d - 1u - i], V[a0 + i], P[a0 + i]);
    const uint32_t tcount = (uint32_t)__popcll(__ballot(pr));         // probes 0 .. tcount-1 true
    const uint32_t nLo = tcount ? Lo + (tcount - 1u) * step + 1u : Lo;
    const uint32_t nHi = Lo + tcount * step < Hi ? Lo + tcount * step : Hi;
    Lo = nLo; Hi = tcount == 0 ? Lo : nHi;
  }
  bool pr = false;
  const uint32_t i = Lo + lane;
  if (i < Hi) pr = !vk_less<T>(V[b0 + d - 1u - i], P[b0 + d - 1u - i], V[a0 + i], P[a0 + i]);
  return Lo + (uint32_t)__popcll(__ballot(pr));
}
template <class T>
__global__ __launch_bounds__(256) void k_view_merge_pass(const T* __restrict__ vin, const uint32_t* __restrict__ pin, T* __restrict__ vout, uint32_t* __restrict__ pout, ViewSegs S, uint32_t L) {
  __shared__ uint32_t cut[2];
  __shared__ T sv[VIEW_PASS_KEYS];
  __shared__ uint32_t sp[VIEW_PASS_KEYS];
  const uint32_t s = S.seg_of(blockIdx.x);
  const uint32_t len = S.len[s], base = S.base[s];
  const uint32_t o0 = (blockIdx.x - S.blk0[s]) * VIEW_PASS_KEYS;     // first output key of this workgroup inside the segment
  if (o0 >= len) return;
  const uint32_t o1 = o0 + VIEW_PASS_KEYS < len ? o0 + VIEW_PASS_KEYS : len;
  const uint32_t pb = o0 / (2u * L) * (2u * L);                        // the pair of runs this stretch of the output belongs to: A = [pb, pb + la), B = [pb + la, pb + la + lb)
  const uint32_t la = pb + L < len ? L : len - pb, lb = pb + 2u * L <= len ? L : (pb + L < len ? len - pb - L : 0u);
  const T* V = vin + base; const uint32_t* P = pin + base;
  const uint32_t w = threadIdx.x >> 6;
  if (w < 2) {
    const uint32_t d = (w == 0 ? o0 : o1) - pb;
    const uint32_t i = lb == 0 ? d : ((S.dbg & 1u) ? (uint32_t)((uint64_t)d * la / (la + lb)) : merge_path_wave<T>(V, P, pb, la, pb + la, lb, d));
    if ((threadIdx.x & 63u) == 0) cut[w] = i;
  }
  __syncthreads();
  const uint32_t i0 = cut[0], i1 = cut[1], j0 = (o0 - pb) - i0, j1 = (o1 - pb) - i1;
  const uint32_t na = i1 - i0, nb = j1 - j0;                            // na + nb = o1 - o0
  for (uint32_t k = threadIdx.x; k < na + nb; k += 256u) { const uint32_t src = k < na ? pb + i0 + k : pb + la + j0 + (k - na); sv[k] = V[src]; sp[k] = P[src]; }
  __syncthreads();
  // every thread merges FOUR consecutive keys of the output: one merge-path search in LDS for its diagonal (ten steps), then four sequential picks — instead of
  // four rank searches (bench_micro/view_merge_micro.hip: the rank searches were 9 of a pass's 24-34 us)
  const uint32_t tot = na + nb, dd = threadIdx.x * 4u;
  if (dd < tot) {
    uint32_t lo = dd > nb ? dd - nb : 0u, hi = dd < na ? dd : na;     // keys of A among the first dd of the merged piece
    if (S.dbg & 2u) lo = hi = (uint32_t)((uint64_t)dd * na / tot);
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const uint32_t bj = na + (dd - 1u - mid);
      if (!vk_less<T>(sv[bj], sp[bj], sv[mid], sp[mid])) lo = mid + 1u; else hi = mid;      // A[mid] is in front of the diagonal (ties: A first)
    }
    uint32_t ia = lo, ib = na + (dd - lo);
    T ov[4]; uint32_t op[4];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      if (dd + k >= tot) break;
      const bool hasA = ia < na, hasB = ib < tot;
      const T av = hasA ? sv[ia] : (T)0, bv = hasB ? sv[ib] : (T)0; const uint32_t ap = hasA ? sp[ia] : 0u, bp = hasB ? sp[ib] : 0u;
      const bool takeA = hasA && (!hasB || !vk_less<T>(bv, bp, av, ap));
      ov[k] = takeA ? av : bv; op[k] = takeA ? ap : bp; ia += takeA ? 1u : 0u; ib += takeA ? 0u : 1u; cnt++;
    }
    T* dv = vout + base + o0 + dd; uint32_t* dp = pout + base + o0 + dd;
    if (cnt == 4u) {
      typedef T tvec __attribute__((ext_vector_type(4)));
      typedef uint32_t uvec __attribute__((ext_vector_type(4)));
      tvec v4; uvec p4;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) { v4[k] = ov[k]; p4[k] = op[k]; }
      __builtin_memcpy(dv, &v4, sizeof(v4)); __builtin_memcpy(dp, &p4, sizeof(p4));          // (4-byte aligned destination: unaligned multi-dword stores)
    } else for (uint32_t k = 0; k < cnt; k++) { dv[k] = ov[k]; dp[k] = op[k]; }
  }
}

// A sample of the view: the key every tile of 2048 starts with (48 828 keys for 10^8 rows: L2-resident).
template <class T>
__global__ __launch_bounds__(256) void k_view_sample(const T* __restrict__ xv, const uint32_t* __restrict__ xp, uint32_t ntiles, T* __restrict__ sv, uint32_t* __restrict__ sp) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t < ntiles) { sv[t] = xv[(uint64_t)t * VIEW_TILE]; sp[t] = xp[(uint64_t)t * VIEW_TILE]; }
}
// per tile of the view: how many deleted keys (d0) and how many inserted keys (y0) sort in front of its first key; entry [ntiles] = the totals. Tile t's
// windows are [d0[t], d0[t + 1]) of D and [y0[t], y0[t + 1]) of Y: the deleted keys ARE keys of the tile, the inserted ones fall between its keys or behind its last.
template <class T>
__global__ __launch_bounds__(256) void k_view_tile_offsets(const T* __restrict__ sv, const uint32_t* __restrict__ sp, uint32_t ntiles, const T* __restrict__ dv, const uint32_t* __restrict__ dp, uint32_t nd,
                                                           const T* __restrict__ yv, const uint32_t* __restrict__ yp, uint32_t ny, uint32_t* __restrict__ d0, uint32_t* __restrict__ y0) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) { d0[t] = nd; y0[t] = ny; return; }
  if (t == 0) { d0[0] = 0; y0[0] = 0; return; }                        // (an inserted key in front of the view's first key belongs to tile 0 too)
  d0[t] = (uint32_t)vk_bound<T>(dv, dp, 0, nd, sv[t], sp[t]);
  y0[t] = (uint32_t)vk_bound<T>(yv, yp, 0, ny, sv[t], sp[t]);
}

template <class T>
struct ViewRun { T* v; uint32_t* p; uint64_t* ids; };
template <class V, class P>
__device__ __forceinline__ void st_vec(P* dst, const V& v) { __builtin_memcpy(dst, &v, sizeof(V)); }
__device__ __forceinline__ int32_t rdlane(int32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, (int)l); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ int64_t rdlane(int64_t v, uint32_t l) {
  return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, (int)l));
}

// The streaming merge: Z = (X without the sorted keys D, which are keys of X) merged with the sorted keys Y; ids travel with the keys of X and are gathered from
// the index's id column for the keys of Y. One workgroup per 2048-key tile of X; it also places the keys of Y that fall between its keys or behind its last.
//   key i of X (alive):   Z[i - (keys of D < key) + (keys of Y < key)]
//   key j of Y:           right behind the last key of X in front of it:  Z[j + (i + 1) - (keys of D <= key_i)],  i = that key's index
// A lane owns FOUR consecutive keys of X: 16-byte loads of values and positions, 2 x 16 bytes of ids, all issued before the first use (two groups per lane).
// A tile's deleted and inserted keys are the windows [d0[t], d0[t + 1]) of D and [y0[t], y0[t + 1]) of Y (k_view_tile_offsets), staged in LDS up to VIEW_WIN
// of each; nine threads rank the tile's 256-key chunk boundaries in them. A wave then takes ITS chunk's window entries into registers — one per lane; a patch of 1M
// keys puts ~5 deleted and ~3 inserted keys into a chunk — and every lane counts, entry by entry (readlane: scalar broadcasts, no memory, no divergence), how many
// sort in front of each of its four keys and whether one IS its key. A group with no deleted key and one rank for all four moves as 16-byte stores (the destination is
// only 4- / 8-byte aligned: gfx950 takes unaligned multi-dword stores), the others key by key. Measured on 10^8 rows (bench_micro/view_merge_micro.hip,
// profiles/r05_view_merge_micro.log): a plain 16-byte copy of the three columns 543 us = 5.89 TB/s; this kernel with an empty patch 552 us.
// Earlier forms, same box class: one key per lane with two LDS window searches per key 931 us; lanes WALKING from the chunk's rank to their own (dependent LDS
// reads, the wave waits for its slowest lane) 847 us + 270 us for the keys of Y placed by workgroups of their own at the end of the grid.
// *err is set when the tile's deleted keys were not all found among its keys (the view and the index columns have drifted apart: the caller sorts from scratch).
template <class T, bool HAS_IDS = true>
__global__ __launch_bounds__(256) void k_view_merge(ViewRun<T> X, uint32_t nx, const T* __restrict__ dv, const uint32_t* __restrict__ dp, const T* __restrict__ yv, const uint32_t* __restrict__ yp,
                                                    const uint64_t* __restrict__ ix_ids, ViewRun<T> Z, const uint32_t* __restrict__ d0s, const uint32_t* __restrict__ y0s, uint32_t* __restrict__ err) {
  typedef T tvec __attribute__((ext_vector_type(4)));
  typedef uint32_t uvec __attribute__((ext_vector_type(4)));
  typedef unsigned long long lvec __attribute__((ext_vector_type(2)));
  constexpr uint32_t CH = 256u, NCH = VIEW_TILE / CH;             // a chunk = the 256 consecutive keys one wave handles per iteration
  constexpr uint32_t IT = VIEW_TILE / 1024u;
  __shared__ T s_dv[VIEW_WIN];
  __shared__ uint32_t s_dp[VIEW_WIN];
  __shared__ T s_yv[VIEW_WIN];
  __shared__ uint32_t s_yp[VIEW_WIN];
  __shared__ uint32_t rB[NCH + 1], yB[NCH + 1];
  __shared__ uint32_t n_gone;
  const uint32_t t = blockIdx.x;
  const uint32_t lo = t * VIEW_TILE, hi = lo + VIEW_TILE < nx ? lo + VIEW_TILE : nx;     // tile [lo, hi) of X, hi > lo
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  tvec xv[IT]; uvec xp[IT]; lvec xa[IT], xb[IT];
#pragma unroll
  for (uint32_t it = 0; it < IT; it++) {
    const uint32_t e = lo + it * 1024u + w * CH + lane * 4u;        // this lane's group [e, e + 4)
    if (e + 4u <= hi) {
      xv[it] = __builtin_nontemporal_load(reinterpret_cast<const tvec*>(X.v + e)); xp[it] = __builtin_nontemporal_load(reinterpret_cast<const uvec*>(X.p + e));
      if (HAS_IDS) { xa[it] = __builtin_nontemporal_load(reinterpret_cast<const lvec*>(X.ids + e)); xb[it] = __builtin_nontemporal_load(reinterpret_cast<const lvec*>(X.ids + e + 2)); }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) { const bool ok = e + k < hi; xv[it][k] = ok ? X.v[e + k] : (T)0; xp[it][k] = ok ? X.p[e + k] : 0u; const unsigned long long id = (HAS_IDS && ok) ? X.ids[e + k] : 0ull; if (k < 2) xa[it][k] = id; else xb[it][k - 2] = id; }
    }
  }
  const uint32_t d0 = d0s[t], d1 = d0s[t + 1], y0 = y0s[t], y1 = y0s[t + 1];
  const uint32_t nd = d1 - d0, nyw = y1 - y0;
  if (nd == 0 && nyw == 0) {                                           // (uniform) nothing of the patch touches this tile: it moves as it is
#pragma unroll
    for (uint32_t it = 0; it < IT; it++) {
      const uint32_t e = lo + it * 1024u + w * CH + lane * 4u;
      const uint64_t o = (uint64_t)e - d0 + y0;
      if (e + 4u <= hi) { st_vec(Z.v + o, xv[it]); st_vec(Z.p + o, xp[it]); if (HAS_IDS) { st_vec(Z.ids + o, xa[it]); st_vec(Z.ids + o + 2, xb[it]); } }
      else for (uint32_t k = 0; k < 4 && e + k < hi; k++) { Z.v[o + k] = xv[it][k]; Z.p[o + k] = xp[it][k]; if (HAS_IDS) Z.ids[o + k] = k < 2 ? xa[it][k] : xb[it][k - 2]; }
    }
    return;
  }
  const bool in_lds = nd <= VIEW_WIN && nyw <= VIEW_WIN;              // (uniform)
  if (threadIdx.x == 0) n_gone = 0;
  if (in_lds) {
    for (uint32_t k = threadIdx.x; k < nd; k += 256u) { s_dv[k] = dv[d0 + k]; s_dp[k] = dp[d0 + k]; }
    for (uint32_t k = threadIdx.x; k < nyw; k += 256u) { s_yv[k] = yv[y0 + k]; s_yp[k] = yp[y0 + k]; }
  }
  __syncthreads();
  // the windows, wherever they are: entry k of the tile's deleted / inserted keys
  const T* dwv = in_lds ? s_dv : dv + d0; const uint32_t* dwp = in_lds ? s_dp : dp + d0;
  const T* ywv = in_lds ? s_yv : yv + y0; const uint32_t* ywp = in_lds ? s_yp : yp + y0;
  if (threadIdx.x <= NCH) {      // ranks (inside the windows) of the chunk boundaries lo, lo + 256, ..., the tile's end
    const uint32_t b = lo + threadIdx.x * CH;
    uint32_t r = nd, y = nyw;
    if (threadIdx.x < NCH && b < hi) { const T bv = X.v[b]; const uint32_t bp = X.p[b]; r = (uint32_t)vk_bound<T>(dwv, dwp, 0, nd, bv, bp); y = (uint32_t)vk_bound<T>(ywv, ywp, 0, nyw, bv, bp); }
    if (threadIdx.x == 0) y = 0;                                       // (inserted keys in front of the tile's first key — only tile 0 has any — are placed by its first lane, below)
    rB[threadIdx.x] = r; yB[threadIdx.x] = y;
  }
  __syncthreads();
  uint32_t gone_cnt = 0;
#pragma unroll
  for (uint32_t it = 0; it < IT; it++) {
    const uint32_t c = it * 4u + w;
    const uint32_t e = lo + c * CH + lane * 4u;
    const uint32_t ra = rB[c], rb = rB[c + 1], ya = yB[c], yb = yB[c + 1];
    const uint32_t nD = rb - ra, nY = yb - ya;                          // (wave-uniform) this chunk's deleted / inserted keys
    uint32_t r[4], y[4]; bool gone[4];
    // (1) the ranks of the group's FIRST key: the chunk's entries, 64 at a time one per lane, broadcast one by one (readlane: no memory, no divergence) and compared
    // with ONE key per lane. (The first form compared every entry with all four keys: ~35 vector instructions per entry and wave, and a rewrite of main with a
    // patch of 7.5 % of the rows took 1.5 ms against 0.75 at 3 %.)
    const T k0v = xv[it][0]; const uint32_t k0p = xp[it][0];
    uint32_t r0 = ra, y0r = ya;
    for (uint32_t b0 = 0; b0 < nD; b0 += 64u) {
      T ev = (T)0; uint32_t ep = 0;
      if (b0 + lane < nD) { ev = dwv[ra + b0 + lane]; ep = dwp[ra + b0 + lane]; }
      const uint32_t cnt = nD - b0 < 64u ? nD - b0 : 64u;
      for (uint32_t j = 0; j < cnt; j++) { const T jv = rdlane(ev, j); const uint32_t jp = rdlane(ep, j); r0 += vk_less<T>(jv, jp, k0v, k0p) ? 1u : 0u; }
    }
    for (uint32_t b0 = 0; b0 < nY; b0 += 64u) {
      T ev = (T)0; uint32_t ep = 0;
      if (b0 + lane < nY) { ev = ywv[ya + b0 + lane]; ep = ywp[ya + b0 + lane]; }
      const uint32_t cnt = nY - b0 < 64u ? nY - b0 : 64u;
      for (uint32_t j = 0; j < cnt; j++) { const T jv = rdlane(ev, j); const uint32_t jp = rdlane(ep, j); y0r += vk_less<T>(jv, jp, k0v, k0p) ? 1u : 0u; }
    }
    if (e >= hi) { r0 = rb; y0r = yb; }                                  // behind the tile's end: what the last real group sees behind itself
    // (2) the next group's ranks = the ranks just behind this group: the entries in [r0, rn) / [y0r, ynx) are the ones that touch THIS group's keys
    uint32_t rn = (uint32_t)__shfl_down((int)r0, 1), ynx = (uint32_t)__shfl_down((int)y0r, 1);
    if (lane == 63u) { rn = rb; ynx = yb; }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { r[k] = r0; y[k] = y0r; gone[k] = false; }
    if (rn != r0 || ynx != y0r) {                                         // (per lane; most groups skip this) a walk over the few entries inside the group
      uint32_t rk = r0, yk = y0r;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const T kv = xv[it][k]; const uint32_t kp = xp[it][k];
        while (rk < rn && vk_less<T>(dwv[rk], dwp[rk], kv, kp)) rk++;
        gone[k] = rk < rn && dwv[rk] == kv && dwp[rk] == kp;
        while (yk < ynx && vk_less<T>(ywv[yk], ywp[yk], kv, kp)) yk++;
        r[k] = rk; y[k] = yk;
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) if (e + k >= hi) { r[k] = rb; y[k] = yb; gone[k] = false; }     // keys behind the tile's end inside a group that straddles it
    // the inserted keys BEHIND key k (in front of the next key of X): ranks [y[k], yn[k])
    const uint32_t yn[4] = {y[1], y[2], y[3], ynx};
    const uint64_t ob = (uint64_t)y0 - d0;                               // output index of key i = i + ob - r + y
    const bool clean = !gone[0] && !gone[1] && !gone[2] && !gone[3] && r[3] == r[0] && y[3] == y[0] && e + 4u <= hi;
    if (e < hi) {
      if (clean) { const uint64_t o = e + ob - r[0] + y[0]; st_vec(Z.v + o, xv[it]); st_vec(Z.p + o, xp[it]); if (HAS_IDS) { st_vec(Z.ids + o, xa[it]); st_vec(Z.ids + o + 2, xb[it]); } }
      else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          if (e + k < hi && !gone[k]) { const uint64_t o = (uint64_t)(e + k) + ob - r[k] + y[k]; Z.v[o] = xv[it][k]; Z.p[o] = xp[it][k]; if (HAS_IDS) Z.ids[o] = k < 2 ? xa[it][k] : xb[it][k - 2]; }
          gone_cnt += gone[k] ? 1u : 0u;
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        if (e + k >= hi) break;
        for (uint32_t j = y[k]; j < yn[k]; j++) {                          // (rare: 1 % of the keys have one behind them)
          const uint64_t o = (uint64_t)(e + k + 1u) + ob - (r[k] + (gone[k] ? 1u : 0u)) + j;
          const uint32_t jp = ywp[j];
          Z.v[o] = ywv[j]; Z.p[o] = jp; if (HAS_IDS) Z.ids[o] = ix_ids[jp];
        }
      }
      if (e == lo) for (uint32_t j = 0; j < y[0]; j++) {                 // tile 0 only: inserted keys in front of the view's first key
        const uint32_t jp = ywp[j];
        Z.v[j] = ywv[j]; Z.p[j] = jp; if (HAS_IDS) Z.ids[j] = ix_ids[jp];
      }
    }
  }
  // every deleted key of the tile's window must have been met
  for (int d = 32; d >= 1; d >>= 1) gone_cnt += (uint32_t)__shfl_xor((int)gone_cnt, d);
  if (lane == 0 && gone_cnt) atomicAdd(&n_gone, gone_cnt);
  __syncthreads();
  if (threadIdx.x == 0 && n_gone != nd) *err = 1u;
}


// ---- merging two sorted runs of similar size (the pending patch's runs with a refresh's change run): merge path over the OUTPUT, like a pass of the sort ----
// keys of run A that the merged sequence holds in front of diagonal d (ties: A first). Whole wave, same arguments.
template <class T>
__device__ __forceinline__ uint32_t merge_path_wave2(const T* __restrict__ av, const uint32_t* __restrict__ ap, uint32_t la, const T* __restrict__ bv, const uint32_t* __restrict__ bp, uint32_t lb, uint32_t d) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t Lo = d > lb ? d - lb : 0u, Hi = d < la ? d : la;
  if (Hi - Lo > 2u * MP_WINDOW) {                                       // (the window round of merge_path_wave)
    const uint32_t g = (uint32_t)(((uint64_t)d * la) / (la + lb));
    uint32_t w0 = g > MP_WINDOW / 2u ? g - MP_WINDOW / 2u : 0u;
    if (w0 < Lo) w0 = Lo;
    if (w0 + MP_WINDOW > Hi) w0 = Hi - MP_WINDOW;
    const uint32_t i = w0 + lane * 63u;
    const bool pr = !vk_less<T>(bv[d - 1u - i], bp[d - 1u - i], av[i], ap[i]);
    const uint32_t tcount = (uint32_t)__popcll(__ballot(pr));
    if (tcount == 0) Hi = w0;
    else { Lo = w0 + (tcount - 1u) * 63u + 1u; if (tcount < 64u) Hi = w0 + tcount * 63u; }
  }
  while (Hi - Lo > 63u) {
    const uint32_t step = (Hi - Lo + 63u) / 64u;
    const uint32_t i = Lo + lane * step;
    bool pr = false;
    if (i < Hi) pr = !vk_less<T>(bv[d - 1u - i], bp[d - 1u - i], av[i], ap[i]);
    const uint32_t tcount = (uint32_t)__popcll(__ballot(pr));
    const uint32_t nLo = tcount ? Lo + (tcount - 1u) * step + 1u : Lo;
    const uint32_t nHi = Lo + tcount * step < Hi ? Lo + tcount * step : Hi;
    Lo = nLo; Hi = tcount == 0 ? Lo : nHi;
  }
  bool pr = false;
  const uint32_t i = Lo + lane;
  if (i < Hi) pr = !vk_less<T>(bv[d - 1u - i], bp[d - 1u - i], av[i], ap[i]);
  return Lo + (uint32_t)__popcll(__ballot(pr));
}
// Z = A merged with B (both sorted, keys unique across them). IDS: the ids of A's keys travel with them, those of B's keys are gathered from the index's id column.
// A workgroup owns 1024 consecutive keys of Z; a thread merges four of them (k_view_merge_pass's form, two source arrays).
template <class T, bool IDS>
__global__ __launch_bounds__(256) void k_view_merge2(ViewRun<T> A, uint32_t la, const T* __restrict__ bv, const uint32_t* __restrict__ bp, uint32_t lb, const uint64_t* __restrict__ ix_ids, ViewRun<T> Z) {
  __shared__ uint32_t cut[2];
  __shared__ T sv[VIEW_PASS_KEYS];
  __shared__ uint32_t sp[VIEW_PASS_KEYS];
  const uint32_t len = la + lb;
  const uint32_t o0 = blockIdx.x * VIEW_PASS_KEYS;
  if (o0 >= len) return;
  const uint32_t o1 = o0 + VIEW_PASS_KEYS < len ? o0 + VIEW_PASS_KEYS : len;
  const uint32_t w = threadIdx.x >> 6;
  if (w < 2) {
    const uint32_t d = w == 0 ? o0 : o1;
    const uint32_t i = lb == 0 ? d : (la == 0 ? 0u : merge_path_wave2<T>(A.v, A.p, la, bv, bp, lb, d));
    if ((threadIdx.x & 63u) == 0) cut[w] = i;
  }
  __syncthreads();
  const uint32_t i0 = cut[0], i1 = cut[1], j0 = o0 - i0, j1 = o1 - i1;
  const uint32_t na = i1 - i0, nb = j1 - j0;
  for (uint32_t k = threadIdx.x; k < na + nb; k += 256u) {
    if (k < na) { sv[k] = A.v[i0 + k]; sp[k] = A.p[i0 + k]; } else { sv[k] = bv[j0 + (k - na)]; sp[k] = bp[j0 + (k - na)]; }
  }
  __syncthreads();
  const uint32_t tot = na + nb, dd = threadIdx.x * 4u;
  if (dd >= tot) return;
  uint32_t lo = dd > nb ? dd - nb : 0u, hi = dd < na ? dd : na;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    const uint32_t bj = na + (dd - 1u - mid);
    if (!vk_less<T>(sv[bj], sp[bj], sv[mid], sp[mid])) lo = mid + 1u; else hi = mid;
  }
  uint32_t ia = lo, ib = na + (dd - lo);
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    if (dd + k >= tot) break;
    const bool hasA = ia < na, hasB = ib < tot;
    const T av = hasA ? sv[ia] : (T)0, bvv = hasB ? sv[ib] : (T)0; const uint32_t ap = hasA ? sp[ia] : 0u, bpp = hasB ? sp[ib] : 0u;
    const bool takeA = hasA && (!hasB || !vk_less<T>(bvv, bpp, av, ap));
    const uint32_t o = o0 + dd + k;
    Z.v[o] = takeA ? av : bvv; Z.p[o] = takeA ? ap : bpp;
    if (IDS) Z.ids[o] = takeA ? A.ids[i0 + ia] : ix_ids[bpp];
    ia += takeA ? 1u : 0u; ib += takeA ? 0u : 1u;
  }
}
// flag[i] = 1 when sorted key i of D is one of the sorted keys of I (one binary search per key, once: the two selects that split D read the flags), and
// dead[j] = 1 for the key j of I it is (zeroed by the caller)
template <class T>
__global__ __launch_bounds__(256) void k_view_flag_in(const T* __restrict__ dv, const uint32_t* __restrict__ dp, uint32_t nd, const T* __restrict__ iv, const uint32_t* __restrict__ ip, uint32_t ni, uint8_t* __restrict__ flag,
                                                      uint8_t* __restrict__ dead) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nd) return;
  const T kv = dv[i]; const uint32_t kp = dp[i];
  const uint32_t at = (uint32_t)vk_bound<T>(iv, ip, 0, ni, kv, kp);
  const bool hit = at < ni && iv[at] == kv && ip[at] == kp;
  flag[i] = hit ? 1u : 0u;
  if (hit) dead[at] = 1u;                    // the insert it cancels (keys are unique: one writer per byte); the select that takes them out of I reads these
}
struct PredFlag { static constexpr int E = 1; const uint8_t* flag; uint32_t want; __device__ uint32_t mask(uint64_t first, uint64_t n) const { return first < n && flag[first] == want ? 1u : 0u; } };
template <class T>
struct EmitRun { const T* xv; const uint32_t* xp; const uint64_t* xi; T* ov; uint32_t* op; uint64_t* oi; __device__ void operator()(uint64_t rank, uint64_t i) const { ov[rank] = xv[i]; op[rank] = xp[i]; oi[rank] = xi[i]; } };

// ---- the view with a PENDING patch (DESIGN section 4 "kept current"): the logical view = main - PD + PI, PD = sorted keys of main that are gone, PI = sorted keys (+ ids)
// that are new; the physical merge above runs when the patch has grown, not on every refresh. ----
// ids of the inserted keys (the patch keeps them so that a query copies one run)
__global__ __launch_bounds__(256) void k_view_gather_ids(const uint32_t* __restrict__ pos, uint32_t n, const uint64_t* __restrict__ ix_ids, uint64_t* __restrict__ out) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = ix_ids[pos[i]];
}
// the deleted keys of a refresh's run split by k_view_flag_in's flags: EmitKeys copies one kind (select.h)
template <class T>
struct EmitKeys { const T* dv; const uint32_t* dp; T* ov; uint32_t* op; __device__ void operator()(uint64_t rank, uint64_t i) const { ov[rank] = dv[i]; op[rank] = dp[i]; } };

// bounds of a value range in all three sorted runs: ab[0,1] main, ab[2,3] pending deleted, ab[4,5] pending inserted; *n_out = matches of the logical view. Six waves.
template <class T>
__global__ __launch_bounds__(384) void k_ordered_bounds_p(const T* __restrict__ v, uint64_t n, const T* __restrict__ dv, uint64_t nd, const T* __restrict__ iv, uint64_t ni, T lo, T hi,
                                                          unsigned long long* __restrict__ ab, unsigned long long* __restrict__ n_out, uint32_t zero_count) {
  __shared__ unsigned long long sh[6];
  const uint32_t w = threadIdx.x >> 6;
  uint64_t r = 0;
  if (lo <= hi) {
    const T* a = w < 2 ? v : (w < 4 ? dv : iv); const uint64_t m = w < 2 ? n : (w < 4 ? nd : ni);
    r = (w & 1u) == 0 ? ordered_bound<T, false>(a, m, lo) : ordered_bound<T, true>(a, m, hi);
  }
  if ((threadIdx.x & 63u) == 0) sh[w] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 6; k += 2) { const unsigned long long x = sh[k], y = sh[k + 1] < sh[k] ? sh[k] : sh[k + 1]; ab[k] = x; ab[k + 1] = y; }
    if (n_out) *n_out = zero_count ? 0ull : (ab[1] - ab[0]) - (ab[3] - ab[2]) + (ab[5] - ab[4]);
  }
}
// The matches of the logical view: main's run [a, b) without the pending deleted keys (each key of the run looks itself up in ITS stretch [da, db) of PD: the deleted
// keys of a value range are exactly the deleted keys of that range's run), in order, then the pending inserted keys' run. out[k] for k < cap.
template <class T, class OutT>
__global__ __launch_bounds__(256) void k_ordered_copy_p(const T* __restrict__ v, const uint32_t* __restrict__ p, const OutT* __restrict__ src, const T* __restrict__ dv, const uint32_t* __restrict__ dp,
                                                        const OutT* __restrict__ isrc, const unsigned long long* __restrict__ ab, OutT* __restrict__ out, uint64_t cap) {
  const uint64_t a = ab[0], m = ab[1] - a, da = ab[2], db = ab[3], ia = ab[4], mi = ab[5] - ia;
  const uint64_t kept = m - (db - da);
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < m + mi; i += (uint64_t)gridDim.x * 256u) {
    if (i < m) {
      const T kv = v[a + i]; const uint32_t kp = p[a + i];
      const uint64_t r = da == db ? da : vk_bound<T>(dv, dp, da, db, kv, kp);
      if (r < db && dv[r] == kv && dp[r] == kp) continue;
      const uint64_t o = i - (r - da);
      if (o < cap) out[o] = src[a + i];
    } else {
      const uint64_t o = kept + (i - m);
      if (o < cap) out[o] = isrc[ia + (i - m)];
    }
  }
}
// declarative filter over the logical view's run: candidates = main's run minus the pending deleted keys, plus the pending inserted keys' run (k_ordered_filter's form)
template <class T, class PF>
__global__ __launch_bounds__(256) void k_ordered_filter_p(const T* __restrict__ v, const uint32_t* __restrict__ p, const uint64_t* __restrict__ s_ids, const T* __restrict__ dv, const uint32_t* __restrict__ dp,
                                                          const uint64_t* __restrict__ i_ids, const unsigned long long* __restrict__ ab, PF P, uint64_t* __restrict__ out, uint64_t cap,
                                                          unsigned long long* __restrict__ n_out) {
  const uint64_t a = ab[0], m = ab[1] - a, da = ab[2], db = ab[3], ia = ab[4], mi = ab[5] - ia;
  const uint64_t tot = m + mi;
  const uint64_t rounds = (tot + (uint64_t)gridDim.x * 256u - 1) / ((uint64_t)gridDim.x * 256u);
  for (uint64_t rd = 0; rd < rounds; rd++) {          // (uniform trip count per wave: the ballot below wants every lane there)
    const uint64_t i = (rd * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
    uint64_t id = 0; bool ok = false;
    if (i < m) {
      const T kv = v[a + i]; const uint32_t kp = p[a + i];
      const uint64_t r = da == db ? da : vk_bound<T>(dv, dp, da, db, kv, kp);
      if (!(r < db && dv[r] == kv && dp[r] == kp)) { id = s_ids[a + i]; ok = P.rest(id); }
    } else if (i < tot) { id = i_ids[ia + (i - m)]; ok = P.rest(id); }
    const unsigned long long bal = __ballot(ok);
    if (bal) {
      unsigned long long base = 0;
      if ((threadIdx.x & 63u) == 0) base = atomicAdd(n_out, (unsigned long long)__popcll(bal));
      base = __shfl(base, 0);
      if (ok) { const uint64_t pos = base + (uint64_t)__popcll(bal & ((1ull << (threadIdx.x & 63u)) - 1ull)); if (out && pos < cap) out[pos] = id; }
    }
  }
}

}  // namespace bmx
