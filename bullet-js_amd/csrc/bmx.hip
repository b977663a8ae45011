// bmx.hip — C ABI (include/bmx.h) over the gfx950 kernels. One context = one GPU + one HIP stream.
// No CPU fallback exists: every entry point fails with BMX_ERR_NO_DEVICE / BMX_ERR_HIP when there is no GPU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <deque>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bmx.h"
#include "merge_kernels.h"
#include "scan_kernels.h"
#include "select.h"
#include "slot.h"
#include "view_kernels.h"

namespace bmx {   // csrc/ordered_sort.hip (rocPRIM's radix sort, an object of its own)
hipError_t sort_pairs_i32(void* tmp, size_t* tmp_bytes, const int32_t* kin, int32_t lo, unsigned bits, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s);
hipError_t sort_pairs_i64(void* tmp, size_t* tmp_bytes, const int64_t* kin, int64_t lo, unsigned bits, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, hipStream_t s);
}

using namespace bmx;

namespace {

struct DevScalars {  // one small device allocation; zeroed at create
  unsigned long long row_count;
  unsigned long long n_out;        // scratch count for host-mode calls
  unsigned long long part_totals[PART_MAX_SHARDS];
  bmx_merge_stats stats;           // scratch stats for host-mode calls
  uint32_t status;
  uint32_t wide;
  unsigned long long seq_diag[3];  // k_seq_wait expiry: {sequence word address, value waited for, value last seen}
  unsigned long long chg_n[2];     // entries in the index change log: batch k reads [k&1], its compaction writes [(k+1)&1]
  unsigned long long ord_ab[6];    // value-ordered view: [first match, one past the last) of the query being answered in the view's main run, its pending deleted keys, its pending inserted keys
  unsigned long long view_cl_n[8]; // keys in the change run of maintained index k (k_ix_update capture mode: view_kernels.h)
  uint32_t view_err, view_pad;     // a deleted key was not found in the view it is patched into
  unsigned long long view_tmp[2];  // how a change run's deleted keys split: keys of main / pending inserted keys
  unsigned long long seqw[2];      // deferred compaction: [0] = number of the latest probe kernel that has started, [1] = of the latest compaction finished on the side stream
};

struct Index {
  uint32_t field = 0;
  uint64_t n = 0, cap = 0;
  uint64_t* ids = nullptr;
  int64_t* v64 = nullptr;
  int32_t* v32 = nullptr;
  bool fits32 = false;
  uint64_t version = ~0ull;  // table version it was built from
  bool has_pos = false;      // its rows' positions are in ctx->slot_pos (it can be maintained from the change log)
  uint64_t content = 0;      // counts the refreshes that really changed something in the columns (a value, a new row, a rebuild)
  // value-ordered view (bmx.h bmx_index_set_ordered): the columns once more, sorted by (value, position)
  uint32_t ordered_after = 0;     // 0 = off; N: a stale view is sorted again by the N-th query since the columns last changed
  uint32_t stale_queries = 0;     // queries answered by the column scan since the columns last CHANGED (not: since the last sort)
  uint64_t stale_content = ~0ull; // the `content` value that count belongs to: a new change starts it again
  double last_sort_us = 0;        // what the last sort of the view cost the caller (BMX_INDEX_ORDERED_AUTO weighs it against the scans it saves)
  uint64_t ord_content = ~0ull;   // `content` the view was sorted from
  uint64_t ord_n = 0, ord_cap = 0, ord_sorts = 0;
  bool ord_fits32 = false;
  void* s_val = nullptr;          // int32_t[ord_n] or int64_t[ord_n], ascending
  uint32_t* s_pos = nullptr;      // position in the index columns
  uint64_t* s_ids = nullptr;      // node id
  // kept current under writes (view_kernels.h): the second set of columns a patch merges into (allocated by the first patch, then the two sets swap),
  // the change run k_ix_update captures, and what the patches did
  void* s_val2 = nullptr; uint32_t* s_pos2 = nullptr; uint64_t* s_ids2 = nullptr; uint64_t ord_cap2 = 0;
  uint32_t* cl_pos = nullptr; int64_t* cl_old = nullptr; uint64_t cl_cap = 0;      // the captured change run, one entry per log entry (holes: POS_NONE)
  uint32_t* cl2_pos = nullptr; int64_t* cl2_old = nullptr;                          // ... and without the holes
  uint64_t ord_patches = 0, ord_patched_keys = 0, ord_merges = 0; double last_patch_us = 0;
  // the view's PENDING patch: logical view = main - pd + pi (both sorted by (value, position); pi carries ids). A refresh merges its change run into these small
  // runs; the streaming merge into main runs when they have grown past ord_n / 16 keys. Two sets: a merge writes the other one.
  void* pd_v[2] = {nullptr, nullptr}; uint32_t* pd_p[2] = {nullptr, nullptr};
  void* pi_v[2] = {nullptr, nullptr}; uint32_t* pi_p[2] = {nullptr, nullptr}; uint64_t* pi_ids[2] = {nullptr, nullptr};
  uint8_t* pi_dead = nullptr;          // one byte per key of pi: set for the inserts a refresh's deleted keys cancel (scratch of the join)
  uint64_t npd = 0, npi = 0, pend_cap = 0; int pcur = 0 /* the current set of pd */, icur = 0 /* the current set of pi */;
  // the rewrite of main (main - pd + pi -> the second set of columns) runs BEHIND the answer of the query that found it due: in flight until its event has completed
  // and its error word has been looked at; until then (main, pd, pi) go on answering
  bool rewrite_due = false, rewrite_inflight = false; uint64_t rewrite_nz = 0;
};
void free_pending(Index& ix) {
  for (int i = 0; i < 2; i++) {
    if (ix.pd_v[i]) (void)hipFree(ix.pd_v[i]); if (ix.pd_p[i]) (void)hipFree(ix.pd_p[i]);
    if (ix.pi_v[i]) (void)hipFree(ix.pi_v[i]); if (ix.pi_p[i]) (void)hipFree(ix.pi_p[i]); if (ix.pi_ids[i]) (void)hipFree(ix.pi_ids[i]);
    ix.pd_v[i] = nullptr; ix.pd_p[i] = nullptr; ix.pi_v[i] = nullptr; ix.pi_p[i] = nullptr; ix.pi_ids[i] = nullptr;
  }
  if (ix.pi_dead) (void)hipFree(ix.pi_dead);
  ix.pi_dead = nullptr;
  ix.npd = ix.npi = 0; ix.pend_cap = 0; ix.pcur = 0; ix.icur = 0; ix.rewrite_due = false; ix.rewrite_inflight = false;
}
void free_ordered_view(Index& ix) {
  free_pending(ix);
  if (ix.s_val) (void)hipFree(ix.s_val);
  if (ix.s_pos) (void)hipFree(ix.s_pos);
  if (ix.s_ids) (void)hipFree(ix.s_ids);
  if (ix.s_val2) (void)hipFree(ix.s_val2);
  if (ix.s_pos2) (void)hipFree(ix.s_pos2);
  if (ix.s_ids2) (void)hipFree(ix.s_ids2);
  if (ix.cl_pos) (void)hipFree(ix.cl_pos);
  if (ix.cl_old) (void)hipFree(ix.cl_old);
  if (ix.cl2_pos) (void)hipFree(ix.cl2_pos);
  if (ix.cl2_old) (void)hipFree(ix.cl2_old);
  ix.cl2_pos = nullptr; ix.cl2_old = nullptr;
  ix.s_val = nullptr; ix.s_pos = nullptr; ix.s_ids = nullptr; ix.ord_cap = 0; ix.ord_n = 0; ix.ord_content = ~0ull;
  ix.s_val2 = nullptr; ix.s_pos2 = nullptr; ix.s_ids2 = nullptr; ix.ord_cap2 = 0;
  ix.cl_pos = nullptr; ix.cl_old = nullptr; ix.cl_cap = 0;
}

thread_local std::string g_err;
constexpr uint32_t PROF_MAX_CALLS = 64;
constexpr int MERGE_FORCE_INTERNAL = 0x4000;   // marker carried by the INTERNAL host-batch helpers (merge_host, submit_host) for bmx_put_rows; never accepted from a caller
inline bool public_mode_ok(int insert_mode) {  // what bmx.h documents: BMX_INSERT_* plus the three optional bits
  return (insert_mode & ~(BMX_INSERT_DELTA | BMX_MERGE_UNIQUE_KEYS | BMX_MERGE_STRICT_FLAGS | BMX_MERGE_MARK_CREATED)) == 0;
}

}  // namespace

// count, stats and device status of one host batch, written by k_small_tail into mapped host memory
struct SmallOut { unsigned long long n_applied; bmx_merge_stats stats; uint32_t status; uint32_t pad; };
struct bmx_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  Slot* slots = nullptr;
  uint64_t nslots = 0, capacity_rows = 0;
  uint32_t load_pct = 50;             // maximum load factor (percent) at capacity_rows: nslots = capacity_rows * 100 / load_pct
  DevScalars* ds = nullptr;
  // per-batch workspace (grown on demand)
  uint32_t ws_cap = 0;
  uint32_t* next = nullptr;
  // what the compaction (K3) reads is kept per batch PARITY, so that the compaction of batch b can run on a second stream while the probe
  // kernel of batch b + 1 fills the other set ("deferred compaction", merge_core): winner bytes, the claimers' slots, the deltas' field
  // hashes (for the index change log) and the sharded counters
  // THREE sets: while batch k is probed the compactions of batches k - 1 (just released) and k - 2 (not waited for yet) may both still be reading theirs
  static constexpr uint32_t WS_SETS = 3, BLK_SEGS = 4;
  uint8_t* wflag[WS_SETS] = {nullptr, nullptr, nullptr};
  uint32_t* slot_of[WS_SETS] = {nullptr, nullptr, nullptr};
  uint32_t* fld_ws[WS_SETS] = {nullptr, nullptr, nullptr};
  uint32_t ws_par = 0;
  uint32_t* blk_info = nullptr;       // 4 x (ws_cap/256 + 16) block summaries: batch k adds into segment k % 4 and k_probe_apply zeroes segment (k + 1) % 4 for batch k + 1;
                                      // four, because the compactions of batches k - 1 and k - 2 may still read theirs while batch k is probed
  uint32_t blk_half = 0, blk_seg = 0; bool blk_clean[BLK_SEGS] = {false, false, false, false};   // blk_clean[h]: segment h is known to be all zero
  uint32_t* blk_follow = nullptr;     // ws_cap/256 epoch tags: a delta of the block got a follower on its row
  unsigned long long* shard_ctr = nullptr;  // WS_SETS x CTR_SHARDS * CTR_STRIDE (one per workspace set)
  // staging for BMX_MEM_HOST calls
  // Two staging sets: batch b+1 is uploaded (copy stream) while batch b is merged (main stream); bmx_merge_submit / bmx_merge_collect
  struct Staging {
    uint32_t cap = 0;
    uint64_t* id = nullptr; uint32_t* field = nullptr; int64_t* ts = nullptr; int64_t* val = nullptr;
    uint32_t* applied = nullptr; uint8_t* flags = nullptr;
    unsigned long long* n_out = nullptr; bmx_merge_stats* stats = nullptr;   // device words of this set
    SmallOut* tail = nullptr;                                                 // the same, in mapped host memory (null: copied down instead)
    hipEvent_t up = nullptr, done = nullptr;                                  // inputs uploaded / kernels of the batch finished
    uint64_t n = 0; bool want_flags = false; bool busy = false; uint64_t ticket = 0;
  } stg[2];
  // small host batches (<= SMALL_HOST_N deltas): inputs are packed into mapped host memory the kernels read directly, results are written
  // straight into mapped host memory: three launches and one stream synchronisation per call, no copies, no second stream
  uint8_t* pin_in = nullptr; uint8_t* pin_out = nullptr;
  SmallOut* stg_tails = nullptr;       // mapped host memory behind stg[i].tail
  hipStream_t copy_stream = nullptr;   // uploads
  hipStream_t down_stream = nullptr;   // downloads (PCIe is full duplex: results of batch b come back while batch b+1 goes up)
  uint64_t next_ticket = 1;
  // persistent device buffers of the host-mode point reads and dumps (grow-only)
  uint64_t pr_cap = 0;
  uint64_t* pr_id = nullptr; uint32_t* pr_field = nullptr; int64_t* pr_ts = nullptr; int64_t* pr_val = nullptr; uint8_t* pr_found = nullptr;
  uint64_t scan_cap = 0; uint64_t* scan_out = nullptr;
  bool scan_defer = false; uint64_t scan_defer_cap = 0;   // host-mode scan split in two (bmx_comm_scan_*): enqueue now, scan_collect() later
  uint32_t* block_counts = nullptr;   // SEL_MAX_BLOCKS
  uint32_t* scan_mask = nullptr;      // scan scratch: one match bit per index row
  uint32_t* scan_counts = nullptr;    // scan scratch: matches per 8192-row block (+ total)
  uint64_t scan_blocks_cap = 0;
  bool fixed_capacity = false;
  uint64_t nbatch = 0;
  uint32_t* part_counts = nullptr;    // PART_MAX_SHARDS * PART_BLOCKS
  uint8_t* part_owner = nullptr;      // owner shard of every delta of the batch being partitioned
  uint64_t part_owner_cap = 0;
  uint32_t epoch = 0;
  uint64_t version = 0;
  uint64_t rows_ub = 0;               // host-side upper bound of resident rows
  // {rows, sequence number of the batch that produced them}, written by every merge's last workgroup into mapped host memory: the capacity
  // guard tightens rows_ub from it (rows seen + deltas of the batches enqueued since) instead of synchronising the stream every few batches
  unsigned long long* host_rows = nullptr;
  uint64_t batch_seq = 0;
  std::deque<std::pair<uint64_t, uint64_t>> inflight;   // (sequence number, deltas) of batches whose row count the host has not seen yet
  std::vector<Index> indexes;
  // incremental index maintenance (scan_kernels.h): slot -> position in its field's index, and the log of the winners' slots since the
  // indices were last brought up to date. chg_valid: the log is complete (every merge since then was logged and nothing moved the slots).
  uint32_t* slot_pos = nullptr; uint64_t slot_pos_n = 0;
  uint2* chg = nullptr; uint64_t chg_cap = 0, chg_ub = 0;
  bool chg_valid = false; uint32_t chg_par = 0;
  uint64_t ix_full_builds = 0, ix_incremental = 0;
  // sort scratch of the view patches (view_kernels.h): two key arrays (value 8 B, position 4 B) the merge sort ping-pongs between
  void* vk_v[2] = {nullptr, nullptr}; uint32_t* vk_p[2] = {nullptr, nullptr}; uint64_t vk_cap = 0;
  void* vk_sv = nullptr; uint32_t* vk_sp = nullptr; uint32_t* vk_d0 = nullptr; uint32_t* vk_y0 = nullptr; uint64_t vk_tiles_cap = 0;   // per tile of the view: its first key (the sample), deleted indices / inserted keys in front of it
  bool view_patching = true;          // BMX_VIEW_PATCH=0 in the environment: a change makes the view stale as in round 4 (A/B switch)
  volatile unsigned long long* hres = nullptr;                            // mapped page-locked result words (HRES_*): counts the host waits for arrive without a download
  uint32_t* view_err_host = nullptr; hipEvent_t view_ev = nullptr;      // a background rewrite's error word (page-locked host memory) and completion event
  int view_test_fail = 0;             // BMX_TEST_VIEW_FAIL (test hook): 1 = every patch reports failure (the view goes stale), 2 = every background rewrite reports failure (main + patch go on answering)
  bool view_own_sort = false;         // BMX_VIEW_SORT=own: a new view is sorted by the patch path's own kernels (tile sort + merge passes) instead of rocPRIM's radix sort (A/B switch)
  bool view_pending = true;           // BMX_VIEW_PENDING=0: every patch rewrites the view's main run at once (no pending patch; A/B switch)
  // bmx_merge_notify: words (possibly in other GPUs' memory) that every merge's last workgroup sets to the number of merges finished since
  SeqPtrs notify{}; uint32_t n_notify = 0; uint64_t notify_seq = 0;
  // bmx_merge_tail_wait: armed = the next default-path merge's resolve kernel polls these words before it ends; waited = a resolve kernel that did so has
  // been enqueued (the bmx_merge_records_after that asks for the same wait then launches no wait kernel)
  struct TailWait { const unsigned long long* words = nullptr; uint32_t n = 0; unsigned long long at_least = 0; } tail_armed, tail_waited;
  bool notify_armed = false;          // set by bmx_merge_records_after around ITS merge: only the merges of the slab protocol count up the peers' free words
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Deferred compaction (merge_core): the compaction of a device-resident batch is not launched with the batch. If the next call is another such
  // merge, it goes to a high-priority side stream behind a one-wave wait for that merge's probe kernel to START (= everything of this batch is
  // done), and runs under that probe kernel; anything else launches it on the context's stream first (flush_pending).
  struct PendingK3 {
    bool on = false;
    const uint8_t* wflag = nullptr; const uint32_t* blk = nullptr; uint32_t n = 0; uint32_t* applied = nullptr;
    FinishMerge Fin{}; ChgLog L{}; uint32_t mark_created = 0; bool notify_after = false; uint64_t notify_seq = 0; uint64_t seq = 0;
  } pend;
  bool defer_enabled = true;
  uint32_t placement_tries = 0, placement_tries_asked = 0; float placement_us_best = 0, placement_us_worst = 0;   // what alloc_table_tuned saw for the current table
  uint64_t n_row_waits = 0;           // merges that waited for a batch in flight to report its row count (wait_for_row_reports)
  int k1_waves = 8;                   // BMX_K1_WAVES (8, 6 or 5): resident waves per SIMD of the probe kernel
  hipStream_t side = nullptr;
  bool side_is_callers = false;       // bmx_set_side_stream: the deferred compactions run on a stream the caller owns (the sharded pipeline's exchange stream)
  uint64_t dseq = 0;                  // deferred merges so far (the sequence numbers in ds->seqw)
  uint64_t side_last = 0, side_prev = 0;   // sequence numbers of the last two compactions launched on the side stream (0: none this stream is not ordered behind already)
  uint64_t n_deferred = 0, n_side = 0;   // merges whose compaction was deferred / actually ran on the side stream (bmx_get_deferred_counts)
  // optional per-kernel profiling (bmx_profile_enable)
  bool prof_on = false;
  uint32_t prof_n = 0;
  std::vector<hipEvent_t> prof_ev;    // 4 events per profiled call
  std::vector<hipEvent_t> scan_ev;    // 3 events per profiled scan call (before the mask pass, after it, after the emit pass)
  uint32_t scan_prof_n = 0;
  std::string err;
};

namespace {

int fail(bmx_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_err = msg;
  return code;
}
inline hipMemcpyKind host_or_dev(int mem) { return mem == BMX_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice; }
int fail_hip(bmx_ctx* c, hipError_t e, const char* what) {
  return fail(c, BMX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call)                                                  \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) return fail_hip(ctx, e__, #call);          \
  } while (0)
#define LAUNCHCHK(name)                                               \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) return fail_hip(ctx, e__, "launch " name); \
  } while (0)

template <class T>
int dev_alloc(bmx_ctx* ctx, T** p, uint64_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
  if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? BMX_ERR_NOMEM : BMX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  return BMX_OK;
}
template <class T>
void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// Pull the sticky device status; translate to an error code.
int check_status(bmx_ctx* ctx) {
  uint32_t st = 0;
  HIPCHK(hipMemcpyAsync(&st, &ctx->ds->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (!st) return BMX_OK;
  HIPCHK(hipMemsetAsync(&ctx->ds->status, 0, sizeof(uint32_t), ctx->stream));
  if (st & ST_SPIN) {
    unsigned long long d[3] = {0, 0, 0};
    (void)hipMemcpy(d, ctx->ds->seq_diag, sizeof(d), hipMemcpyDeviceToHost);
    (void)hipMemset(ctx->ds->seq_diag, 0, sizeof(d));
    if (d[0]) {
      char buf[200];
      snprintf(buf, sizeof(buf), "device protocol fault: bmx_seq_wait on word %p expired after ~60 s waiting for %llu (last seen %llu): the signalling stream or peer never got there",
               (void*)(uintptr_t)d[0], d[1], d[2]);
      return fail(ctx, BMX_ERR_INTERNAL, buf);
    }
    return fail(ctx, BMX_ERR_INTERNAL, "device protocol fault: bounded spin expired");
  }
  if (st & ST_FULL) return fail(ctx, BMX_ERR_FULL, "resident table is full");
  if (st & ST_SLAB) return fail(ctx, BMX_ERR_OVERFLOW, "an exchange slab was too small for the records routed to one shard: records were dropped, re-route the batch with bmx_partition_by_owner");
  return fail(ctx, BMX_ERR_RANGE, "delta out of domain: reserved key, ts outside [0, 2^53-1] or |val| > 2^53-1");
}

int refresh_rows(bmx_ctx* ctx) {
  unsigned long long r = 0;
  if (ctx->host_rows && ctx->batch_seq > 0) {      // every change of the row count went through a merge, whose last workgroup mirrored it to the host
    HIPCHK(hipStreamSynchronize(ctx->stream));
    r = __atomic_load_n(ctx->host_rows, __ATOMIC_ACQUIRE);
  } else {
    HIPCHK(hipMemcpyAsync(&r, &ctx->ds->row_count, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  ctx->rows_ub = r;
  ctx->inflight.clear();
  return BMX_OK;
}

// rows_ub from the mirror the merges write, no sync: rows the host has seen + every delta of the batches enqueued after that one
void tighten_rows_ub(bmx_ctx* ctx) {
  if (!ctx->host_rows) return;
  const uint64_t seen_seq = __atomic_load_n(ctx->host_rows + 1, __ATOMIC_ACQUIRE);
  const uint64_t seen_rows = __atomic_load_n(ctx->host_rows, __ATOMIC_RELAXED);   // this count, or a later one: still an upper bound with the sum below
  if (seen_seq == 0) return;
  while (!ctx->inflight.empty() && ctx->inflight.front().first <= seen_seq) ctx->inflight.pop_front();
  uint64_t pending = 0;
  for (const auto& b : ctx->inflight) pending += b.second;
  ctx->rows_ub = std::min<uint64_t>(ctx->rows_ub, seen_rows + pending);
}

int flush_pending(bmx_ctx* ctx);
// (merge_core's capacity guard) Blocks the CALLER, not the device: until the row reports of enough batches in flight have arrived for the bound to clear,
// or none is left. A report that does not come within two seconds (a wedged queue) leaves the decision to the synchronising path.
int wait_for_row_reports(bmx_ctx* ctx, uint64_t n) {
  if (!ctx->host_rows) return BMX_OK;
  const auto t0 = std::chrono::steady_clock::now();
  while ((ctx->rows_ub + n >= ctx->nslots || ctx->rows_ub > ctx->capacity_rows) && !ctx->inflight.empty()) {
    if (ctx->inflight.size() == 1 && ctx->pend.on) { if (int frc = flush_pending(ctx)) return frc; }   // the only report outstanding is that of a compaction not launched yet
    const uint64_t want = ctx->inflight.front().first;
    uint32_t spins = 0;
    while (__atomic_load_n(ctx->host_rows + 1, __ATOMIC_ACQUIRE) < want) {
      if ((++spins & 1023u) == 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return BMX_OK;
        std::this_thread::yield();
      }
    }
    tighten_rows_ub(ctx);
    ctx->n_row_waits++;
  }
  return BMX_OK;
}
int ensure_workspace(bmx_ctx* ctx, uint64_t n) {
  if (n <= ctx->ws_cap) return BMX_OK;
  if (int frc = flush_pending(ctx)) return frc;      // a compaction not launched yet reads the workspace this call frees
  HIPCHK(hipStreamSynchronize(ctx->stream));
  uint64_t cap = std::max<uint64_t>(n, std::min<uint64_t>((uint64_t)ctx->ws_cap * 2, MAX_BATCH));
  cap = std::max<uint64_t>(cap, 1u << 16);
  cap = (cap + 255) & ~255ull;
  dev_free(ctx->next); dev_free(ctx->blk_info); dev_free(ctx->blk_follow);
  for (uint32_t h = 0; h < bmx_ctx::WS_SETS; h++) { dev_free(ctx->wflag[h]); dev_free(ctx->slot_of[h]); dev_free(ctx->fld_ws[h]); }
  ctx->ws_cap = 0;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->next, cap))) return rc;
  for (uint32_t h = 0; h < bmx_ctx::WS_SETS; h++)
    if ((rc = dev_alloc(ctx, &ctx->wflag[h], cap + 16)) || (rc = dev_alloc(ctx, &ctx->slot_of[h], cap)) || (rc = dev_alloc(ctx, &ctx->fld_ws[h], cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->blk_info, bmx_ctx::BLK_SEGS * ((cap / 256 + 16 + 3) & ~3ull))) || (rc = dev_alloc(ctx, &ctx->blk_follow, cap / 256 + 16))) return rc;
  HIPCHK(hipMemsetAsync(ctx->next, 0, cap * sizeof(uint32_t), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->blk_follow, 0, (cap / 256 + 16) * sizeof(uint32_t), ctx->stream));
  ctx->blk_half = (uint32_t)((cap / 256 + 16 + 3) & ~3ull);    // a multiple of four entries: every segment stays 16-byte aligned for the compaction's wide loads
  HIPCHK(hipMemsetAsync(ctx->blk_info, 0, bmx_ctx::BLK_SEGS * (size_t)ctx->blk_half * sizeof(uint32_t), ctx->stream));
  for (uint32_t h = 0; h < bmx_ctx::BLK_SEGS; h++) ctx->blk_clean[h] = true;
  ctx->ws_cap = (uint32_t)cap;
  return BMX_OK;
}

// The copy streams exist only once a host batch is submitted: HIP maps streams onto a few hardware queues, and a device-mode caller
// that overlaps its own streams (the sharded pipeline: exchange beside merge) must not find them sharing a queue with idle ones of ours
// (measured: with two extra streams per context the exchange kernel serialised behind the merge kernels, 164 vs 125 us per step).
int ensure_copy_streams(bmx_ctx* ctx) {
  if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!ctx->down_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
  return BMX_OK;
}

int ensure_staging(bmx_ctx* ctx, int k, uint64_t n) {
  bmx_ctx::Staging& S = ctx->stg[k];
  if (n <= S.cap) return BMX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->copy_stream));
  HIPCHK(hipStreamSynchronize(ctx->down_stream));
  uint64_t cap = std::max<uint64_t>(n, 1u << 16);
  cap = (cap + 255) & ~255ull;
  dev_free(S.id); dev_free(S.field); dev_free(S.ts); dev_free(S.val); dev_free(S.applied); dev_free(S.flags);
  S.cap = 0;
  int rc;
  if ((rc = dev_alloc(ctx, &S.id, cap)) || (rc = dev_alloc(ctx, &S.field, cap)) || (rc = dev_alloc(ctx, &S.ts, cap)) ||
      (rc = dev_alloc(ctx, &S.val, cap)) || (rc = dev_alloc(ctx, &S.applied, cap)) || (rc = dev_alloc(ctx, &S.flags, cap)))
    return rc;
  S.cap = (uint32_t)cap;
  return BMX_OK;
}

int ensure_point_read(bmx_ctx* ctx, uint64_t n) {
  if (n <= ctx->pr_cap) return BMX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dev_free(ctx->pr_id); dev_free(ctx->pr_field); dev_free(ctx->pr_ts); dev_free(ctx->pr_val); dev_free(ctx->pr_found);
  ctx->pr_cap = 0;
  const uint64_t cap = (std::max<uint64_t>(n + n / 4, 1u << 12) + 255) & ~255ull;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->pr_id, cap)) || (rc = dev_alloc(ctx, &ctx->pr_field, cap)) || (rc = dev_alloc(ctx, &ctx->pr_ts, cap)) ||
      (rc = dev_alloc(ctx, &ctx->pr_val, cap)) || (rc = dev_alloc(ctx, &ctx->pr_found, cap)))
    return rc;
  ctx->pr_cap = cap;
  return BMX_OK;
}

// Slots of a table that holds `capacity_rows` rows at load factor <= load_pct %. 0 = does not fit the 32-bit slot indices the
// per-delta workspace (slot_of[]) carries: 2^32 slots x 32 B = 137 GB would fit the 288 GB of HBM, so it is refused explicitly.
uint64_t slots_for(uint64_t capacity_rows, uint32_t load_pct) {
  if (capacity_rows > (1ull << 40)) return 0;
  uint64_t nslots = std::max<uint64_t>(4096, (capacity_rows * 100 + load_pct - 1) / load_pct);
  nslots = (nslots + 3) & ~3ull;
  return nslots >= (1ull << 32) ? 0 : nslots;   // the last 32-bit value is a sentinel (STRICT_NO_ROW)
}

// Where a table lands matters: the same kernels on the same rows take 68-72 us per 1M-delta launch on some allocations of a 1.4 GB table and 77-80 us on
// others made in the same process minutes apart — a property of the allocation that stays for its lifetime (profiles/r04_placement_probe.log: six tables alive at
// once, three passes; which ones are fast changes from run to run). So a large table is allocated up to PLACEMENT_TRIES times (fewer once a clearly faster candidate has turned up), every candidate is timed with
// the probe kernel's own request mix (k_placement_probe: 2^20 random slot reads + head exchanges + 16-byte stores, best of three launches, ~0.25 ms per
// candidate), the fastest is kept and the others are freed. Candidates stay allocated while the next one is made (otherwise the allocator hands the same range
// back); tables too large for that many copies get fewer tries. BMX_TABLE_PLACEMENT_TRIES=1 switches it off. -> the chosen allocation (uninitialised)
constexpr int PLACEMENT_TRIES = 4;   // at create (BMX_CTX_PLACEMENT_TRIES(n) in bmx_create_ex's flags, then BMX_TABLE_PLACEMENT_TRIES in the environment, override: 1..8);
                                     // round 4 tried up to eight: the transient footprint (8 x 1.4 GB) bought ~1 us over four (profiles/r04_placement_probe.log)
constexpr int PLACEMENT_TRIES_GROW = 3;  // while the old table is alive as well: old + 3 candidates = 4 x the table at the peak
constexpr uint64_t PLACEMENT_MIN_BYTES = 256ull << 20;     // below the Infinity Cache's size a table's lines are served on-die wherever they live
int alloc_table_tuned(bmx_ctx* ctx, uint64_t nslots, Slot** out, bool growing = false) {
  *out = nullptr;
  int tries = growing ? PLACEMENT_TRIES_GROW : PLACEMENT_TRIES;
  if (ctx->placement_tries_asked) tries = growing ? std::min<int>(ctx->placement_tries_asked, PLACEMENT_TRIES_GROW) : (int)ctx->placement_tries_asked;
  if (const char* t = std::getenv("BMX_TABLE_PLACEMENT_TRIES")) { const int v = std::atoi(t); if (v >= 1 && v <= 8) tries = v; }
  const uint64_t bytes = nslots * sizeof(Slot);
  if (bytes < PLACEMENT_MIN_BYTES) tries = 1;
  else {
    // never more than HALF of what is free right now for the candidates together (other contexts, torch's allocator and other processes share the
    // device: ADVICE r4), and never without 2 GB to spare
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { while (tries > 1 && ((uint64_t)tries * bytes > free_b / 2 || (uint64_t)tries * bytes + (2ull << 30) > free_b)) tries--; } else (void)hipGetLastError();
  }
  int rc;
  if (tries == 1) return dev_alloc(ctx, out, nslots);
  std::vector<Slot*> cand;
  std::vector<float> us;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); if (e0) (void)hipEventDestroy(e0); return dev_alloc(ctx, out, nslots); }
  constexpr uint32_t PN = 1u << 20;
  const char* cenv = std::getenv("BMX_TABLE_CONTIGUOUS");
  const int n_contig = cenv ? std::atoi(cenv) : 0;         // measurement switch: the first n candidates are asked for as physically contiguous memory
  for (int k = 0; k < tries; k++) {
    Slot* p = nullptr;
    if (k < n_contig) {
      if (hipExtMallocWithFlags(reinterpret_cast<void**>(&p), bytes, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
      if (std::getenv("BMX_PLACEMENT_DEBUG")) fprintf(stderr, "bmx placement: candidate %d: contiguous allocation %s\n", k, p ? "granted" : "refused");
    }
    if (!p && (rc = dev_alloc(ctx, &p, nslots))) { if (cand.empty()) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; } break; }
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {                    // (the first launch on a fresh allocation also pays its page-table walk misses: not counted)
      (void)hipEventRecord(e0, ctx->stream);
      hipLaunchKernelGGL(k_placement_probe, dim3(PN / 64), dim3(64), 0, ctx->stream, p, nslots, PN, (uint32_t)(k * 16 + rep));
      (void)hipEventRecord(e1, ctx->stream);
      float ms = 0;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); ms = 1e9f; }
      if (rep > 0) best = std::min(best, ms * 1000.f);
    }
    cand.push_back(p); us.push_back(best);
    if (std::getenv("BMX_PLACEMENT_DEBUG")) fprintf(stderr, "bmx placement: candidate %d at %p (%llu MB): probe %.2f us\n", k, (void*)p, (unsigned long long)(bytes >> 20), best);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  size_t pick = 0;
  for (size_t k = 1; k < cand.size(); k++) if (us[k] < us[pick]) pick = k;
  for (size_t k = 0; k < cand.size(); k++) if (k != pick) (void)hipFree(cand[k]);
  ctx->placement_tries = (uint32_t)cand.size(); ctx->placement_us_best = us[pick];
  ctx->placement_us_worst = *std::max_element(us.begin(), us.end());
  *out = cand[pick];
  return BMX_OK;
}

// Rehash into a table for `capacity_rows` rows. Synchronous.
int grow_table(bmx_ctx* ctx, uint64_t capacity_rows) {
  if (capacity_rows <= ctx->capacity_rows) return BMX_OK;
  int rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t nslots = slots_for(capacity_rows, ctx->load_pct);
  if (!nslots) return fail(ctx, BMX_ERR_INVALID, "table would need more than 2^32 slots (slot indices are 32-bit): shard the graph over more contexts");
  Slot* fresh = nullptr;
  if ((rc = alloc_table_tuned(ctx, nslots, &fresh, /*growing=*/true))) return rc;
  hipLaunchKernelGGL(k_init_slots, dim3(2048), dim3(256), 0, ctx->stream, fresh, nslots);
  hipLaunchKernelGGL(k_rehash, dim3(2048), dim3(256), 0, ctx->stream, ctx->slots, ctx->nslots, fresh, nslots, &ctx->ds->status);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { dev_free(fresh); return fail_hip(ctx, e, "grow_table"); }
  dev_free(ctx->slots);
  ctx->slots = fresh; ctx->nslots = nslots; ctx->capacity_rows = capacity_rows;
  // the batch epoch keeps counting: next[] still holds links tagged with earlier epochs; the new heads are all 0
  ctx->version++;          // indices are rebuilt on their next use
  ctx->chg_valid = false;  // every row moved: the recorded slot positions mean nothing any more
  dev_free(ctx->slot_pos); ctx->slot_pos_n = 0;
  for (auto& ix : ctx->indexes) ix.has_pos = false;
  return check_status(ctx);
}

// ---- deferred compaction ---------------------------------------------------------------------------------------------------------------
// K3 (k_compact_winners) reads only per-batch workspace — winner bytes, block counts, the claimers' slots — and writes what the CALLER reads
// after bmx_sync(): applied_idx, n_applied, stats. Nothing the next batch's probe kernel needs. So for a stream of device-resident batches the
// compaction of batch b runs on a second, high-priority stream UNDER the probe kernel of batch b + 1 (which is bound by memory-side requests in
// flight, not by CUs), and the context's stream carries K1 -> K2 -> K1 -> K2 ... only. Ordering without command-processor markers (an event
// record between two kernels holds a stream ~10 us, bmx.h "bmx_seq_signal"):
//   side stream:  k_seq_wait(seqw[0] >= q + 1)  ->  K3(q)  ->  k_seq_signal(seqw[1] = q)
//   main stream:  K1(q + 1) [block 0 stores seqw[0] = q + 1 when it starts: K2(q) is done]  ->  K2(q + 1) [block 0 returns once seqw[1] >= q]
// so K3(q) is complete before K1(q + 2) overwrites the workspace half it read, whatever the side stream's queue does. K3 is launched LATE: a
// merge only records it (ctx->pend); the next deferring merge puts it on the side stream, anything else (bmx_sync, a scan, a host batch, a
// merge on another path, ...) launches it on the context's own stream first (flush_pending, called by every entry point) — after which that
// stream is ordered behind everything, because the K2 in front of it waited for the only compaction that could still be running on the side.
void launch_k3(bmx_ctx* ctx, const bmx_ctx::PendingK3& P, hipStream_t ks) {
  hipLaunchKernelGGL((k_compact_winners<FinishMerge>), dim3((uint32_t)(((uint64_t)P.n + 4095) / 4096)), dim3(SEL_THREADS), 0, ks, P.wflag, P.blk, P.n,
                     P.applied, P.Fin, P.L, P.mark_created);
  if (P.notify_after) hipLaunchKernelGGL(k_seq_signal_multi, dim3(1), dim3(64), 0, ks, ctx->notify, ctx->n_notify, (unsigned long long)P.notify_seq);
}
// The deferral protocol lets a one-wave kernel on the side stream wait for a kernel on the context's stream to START. Where the runtime or a tool runs
// kernels strictly one at a time — rocprofv3 counter collection (--pmc serialises every dispatch of the device), HIP_LAUNCH_BLOCKING,
// AMD_SERIALIZE_KERNEL — that wait would never end (it expires after ~60 s and raises BMX_ERR_INTERNAL). Such processes keep every launch in stream order.
bool launches_are_serialized() {
  auto on = [](const char* name) { const char* v = std::getenv(name); return v && v[0] && !(v[0] == '0' && !v[1]) && std::strcmp(v, "False") && std::strcmp(v, "false"); };
  return on("ROCPROF_COUNTER_COLLECTION") || on("HIP_LAUNCH_BLOCKING") || on("AMD_SERIALIZE_KERNEL") || on("BMX_NO_DEFERRED_COMPACTION");
}
int flush_pending(bmx_ctx* ctx) {
  if (ctx->side_last) {
    // the resolve kernel at the end of this stream waited for the compaction launched on the side stream TWO batches ago only: the last one may still be
    // running there (it was released when the last probe kernel started, so this one-wave wait is a formality, and it cannot starve anything)
    hipLaunchKernelGGL(k_seq_wait, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long*)&ctx->ds->seqw[1], (unsigned long long)ctx->side_last, &ctx->ds->status, ctx->ds->seq_diag);
    LAUNCHCHK("k_seq_wait");
    ctx->side_last = ctx->side_prev = 0;
  }
  if (!ctx->pend.on) return BMX_OK;
  ctx->pend.on = false;
  launch_k3(ctx, ctx->pend, ctx->stream);
  LAUNCHCHK("k_compact_winners");
  return BMX_OK;
}
// every entry point that is not a deferring merge: bind the device, launch a compaction that is still only recorded
int enter(bmx_ctx* ctx) {
  HIPCHK(hipSetDevice(ctx->device));
  return flush_pending(ctx);
}
constexpr uint64_t DEFER_MIN_N = 1u << 16;   // below this a batch is launch-bound: the side stream's three extra launches would cost more than the compaction

// The merge proper: all pointers are device pointers; only enqueues work. `defer`: the caller reads applied_idx / n_applied / stats only after
// bmx_sync() or another bmx_* call on this context (the BMX_MEM_DEVICE contract), so the compaction may be deferred (above).
template <bool AOS>
int merge_core(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
               const bmx_delta_rec* recs, int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags,
               bmx_merge_stats* stats, bool defer = false, bool force = false) {
  if (n > MAX_BATCH) return fail(ctx, BMX_ERR_INVALID, "batch larger than 2^24 deltas: split it (sequential semantics are preserved)");
  const uint32_t mark_created = (insert_mode & BMX_MERGE_MARK_CREATED) ? 1u : 0u;
  const bool unique = (insert_mode & BMX_MERGE_UNIQUE_KEYS) != 0 || force;      // force = bmx_put_rows: unique keys, stored as given
  const bool strict = (insert_mode & BMX_MERGE_STRICT_FLAGS) != 0;
  if (insert_mode & ~(BMX_INSERT_DELTA | BMX_MERGE_UNIQUE_KEYS | BMX_MERGE_STRICT_FLAGS | BMX_MERGE_MARK_CREATED)) return fail(ctx, BMX_ERR_INVALID, "bad insert_mode");
  insert_mode &= ~(BMX_MERGE_UNIQUE_KEYS | BMX_MERGE_STRICT_FLAGS | BMX_MERGE_MARK_CREATED);
  if (force) insert_mode = BMX_INSERT_DELTA;
  if (unique && strict) return fail(ctx, BMX_ERR_INVALID, "BMX_MERGE_STRICT_FLAGS cannot be combined with BMX_MERGE_UNIQUE_KEYS");
  if (insert_mode != BMX_INSERT_REFERENCE && insert_mode != BMX_INSERT_DELTA) return fail(ctx, BMX_ERR_INVALID, "bad insert_mode");
  int rc;
  hipEvent_t* pe = (ctx->prof_on && ctx->prof_n < PROF_MAX_CALLS) ? &ctx->prof_ev[4 * ctx->prof_n] : nullptr;
  // the compaction of THIS batch is deferred iff the default path runs (K1 + K2) on a batch big enough to hide it behind; per-kernel profiling brackets every launch
  const bool deferring = defer && ctx->defer_enabled && !strict && !unique && !pe && n >= DEFER_MIN_N;
  if (!deferring && (rc = flush_pending(ctx))) return rc;     // everything else sees the stream in order
  if (n == 0) {
    if ((rc = flush_pending(ctx))) return rc;
    if (n_applied) HIPCHK(hipMemsetAsync(n_applied, 0, sizeof(uint64_t), ctx->stream));
    if (stats) HIPCHK(hipMemsetAsync(stats, 0, sizeof(bmx_merge_stats), ctx->stream));
    return BMX_OK;
  }
  // capacity guards: physical (never let probing run out of empty slots) and logical (capacity_rows)
  if (ctx->rows_ub + n >= ctx->nslots || ctx->rows_ub > ctx->capacity_rows) tighten_rows_ub(ctx);
  if (ctx->rows_ub + n >= ctx->nslots || ctx->rows_ub > ctx->capacity_rows) {
    // The bound counts every delta of every batch still in flight as a new row. A host that runs many batches ahead of the device (a stream of
    // device batches: ~10 us per call against ~80 us per batch) reaches it long before the table is full: wait for the OLDEST batch in flight to
    // report its row count (its compaction writes the host-visible mirror) and look again — the device keeps its queue, nothing drains. Only when
    // nothing is left in flight does the exact count decide (below).
    if ((rc = wait_for_row_reports(ctx, n))) return rc;
  }
  if (ctx->rows_ub + n >= ctx->nslots || ctx->rows_ub > ctx->capacity_rows) {
    if ((rc = flush_pending(ctx))) return rc;                 // the exact row count is the last compaction's
    rc = refresh_rows(ctx);
    if (rc) return rc;
    if (ctx->rows_ub + n >= ctx->nslots || ctx->rows_ub > ctx->capacity_rows) {
      if (ctx->fixed_capacity) return fail(ctx, BMX_ERR_FULL, "resident table is full (capacity_rows exceeded)");
      uint64_t want = std::max<uint64_t>(ctx->capacity_rows * 2, ctx->rows_ub + n + n / 2);   // amortised doubling
      if ((rc = grow_table(ctx, want))) return rc;
    }
  }
  rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  if (deferring && !ctx->side) {
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // hi = the numerically smallest = highest priority: a hardware queue of its own
    HIPCHK(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, hi));
  }
  if (++ctx->epoch > EPOCH_MAX) {  // tags wrap: forget every claim
    hipLaunchKernelGGL(k_sweep_heads, dim3(2048), dim3(256), 0, ctx->stream, ctx->slots, ctx->nslots);
    LAUNCHCHK("k_sweep_heads");
    HIPCHK(hipMemsetAsync(ctx->next, 0, (size_t)ctx->ws_cap * sizeof(uint32_t), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->blk_follow, 0, ((size_t)ctx->ws_cap / 256 + 16) * sizeof(uint32_t), ctx->stream));
    ctx->epoch = 1;
  }
  // this batch's workspace set and block-summary segment
  const uint32_t par = (ctx->ws_par = (ctx->ws_par + 1u) % bmx_ctx::WS_SETS);
  const uint32_t seg = ctx->blk_seg, seg_next = (seg + 1u) % bmx_ctx::BLK_SEGS;
  ctx->blk_seg = seg_next;
  uint8_t* wflag = ctx->wflag[par];
  unsigned long long* ctr = ctx->shard_ctr + (size_t)par * CTR_SHARDS * CTR_STRIDE;
  MergeArgs A;
  A.slots = ctx->slots; A.nslots = ctx->nslots;
  A.id = id; A.field = field; A.ts = ts; A.val = val; A.recs = recs;
  A.n = (uint32_t)n; A.epoch = ctx->epoch;
  A.next = ctx->next; A.wflag = wflag; A.flags = flags;
  A.slot_of = ctx->slot_of[par]; A.blk_follow = ctx->blk_follow; A.shard_ctr = ctr; A.status = &ctx->ds->status;
  A.blk_info = ctx->blk_info + (size_t)seg * ctx->blk_half; A.blk_next = ctx->blk_info + (size_t)seg_next * ctx->blk_half; A.blk_ents = ctx->blk_half;
  const bool wave_k1 = !strict;       // k_probe_apply: adds into its segment, zeroes the next one
  if (wave_k1 && !ctx->blk_clean[seg]) HIPCHK(hipMemsetAsync(A.blk_info, 0, (size_t)ctx->blk_half * sizeof(uint32_t), ctx->stream));   // a batch on another path used this segment last
  ctx->blk_clean[seg] = false; ctx->blk_clean[seg_next] = wave_k1;
  A.force = force ? 1u : 0u;
  // the index change log of this batch (written by its compaction): decided here because a deferred compaction reads the deltas' fields from a copy
  ChgLog L{};
  if (ctx->chg_valid) {
    if (!strict && (!unique || force) && ctx->chg_ub + n <= ctx->chg_cap && ctx->nslots < (1ull << 31)) {
      L.chg = ctx->chg; L.base = &ctx->ds->chg_n[ctx->chg_par]; L.next = &ctx->ds->chg_n[ctx->chg_par ^ 1u];
      L.slot_of = ctx->slot_of[par]; L.field = field; L.recs = recs; L.cap = ctx->chg_cap;
      if (deferring) { A.fld_out = ctx->fld_ws[par]; L.field = ctx->fld_ws[par]; L.recs = nullptr; }   // the caller's columns need not outlive this call's kernels
      ctx->chg_par ^= 1u; ctx->chg_ub += n;
    } else {
      ctx->chg_valid = false;   // this batch is not in the log (another merge path, or the log is full): the next scan rebuilds
    }
  }
  A.log_slots = (L.chg != nullptr) ? 1u : 0u;
  const bool side_k3 = deferring && ctx->pend.on;     // the compaction of the batch before goes to the side stream, under this batch's probe kernel
  if (deferring) {
    ++ctx->dseq;
    A.started = &ctx->ds->seqw[0]; A.started_val = ctx->dseq;
    if (side_k3) {
      // this batch's resolve kernel ends only once the compaction launched on the side stream BEFORE the one that goes there now is done: the next probe
      // kernel then reuses nothing a compaction still reads (three workspace sets), and that compaction has had two probe kernels' time
      if (ctx->side_last) { A.k3_done = &ctx->ds->seqw[1]; A.k3_wait = ctx->side_last; }
      if (ctx->pend.Fin.n_notify) {   // the slab set of the batch before is free the moment this probe kernel starts: said there, not under it
        A.notify = ctx->pend.Fin.notify; A.n_notify = ctx->pend.Fin.n_notify; A.notify_value = ctx->pend.Fin.notify_value;
        ctx->pend.Fin.n_notify = 0;
      }
    }
  }
  bool tail_used = false;
  if (ctx->tail_armed.n && !strict && !unique) {     // (paths without a resolve kernel leave it armed for nobody: the later wait launch is then not skipped)
    A.tail_words = ctx->tail_armed.words; A.tail_n = ctx->tail_armed.n; A.tail_at_least = ctx->tail_armed.at_least; A.tail_diag = ctx->ds->seq_diag;
    tail_used = true;
  }
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  const uint32_t rblocks = blocks;   // one lane per delta
  if (pe) HIPCHK(hipEventRecord(pe[0], ctx->stream));
  if (strict) {
    hipLaunchKernelGGL((k_probe_link_strict<AOS>), dim3(blocks), dim3(256), 0, ctx->stream, A);
  } else {
    constexpr int NT = 64;    // every wave its own workgroup (profiles/r03_ab_inserts.log)
    const dim3 grid((uint32_t)((n + NT - 1) / NT));
    const int kw = ctx->k1_waves;   // resident waves per SIMD the probe kernel may take (8 = all; 6 / 5 leave room for the kernels that run beside it)
#define BMX_LAUNCH_K1(KERN) do { \
      if (insert_mode == BMX_INSERT_REFERENCE) { \
        if (unique) hipLaunchKernelGGL((KERN<AOS, BMX_INSERT_REFERENCE, true, NT>), grid, dim3(NT), 0, ctx->stream, A); \
        else hipLaunchKernelGGL((KERN<AOS, BMX_INSERT_REFERENCE, false, NT>), grid, dim3(NT), 0, ctx->stream, A); \
      } else { \
        if (unique) hipLaunchKernelGGL((KERN<AOS, BMX_INSERT_DELTA, true, NT>), grid, dim3(NT), 0, ctx->stream, A); \
        else hipLaunchKernelGGL((KERN<AOS, BMX_INSERT_DELTA, false, NT>), grid, dim3(NT), 0, ctx->stream, A); \
      } } while (0)
    if (kw == 6) BMX_LAUNCH_K1(k_probe_apply_w6); else if (kw == 5) BMX_LAUNCH_K1(k_probe_apply_w5); else if (kw == 4) BMX_LAUNCH_K1(k_probe_apply_w4);
    else if (kw == 3) BMX_LAUNCH_K1(k_probe_apply_w3); else BMX_LAUNCH_K1(k_probe_apply);
#undef BMX_LAUNCH_K1
  }
  LAUNCHCHK("k_probe_apply");
  if (side_k3) {
    // K1 of this batch is enqueued: the wait below cannot be left without its signal. K3 of the batch before, on the side stream.
    hipLaunchKernelGGL(k_seq_wait, dim3(1), dim3(64), 0, ctx->side, (const unsigned long long*)&ctx->ds->seqw[0], (unsigned long long)ctx->dseq, &ctx->ds->status, ctx->ds->seq_diag);
    launch_k3(ctx, ctx->pend, ctx->side);
    hipLaunchKernelGGL(k_seq_signal, dim3(1), dim3(64), 0, ctx->side, &ctx->ds->seqw[1], (unsigned long long)ctx->pend.seq);
    ctx->side_prev = ctx->side_last; ctx->side_last = ctx->pend.seq;
    ctx->pend.on = false;
    ctx->n_side++;
    LAUNCHCHK("deferred k_compact_winners");
  }
  if (pe) HIPCHK(hipEventRecord(pe[1], ctx->stream));
  if (strict) {   // flags for every delta against the untouched rows, then the final state by the last claimers
    if (insert_mode == BMX_INSERT_REFERENCE) {
      if (flags) hipLaunchKernelGGL((k_resolve_strict<AOS, BMX_INSERT_REFERENCE, false>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL((k_resolve_strict<AOS, BMX_INSERT_REFERENCE, true>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
    } else {
      if (flags) hipLaunchKernelGGL((k_resolve_strict<AOS, BMX_INSERT_DELTA, false>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
      hipLaunchKernelGGL((k_resolve_strict<AOS, BMX_INSERT_DELTA, true>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
    }
  } else if (!unique) {  // duplicate keys can only exist without the caller's guarantee
    if (insert_mode == BMX_INSERT_REFERENCE) hipLaunchKernelGGL((k_resolve_lists<AOS, BMX_INSERT_REFERENCE>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
    else hipLaunchKernelGGL((k_resolve_lists<AOS, BMX_INSERT_DELTA>), dim3(rblocks), dim3(256), 0, ctx->stream, A);
  }
  LAUNCHCHK("k_resolve_lists");
  if (tail_used) { ctx->tail_waited = ctx->tail_armed; }
  ctx->tail_armed = bmx_ctx::TailWait{};
  if (pe) HIPCHK(hipEventRecord(pe[2], ctx->stream));
  // K3: ordered compaction of the winner bytes (+ the index change log while an index is being maintained)
  bmx_ctx::PendingK3 P;
  P.wflag = wflag; P.blk = A.blk_info; P.n = (uint32_t)n; P.applied = applied_idx; P.L = L; P.mark_created = mark_created;
  P.Fin = FinishMerge{reinterpret_cast<unsigned long long*>(n_applied), stats, ctr, &ctx->ds->row_count};
  if (ctx->host_rows) { P.Fin.host_mirror = ctx->host_rows; P.Fin.seq = ++ctx->batch_seq; ctx->inflight.emplace_back(P.Fin.seq, n); }
  const bool notifying = ctx->n_notify && ctx->notify_armed;
  P.notify_after = notifying && L.chg && !deferring;   // a change log read from the caller's columns: the compaction's workgroups still read the batch, so the peers are told from a launch behind it
  if (notifying) { P.notify_seq = ++ctx->notify_seq; if (!P.notify_after) { P.Fin.notify = ctx->notify; P.Fin.n_notify = ctx->n_notify; P.Fin.notify_value = ctx->notify_seq; } }
  if (deferring) {
    P.on = true; P.seq = ctx->dseq;
    ctx->pend = P;
    ctx->n_deferred++;
  } else {
    launch_k3(ctx, P, ctx->stream);
    LAUNCHCHK("k_compact_winners");
  }
  if (pe) { HIPCHK(hipEventRecord(pe[3], ctx->stream)); ctx->prof_n++; }
  ctx->nbatch++;
  ctx->rows_ub += n;
  ctx->version++;
  return BMX_OK;
}

// records already on the device, for callers INSIDE the library (the communicator): the put marker is honoured, the compaction is never deferred
int merge_records_internal(bmx_ctx* ctx, uint64_t n, const bmx_delta_rec* recs, int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, bmx_merge_stats* stats) {
  HIPCHK(hipSetDevice(ctx->device));
  return merge_core<true>(ctx, n, nullptr, nullptr, nullptr, nullptr, recs, insert_mode & ~MERGE_FORCE_INTERNAL, applied_idx, n_applied, nullptr, stats, false,
                          (insert_mode & MERGE_FORCE_INTERNAL) != 0);
}

// Host batches go through two staging sets. submit: upload on the copy stream, then the merge on the main stream behind an event;
// collect: results back on the copy stream once the batch's kernels are done. While the host uploads batch b+1 (a pageable
// hipMemcpyAsync keeps the calling thread busy for the whole transfer) the GPU merges batch b.
__global__ void k_noop() {}
__global__ void k_small_tail(const unsigned long long* n_applied, const bmx_merge_stats* stats, const uint32_t* status, SmallOut* out) {
  if (threadIdx.x == 0) { out->n_applied = *n_applied; out->stats = *stats; out->status = *status; }
}
int submit_host(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                int insert_mode, bool want_flags, uint64_t* ticket, bool inputs_free_on_return) {
  int k = -1;
  for (int i = 0; i < 2; i++) if (!ctx->stg[i].busy) { k = i; break; }
  if (k < 0) return fail(ctx, BMX_ERR_INVALID, "two batches are already in flight: collect the oldest first (bmx_merge_collect)");
  bmx_ctx::Staging& S = ctx->stg[k];
  int rc = ensure_copy_streams(ctx);
  if (rc) return rc;
  if ((rc = ensure_staging(ctx, k, n))) return rc;
  if (n) {
    HIPCHK(hipMemcpyAsync(S.id, id, n * 8, hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(hipMemcpyAsync(S.field, field, n * 4, hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(hipMemcpyAsync(S.ts, ts, n * 8, hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(hipMemcpyAsync(S.val, val, n * 8, hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(hipEventRecord(S.up, ctx->copy_stream));
    HIPCHK(hipStreamWaitEvent(ctx->stream, S.up, 0));
    // a copy from page-locked memory (bmx_host_alloc) is truly asynchronous: bmx_merge_submit promises that the arrays may be reused on return
    if (inputs_free_on_return) HIPCHK(hipEventSynchronize(S.up));
  }
  rc = merge_core<false>(ctx, n, S.id, S.field, S.ts, S.val, nullptr, insert_mode & ~MERGE_FORCE_INTERNAL, S.applied, reinterpret_cast<uint64_t*>(S.n_out),
                         want_flags ? S.flags : nullptr, S.stats, false, (insert_mode & MERGE_FORCE_INTERNAL) != 0);
  if (rc) return rc;
  if (S.tail) {
    hipLaunchKernelGGL(k_small_tail, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long*)S.n_out, (const bmx_merge_stats*)S.stats, (const uint32_t*)&ctx->ds->status, S.tail);
    LAUNCHCHK("k_small_tail");
  }
  HIPCHK(hipEventRecord(S.done, ctx->stream));
  S.n = n; S.want_flags = want_flags; S.busy = true; S.ticket = ctx->next_ticket++;
  *ticket = S.ticket;
  return BMX_OK;
}

int collect_host(bmx_ctx* ctx, uint64_t ticket, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  int k = -1;
  for (int i = 0; i < 2; i++) if (ctx->stg[i].busy && ctx->stg[i].ticket == ticket) k = i;
  if (k < 0) return fail(ctx, BMX_ERR_INVALID, "unknown or already collected ticket");
  if (ctx->stg[1 - k].busy && ctx->stg[1 - k].ticket < ticket) return fail(ctx, BMX_ERR_INVALID, "collect tickets in submission order");
  bmx_ctx::Staging& S = ctx->stg[k];
  S.busy = false;
  bmx_merge_stats hs; std::memset(&hs, 0, sizeof(hs));
  uint32_t st = 0;
  if (S.tail) {                         // count, stats and status are in mapped host memory once the batch's last launch is done
    HIPCHK(hipEventSynchronize(S.done));
    hs = S.tail->stats; st = S.tail->status;
  } else {
    HIPCHK(hipStreamWaitEvent(ctx->down_stream, S.done, 0));
    HIPCHK(hipMemcpyAsync(&hs, S.stats, sizeof(hs), hipMemcpyDeviceToHost, ctx->down_stream));
    HIPCHK(hipMemcpyAsync(&st, &ctx->ds->status, sizeof(st), hipMemcpyDeviceToHost, ctx->down_stream));
    HIPCHK(hipStreamSynchronize(ctx->down_stream));
  }
  if (st) return check_status(ctx);     // sticky device error of this (or an earlier, uncollected) batch
  if (S.n == 0) std::memset(&hs, 0, sizeof(hs));
  if (applied_idx && hs.n_applied) HIPCHK(hipMemcpyAsync(applied_idx, S.applied, hs.n_applied * 4, hipMemcpyDeviceToHost, ctx->down_stream));
  if (flags && S.n && S.want_flags) HIPCHK(hipMemcpyAsync(flags, S.flags, S.n, hipMemcpyDeviceToHost, ctx->down_stream));
  HIPCHK(hipStreamSynchronize(ctx->down_stream));
  if (!ctx->stg[1 - k].busy && S.n) ctx->rows_ub = hs.n_rows;   // exact again once nothing else is in flight
  if (n_applied) *n_applied = hs.n_applied;
  if (stats) *stats = hs;
  return BMX_OK;
}

// Small host batch (the reference's sync chunks hold 50 entries, src/bullet-network-sync.js:18): the general path costs ~115 us per call whatever
// the size (four pageable uploads, two extra streams, three downloads); here the columns are packed into mapped host memory that the kernels read
// over PCIe, and winners, count, stats and the device status come back through mapped host memory as well.
constexpr uint64_t SMALL_HOST_N = 32768;
constexpr int SMALL_PATH_UNAVAILABLE = 1;
constexpr size_t SMALL_IN_BYTES = SMALL_HOST_N * 28, SMALL_OUT_APPLIED = 0, SMALL_OUT_FLAGS = SMALL_HOST_N * 4, SMALL_OUT_TAIL = SMALL_HOST_N * 5,
                 SMALL_OUT_BYTES = SMALL_OUT_TAIL + sizeof(SmallOut);
// result words in mapped host memory: a kernel's last workgroup (or one copy) writes them, the host reads them after the synchronisation it needs anyway
constexpr int HRES_TOTALS = 0 /* 2 per maintained index */, HRES_RUN = PART_MAX_SHARDS /* one per index */, HRES_ERR = HRES_RUN + PART_MAX_SHARDS / 2, HRES_SPLIT = HRES_ERR + 1 /* 2 */,
              HRES_SCAN_N = HRES_SPLIT + 2, HRES_WORDS = HRES_SCAN_N + 1;
bool ensure_hres(bmx_ctx* ctx) {
  if (ctx->hres) return true;
  void* p = nullptr;
  if (hipHostMalloc(&p, HRES_WORDS * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
  std::memset(p, 0, HRES_WORDS * sizeof(unsigned long long));
  ctx->hres = static_cast<volatile unsigned long long*>(p);
  return true;
}
bool ensure_pinned(bmx_ctx* ctx) {   // the two mapped host buffers of the small-call paths (merge, point reads, scans); false = fall back to copies
  if (ctx->pin_in) return true;
  { const char* t = std::getenv("BMX_TEST_FAIL_PINNED"); if (t && t[0] == '1') return false; }   // test hook: as if the page-locked allocation had failed
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pin_in), SMALL_IN_BYTES, hipHostMallocMapped) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&ctx->pin_out), SMALL_OUT_BYTES, hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    if (ctx->pin_in) { (void)hipHostFree(ctx->pin_in); ctx->pin_in = nullptr; }
    ctx->pin_out = nullptr;
    return false;
  }
  return true;
}
int merge_host_small(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                     int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  if (!ensure_pinned(ctx)) return SMALL_PATH_UNAVAILABLE;
  // the previous small batch's kernels are done (every call ends with a synchronisation): the buffers are free
  uint64_t* p_id = reinterpret_cast<uint64_t*>(ctx->pin_in);
  int64_t* p_ts = reinterpret_cast<int64_t*>(ctx->pin_in + n * 8);
  int64_t* p_val = reinterpret_cast<int64_t*>(ctx->pin_in + n * 16);
  uint32_t* p_field = reinterpret_cast<uint32_t*>(ctx->pin_in + n * 24);
  std::memcpy(p_id, id, n * 8); std::memcpy(p_ts, ts, n * 8); std::memcpy(p_val, val, n * 8); std::memcpy(p_field, field, n * 4);
  uint32_t* o_applied = reinterpret_cast<uint32_t*>(ctx->pin_out + SMALL_OUT_APPLIED);
  uint8_t* o_flags = ctx->pin_out + SMALL_OUT_FLAGS;
  SmallOut* o_tail = reinterpret_cast<SmallOut*>(ctx->pin_out + SMALL_OUT_TAIL);
  // count and stats go through device scalars first (the merge's last workgroup read-modify-writes them), then one thread copies them out
  int rc = merge_core<false>(ctx, n, p_id, p_field, p_ts, p_val, nullptr, insert_mode & ~MERGE_FORCE_INTERNAL, applied_idx ? o_applied : nullptr, reinterpret_cast<uint64_t*>(&ctx->ds->n_out),
                             flags ? o_flags : nullptr, &ctx->ds->stats, false, (insert_mode & MERGE_FORCE_INTERNAL) != 0);
  if (rc) return rc;
  hipLaunchKernelGGL(k_small_tail, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long*)&ctx->ds->n_out, (const bmx_merge_stats*)&ctx->ds->stats,
                     (const uint32_t*)&ctx->ds->status, o_tail);
  LAUNCHCHK("k_small_tail");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (o_tail->status) return check_status(ctx);
  const bmx_merge_stats hs = o_tail->stats;
  if (applied_idx && hs.n_applied) std::memcpy(applied_idx, o_applied, hs.n_applied * 4);
  if (flags) std::memcpy(flags, o_flags, n);
  ctx->rows_ub = hs.n_rows; ctx->inflight.clear();
  if (n_applied) *n_applied = hs.n_applied;
  if (stats) *stats = hs;
  return BMX_OK;
}

int merge_host(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
               int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  for (int i = 0; i < 2; i++)
    if (ctx->stg[i].busy) return fail(ctx, BMX_ERR_INVALID, "a submitted batch is still in flight: collect it before a synchronous merge");
  if (n && n <= SMALL_HOST_N) {
    int src = merge_host_small(ctx, n, id, field, ts, val, insert_mode, applied_idx, n_applied, flags, stats);
    if (src != SMALL_PATH_UNAVAILABLE) return src;
  }
  uint64_t ticket = 0;
  int rc = submit_host(ctx, n, id, field, ts, val, insert_mode, flags != nullptr, &ticket, false);   // collect_host waits for the whole batch
  if (rc) return rc;
  return collect_host(ctx, ticket, applied_idx, n_applied, flags, stats);
}

Index* find_index(bmx_ctx* ctx, uint32_t field) {
  for (auto& ix : ctx->indexes)
    if (ix.field == field) return &ix;
  return nullptr;
}

constexpr size_t IX_MAINTAINED_MAX = PART_MAX_SHARDS / 2;   // two scratch words of DevScalars::part_totals per maintained index

// slot -> index position map (4 B per slot) and the change log; both exist from the first index build on
int ensure_ix_maintenance(bmx_ctx* ctx) {
  int rc;
  if (ctx->slot_pos_n != ctx->nslots) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->slot_pos); ctx->slot_pos_n = 0;
    for (auto& ix : ctx->indexes) ix.has_pos = false;
    ctx->chg_valid = false;
    if (ctx->nslots >= (1ull << 31)) return BMX_OK;      // bit 31 of a log entry is the "created" mark: larger tables are rebuilt, not maintained
    if ((rc = dev_alloc(ctx, &ctx->slot_pos, ctx->nslots))) { g_err.clear(); ctx->err.clear(); return BMX_OK; }   // no memory for it: fall back to rebuilds
    ctx->slot_pos_n = ctx->nslots;
    HIPCHK(hipMemsetAsync(ctx->slot_pos, 0xFF, ctx->nslots * sizeof(uint32_t), ctx->stream));
  }
  // the log is only used while it is shorter than max(nslots/8, 1M) entries (fresh_index): size it for that, not for the largest table
  const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(ctx->nslots / 4, 1u << 20) + (1u << 16), 1u << 26);   // 1M .. 64M entries of 8 B
  if (ctx->chg_cap < want) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->chg); ctx->chg_cap = 0; ctx->chg_valid = false;
    if ((rc = dev_alloc(ctx, &ctx->chg, want))) { g_err.clear(); ctx->err.clear(); return BMX_OK; }
    ctx->chg_cap = want;
  }
  return BMX_OK;
}

// forget the log: every index is either fresh or about to be rebuilt
int reset_chg_log(bmx_ctx* ctx) {
  HIPCHK(hipMemsetAsync(ctx->ds->chg_n, 0, sizeof(ctx->ds->chg_n), ctx->stream));
  ctx->chg_par = 0; ctx->chg_ub = 0;
  return BMX_OK;
}

// (Re)build the dense columns of `field` from the table, in slot order. Synchronous.
int build_index(bmx_ctx* ctx, Index* ix) {
  int mrc = ensure_ix_maintenance(ctx);
  if (mrc) return mrc;
  PredSlotField P{ctx->slots, ix->field};
  SelGeom g = sel_geom<PredSlotField::E>(ctx->nslots);
  hipLaunchKernelGGL((k_sel_count<PredSlotField>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, ctx->nslots, g.tiles_per_block, ctx->block_counts);
  LAUNCHCHK("k_sel_count(index)");
  hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(SEL_THREADS), 0, ctx->stream, ctx->block_counts, g.blocks, &ctx->ds->n_out);
  LAUNCHCHK("k_sum_counts");
  unsigned long long n = 0;
  HIPCHK(hipMemcpyAsync(&n, &ctx->ds->n_out, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (n + (n >> 4) + (1u << 16) > ix->cap) {   // too little head room left for appended rows: a new set of columns
    dev_free(ix->ids); dev_free(ix->v64); dev_free(ix->v32);
    ix->cap = 0;
    uint64_t cap = (n + n / 8 + (1u << 16) + 1023) & ~1023ull;   // head room: rows created later are appended
    int rc;
    if ((rc = dev_alloc(ctx, &ix->ids, cap)) || (rc = dev_alloc(ctx, &ix->v64, cap)) || (rc = dev_alloc(ctx, &ix->v32, cap + 4))) return rc;
    ix->cap = cap;
  }
  HIPCHK(hipMemsetAsync(&ctx->ds->wide, 0, sizeof(uint32_t), ctx->stream));
  EmitIndex Em{ctx->slots, ix->ids, ix->v64, ix->v32, &ctx->ds->wide, ctx->slot_pos};
  FinishCount Fin{nullptr};
  hipLaunchKernelGGL((k_sel_write<PredSlotField, EmitIndex, FinishCount>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, Em, Fin, ctx->nslots,
                     g.tiles_per_block, ctx->block_counts);
  LAUNCHCHK("k_sel_write(index)");
  uint32_t wide = 0;
  HIPCHK(hipMemcpyAsync(&wide, &ctx->ds->wide, sizeof(wide), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ix->n = n;
  ix->fits32 = wide == 0;
  ix->content++;             // every position may be another row's now
  ix->version = ctx->version;
  ix->has_pos = ctx->slot_pos != nullptr;
  ctx->ix_full_builds++;
  // the log starts (or goes on) only if every index now knows its rows' positions and none is waiting for entries already logged
  if (ctx->slot_pos && ctx->chg && !ctx->chg_valid && ctx->indexes.size() <= IX_MAINTAINED_MAX) {
    bool all = true;
    for (auto& o : ctx->indexes) all = all && o.has_pos && o.version == ctx->version;
    if (all) { int rc = reset_chg_log(ctx); if (rc) return rc; ctx->chg_valid = true; }
  }
  return BMX_OK;
}

// ---- the value-ordered view kept current (view_kernels.h) ----
constexpr uint64_t VIEW_PATCH_MAX_LOG = 1ull << 24;     // a longer change log is not captured: the view goes stale and is sorted again (a sort of 10^8 rows costs less than a patch that large)
int ensure_view_scratch(bmx_ctx* ctx, uint64_t keys, uint64_t tiles) {
  if (keys > ctx->vk_cap) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; i++) { if (ctx->vk_v[i]) (void)hipFree(ctx->vk_v[i]); ctx->vk_v[i] = nullptr; dev_free(ctx->vk_p[i]); }
    ctx->vk_cap = 0;
    const uint64_t cap = (keys + keys / 4 + (1u << 16) + 255) & ~255ull;
    for (int i = 0; i < 2; i++) {
      if (hipMalloc(&ctx->vk_v[i], cap * 8) != hipSuccess) { (void)hipGetLastError(); ctx->vk_v[i] = nullptr; return fail(ctx, BMX_ERR_NOMEM, "view patch: out of device memory"); }
      if (int rc = dev_alloc(ctx, &ctx->vk_p[i], cap)) return rc;
    }
    ctx->vk_cap = cap;
  }
  if (tiles + 1 > ctx->vk_tiles_cap) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->vk_sv) (void)hipFree(ctx->vk_sv); ctx->vk_sv = nullptr; dev_free(ctx->vk_sp); dev_free(ctx->vk_d0); dev_free(ctx->vk_y0); ctx->vk_tiles_cap = 0;
    const uint64_t cap = tiles + tiles / 4 + 1024;
    if (hipMalloc(&ctx->vk_sv, cap * 8) != hipSuccess) { (void)hipGetLastError(); ctx->vk_sv = nullptr; return fail(ctx, BMX_ERR_NOMEM, "view patch: out of device memory"); }
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->vk_sp, cap)) || (rc = dev_alloc(ctx, &ctx->vk_d0, cap)) || (rc = dev_alloc(ctx, &ctx->vk_y0, cap))) return rc;
    ctx->vk_tiles_cap = cap;
  }
  return BMX_OK;
}
// Z = (X without the sorted keys D, all of which are keys of X) merged with the sorted keys Y — k_view_merge over the tiles of X, with its sample and tile offsets
// in the context's scratch. X may be empty (then D is, and Z = Y). Enqueue only; a deleted key that is not in X raises ds->view_err.
template <class T, bool HAS_IDS>
void launch_run_merge(bmx_ctx* ctx, ViewRun<T> X, uint64_t nx, const T* dv, const uint32_t* dp, uint64_t nd, const T* yv, const uint32_t* yp, uint64_t ny, const uint64_t* ix_ids, ViewRun<T> Z) {
  hipStream_t st = ctx->stream;
  if (nx == 0) {
    if (ny) {
      (void)hipMemcpyAsync(Z.v, yv, ny * sizeof(T), hipMemcpyDeviceToDevice, st); (void)hipMemcpyAsync(Z.p, yp, ny * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
      if (HAS_IDS) hipLaunchKernelGGL(k_view_gather_ids, dim3((uint32_t)std::min<uint64_t>((ny + 255) / 256, 4096)), dim3(256), 0, st, yp, (uint32_t)ny, ix_ids, Z.ids);
    }
    return;
  }
  const uint32_t ntiles = (uint32_t)((nx + VIEW_TILE - 1) / VIEW_TILE);
  T* sv = static_cast<T*>(ctx->vk_sv);
  hipLaunchKernelGGL((k_view_sample<T>), dim3((ntiles + 255) / 256), dim3(256), 0, st, (const T*)X.v, (const uint32_t*)X.p, ntiles, sv, ctx->vk_sp);
  hipLaunchKernelGGL((k_view_tile_offsets<T>), dim3((ntiles + 1 + 255) / 256), dim3(256), 0, st, (const T*)sv, (const uint32_t*)ctx->vk_sp, ntiles, dv, dp, (uint32_t)nd, yv, yp, (uint32_t)ny, ctx->vk_d0, ctx->vk_y0);
  hipLaunchKernelGGL((k_view_merge<T, HAS_IDS>), dim3(ntiles), dim3(256), 0, st, X, (uint32_t)nx, dv, dp, yv, yp, ix_ids, Z, (const uint32_t*)ctx->vk_d0, (const uint32_t*)ctx->vk_y0, &ctx->ds->view_err);
}
// the pending patch's buffers for at least `need` keys in each run (what is there is kept)
template <class T>
int ensure_pending(bmx_ctx* ctx, Index& ix, uint64_t need) {
  if (need <= ix.pend_cap) return BMX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t cap = need + need / 8 + (1u << 16);
  void* nv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; uint32_t* np[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; uint64_t* ni[2] = {nullptr, nullptr};
  uint8_t* nd = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&nd), cap) == hipSuccess;
  for (int i = 0; i < 2 && ok; i++)
    ok = hipMalloc(&nv[0][i], cap * sizeof(T)) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&np[0][i]), cap * 4) == hipSuccess && hipMalloc(&nv[1][i], cap * sizeof(T)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&np[1][i]), cap * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&ni[i]), cap * 8) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    for (int i = 0; i < 2; i++) { for (int k = 0; k < 2; k++) { if (nv[k][i]) (void)hipFree(nv[k][i]); if (np[k][i]) (void)hipFree(np[k][i]); } if (ni[i]) (void)hipFree(ni[i]); }
    if (nd) (void)hipFree(nd);
    return fail(ctx, BMX_ERR_NOMEM, "view patch: out of device memory");
  }
  const int c = ix.pcur, ci = ix.icur;
  if (ix.npd) { HIPCHK(hipMemcpy(nv[0][c], ix.pd_v[c], ix.npd * sizeof(T), hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(np[0][c], ix.pd_p[c], ix.npd * 4, hipMemcpyDeviceToDevice)); }
  if (ix.npi) { HIPCHK(hipMemcpy(nv[1][ci], ix.pi_v[ci], ix.npi * sizeof(T), hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(np[1][ci], ix.pi_p[ci], ix.npi * 4, hipMemcpyDeviceToDevice));
                HIPCHK(hipMemcpy(ni[ci], ix.pi_ids[ci], ix.npi * 8, hipMemcpyDeviceToDevice)); }
  const uint64_t kd = ix.npd, ki = ix.npi; const bool due = ix.rewrite_due;
  free_pending(ix);
  for (int i = 0; i < 2; i++) { ix.pd_v[i] = nv[0][i]; ix.pd_p[i] = np[0][i]; ix.pi_v[i] = nv[1][i]; ix.pi_p[i] = np[1][i]; ix.pi_ids[i] = ni[i]; }
  ix.pi_dead = nd;
  ix.npd = kd; ix.npi = ki; ix.pend_cap = cap; ix.pcur = c; ix.icur = ci; ix.rewrite_due = due;
  return BMX_OK;
}
// the second set of the view's columns, for `nz` rows
template <class T>
bool ensure_view_spare(bmx_ctx* ctx, Index& ix, uint64_t nz) {
  if (nz <= ix.ord_cap2 && ix.s_val2) return true;
  if (ix.s_val2) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ix.s_val2); (void)hipFree(ix.s_pos2); (void)hipFree(ix.s_ids2); }
  ix.s_val2 = nullptr; ix.s_pos2 = nullptr; ix.s_ids2 = nullptr; ix.ord_cap2 = 0;
  const uint64_t cap = std::max<uint64_t>(ix.ord_cap, nz + nz / 8 + 1024);
  if (hipMalloc(&ix.s_val2, cap * sizeof(T)) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&ix.s_pos2), cap * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&ix.s_ids2), cap * sizeof(uint64_t)) != hipSuccess) {
    (void)hipGetLastError();
    if (ix.s_val2) (void)hipFree(ix.s_val2); if (ix.s_pos2) (void)hipFree(ix.s_pos2); if (ix.s_ids2) (void)hipFree(ix.s_ids2);
    ix.s_val2 = nullptr; ix.s_pos2 = nullptr; ix.s_ids2 = nullptr;
    return false;
  }
  ix.ord_cap2 = cap;
  return true;
}
// ---- the rewrite of a view's main run, behind the answer ----
// finish_rewrite: a rewrite in flight whose event has completed (wait = true: wait for it) is looked at: error word 0 -> the second set of columns becomes main and the
// pending patch is empty; otherwise main and the patch stay what they are (they were never touched).
template <class T>
void finish_rewrite(bmx_ctx* ctx, Index& ix, bool wait) {
  if (!ix.rewrite_inflight) return;
  if (wait) (void)hipEventSynchronize(ctx->view_ev);
  else if (hipEventQuery(ctx->view_ev) != hipSuccess) { (void)hipGetLastError(); return; }
  ix.rewrite_inflight = false;
  if (*ctx->view_err_host == 0 && ctx->view_test_fail != 2) {
    std::swap(ix.s_val, ix.s_val2); std::swap(ix.s_pos, ix.s_pos2); std::swap(ix.s_ids, ix.s_ids2); std::swap(ix.ord_cap, ix.ord_cap2);
    ix.ord_n = ix.rewrite_nz; ix.npd = ix.npi = 0; ix.ord_merges++;
  }
}
// start_rewrite: enqueue main - pd + pi -> the second set of columns, then the copy of the error word and the event. Nothing waits.
template <class T>
void start_rewrite(bmx_ctx* ctx, Index& ix) {
  ix.rewrite_due = false;
  if (ix.rewrite_inflight || ix.npd + ix.npi == 0 || ix.npd > ix.ord_n) return;
  if (!ctx->view_err_host && hipHostMalloc(reinterpret_cast<void**>(&ctx->view_err_host), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->view_err_host = nullptr; return; }
  if (!ctx->view_ev && hipEventCreateWithFlags(&ctx->view_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->view_ev = nullptr; return; }
  for (auto& o : ctx->indexes) if (o.rewrite_inflight) return;                 // one at a time: they share the error word and the event
  const uint64_t nz = ix.ord_n - ix.npd + ix.npi;
  const uint32_t ntiles = (uint32_t)((ix.ord_n + VIEW_TILE - 1) / VIEW_TILE);
  if (ntiles + 1 > ctx->vk_tiles_cap || !ensure_view_spare<T>(ctx, ix, nz)) return;
  const int q = ix.pcur, qi = ix.icur;
  (void)hipMemsetAsync(&ctx->ds->view_err, 0, sizeof(uint32_t), ctx->stream);
  ViewRun<T> X{static_cast<T*>(ix.s_val), ix.s_pos, ix.s_ids}, Z{static_cast<T*>(ix.s_val2), ix.s_pos2, ix.s_ids2};
  launch_run_merge<T, true>(ctx, X, ix.ord_n, static_cast<const T*>(ix.pd_v[q]), ix.pd_p[q], ix.npd, static_cast<const T*>(ix.pi_v[qi]), ix.pi_p[qi], ix.npi, (const uint64_t*)ix.ids, Z);
  *ctx->view_err_host = 1u;                                                       // (overwritten by the copy below: an event that somehow completed without it reads as a failure)
  if (hipMemcpyAsync(ctx->view_err_host, &ctx->ds->view_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ctx->view_ev, ctx->stream) != hipSuccess) {
    (void)hipGetLastError(); (void)hipStreamSynchronize(ctx->stream); return;    // nothing was swapped: main and the patch go on answering
  }
  ix.rewrite_inflight = true; ix.rewrite_nz = nz;
}
void view_after_query(bmx_ctx* ctx, Index* ix) {         // behind the answer of an ordered query
  if (!ix->rewrite_due) return;
  const auto t0 = std::chrono::steady_clock::now();
  if (ix->ord_fits32) start_rewrite<int32_t>(ctx, *ix); else start_rewrite<int64_t>(ctx, *ix);
  if (std::getenv("BMX_VIEW_DEBUG")) std::fprintf(stderr, "bmx: rewrite enqueued in %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
}
void view_before_use(bmx_ctx* ctx, Index* ix, bool wait) {   // in front of anything that reads or changes the view
  if (!ix->rewrite_inflight) return;
  if (ix->ord_fits32) finish_rewrite<int32_t>(ctx, *ix, wait); else finish_rewrite<int64_t>(ctx, *ix, wait);
}

// Patch the view of `ix` with the change run k_ix_update captured (c changed rows: ix.cl2_pos / ix.cl2_old) and the rows appended at positions [n0, n0 + added).
// 0 = the (logical) view equals a fresh sort of the columns again; 1 = it could not be patched (no memory, or a deleted key was not where it should be): the caller
// leaves it stale and the next queries scan / re-sort as ever. Synchronous at its end (one or two words come back).
//   1. the run's deleted keys (old value, position) and inserted keys (value, position) are sorted;
//   2. they join the view's PENDING patch (pd, pi): a deleted key that is a pending inserted key cancels it, the others are deleted keys of main; the inserted keys are
//      merged into pi. All on runs of a few million keys: L2 / Infinity-Cache traffic;
//   3. once the pending patch holds more than ord_n / 16 keys, main is rewritten BEHIND the answer of the query that brought the refresh about: one streaming pass,
//      main - pd + pi (start_rewrite / finish_rewrite). Only a run of more than ord_n / 4 keys is merged into main at once, in front of the answer.
// A 1M-delta merge into a 10^8-row index: steps 1-2 on every refresh, step 3 behind every fourth; into a 10^7-row index: step 3 behind every answer.
template <class T>
int patch_view_t(bmx_ctx* ctx, Index& ix, uint64_t c, uint64_t n0, uint64_t added) {
  const auto t0 = std::chrono::steady_clock::now();
  if (ctx->view_test_fail == 1) return 1;
  finish_rewrite<T>(ctx, ix, /*wait=*/true);            // a rewrite still in flight is looked at first: the change run's keys are keys of the view as it is NOW
  const uint64_t m = c + added, ktot = c + m, nx = ix.ord_n;
  if (nx + m >= 0xFFFFFFFFull || ktot >= 0xFFFFFFFFull) return 1;
  auto soft = [&](int line) {
    if (std::getenv("BMX_VIEW_DEBUG")) std::fprintf(stderr, "bmx: view patch of field %u gave up (bmx.hip:%d): %s\n", ix.field, line, ctx->err.c_str());
    g_err.clear(); ctx->err.clear(); (void)hipGetLastError(); return 1;
  };
  const uint64_t thr = std::max<uint64_t>(nx / 16, 1u << 16);                 // a pending patch beyond this many keys makes a rewrite of main due (behind the answer)
  const uint64_t thr_direct = std::max<uint64_t>(nx / 4, 1u << 16);          // a run beyond this many keys is merged into main at once, in front of the answer
  const uint32_t ntiles_main = (uint32_t)((nx + VIEW_TILE - 1) / VIEW_TILE);
  if (!ensure_hres(ctx) || ensure_view_scratch(ctx, ktot, std::max<uint64_t>(ntiles_main, (ix.npi + ix.npd + VIEW_TILE) / VIEW_TILE + 2))) return soft(__LINE__);
  hipStream_t st = ctx->stream;
  T* kv[2] = {static_cast<T*>(ctx->vk_v[0]), static_cast<T*>(ctx->vk_v[1])};
  uint32_t* kp[2] = {ctx->vk_p[0], ctx->vk_p[1]};
  const T* col = sizeof(T) == 4 ? reinterpret_cast<const T*>(ix.v32) : reinterpret_cast<const T*>(ix.v64);
  hipLaunchKernelGGL((k_view_keys<T>), dim3((uint32_t)std::min<uint64_t>((ktot + 255) / 256, 4096)), dim3(256), 0, st, (const uint32_t*)ix.cl2_pos, (const int64_t*)ix.cl2_old, (uint32_t)c, col,
                     (uint32_t)n0, (uint32_t)added, kv[0], kp[0]);
  // 1. sort the deleted keys [0, c) and the inserted keys [c, c + m): tiles in LDS, then merge-path passes
  ViewSegs S{}; S.base[0] = 0; S.len[0] = (uint32_t)c; S.base[1] = (uint32_t)c; S.len[1] = (uint32_t)m;
  const uint32_t t0b = (uint32_t)((c + VIEW_SORT_TILE - 1) / VIEW_SORT_TILE), t1b = (uint32_t)((m + VIEW_SORT_TILE - 1) / VIEW_SORT_TILE);
  S.blk0[0] = 0; S.blk0[1] = t0b; S.blk0[2] = t0b + t1b;
  hipLaunchKernelGGL((k_view_tile_sort<T>), dim3(t0b + t1b), dim3(VIEW_SORT_THREADS), 0, st, (const T*)kv[0], (const uint32_t*)kp[0], kv[1], kp[1], S);
  int cur = 1;
  ViewSegs P = S; P.blk0[1] = (uint32_t)((c + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS); P.blk0[2] = P.blk0[1] + (uint32_t)((m + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS);
  for (uint64_t L = VIEW_SORT_TILE; L < std::max<uint64_t>(c, m); L *= 2) {
    hipLaunchKernelGGL((k_view_merge_pass<T>), dim3(P.blk0[2]), dim3(256), 0, st, (const T*)kv[cur], (const uint32_t*)kp[cur], kv[cur ^ 1], kp[cur ^ 1], P, (uint32_t)L);
    cur ^= 1;
  }
  const T* Dv = kv[cur]; const uint32_t* Dp = kp[cur]; const T* Iv = kv[cur] + c; const uint32_t* Ip = kp[cur] + c;
  (void)hipMemsetAsync(&ctx->ds->view_err, 0, sizeof(uint32_t), st);
  ViewRun<T> X{static_cast<T*>(ix.s_val), ix.s_pos, ix.s_ids};
  auto finish = [&]() -> int {       // the error word comes back; 0 = everything enqueued above did what it should
    hipError_t e = hipGetLastError();
    ctx->hres[HRES_ERR] = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<unsigned long long*>(&ctx->hres[HRES_ERR]), &ctx->ds->view_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return (e != hipSuccess || (uint32_t)ctx->hres[HRES_ERR]) ? 1 : 0;
  };
  auto rewrite_main = [&](const T* dv, const uint32_t* dp, uint64_t nd, const T* yv, const uint32_t* yp, uint64_t ny) -> bool {   // step 3
    const uint64_t nz = nx - nd + ny;
    if (!ensure_view_spare<T>(ctx, ix, nz)) return false;
    ViewRun<T> Z{static_cast<T*>(ix.s_val2), ix.s_pos2, ix.s_ids2};
    launch_run_merge<T, true>(ctx, X, nx, dv, dp, nd, yv, yp, ny, (const uint64_t*)ix.ids, Z);
    if (finish()) return false;
    std::swap(ix.s_val, ix.s_val2); std::swap(ix.s_pos, ix.s_pos2); std::swap(ix.s_ids, ix.s_ids2); std::swap(ix.ord_cap, ix.ord_cap2);
    ix.ord_n = nz; ix.ord_merges++;
    return true;
  };
  const bool have = ix.npd + ix.npi > 0;
  if (!ctx->view_pending || (!have && ktot > thr_direct)) {
    // the run is a large part of the view (joining it to a patch would cost what the rewrite costs), or the pending patch is switched off (BMX_VIEW_PENDING=0).
    // A run between thr and thr_direct keys joins the (empty) patch and makes the rewrite due at once: the same work, but BEHIND the answer
    if (have) return soft(__LINE__);
    if (c > nx || !rewrite_main(Dv, Dp, c, Iv, Ip, m)) return soft(__LINE__);
  } else {
    // room for the patch at its largest (a rewrite falls due beyond thr keys; the run that crosses the line is still taken in), allocated once: a growing
    // buffer would put its reallocation in front of some query's answer
    if (ensure_pending<T>(ctx, ix, std::max<uint64_t>(std::max<uint64_t>(ix.npd + c, ix.npi + m), thr + 2 * std::max<uint64_t>(c, m)))) return soft(__LINE__);
    if (!ctx->view_err_host && hipHostMalloc(reinterpret_cast<void**>(&ctx->view_err_host), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->view_err_host = nullptr; }
    if (!ctx->view_ev && hipEventCreateWithFlags(&ctx->view_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->view_ev = nullptr; }
    if (!ix.s_val2 && !ensure_view_spare<T>(ctx, ix, nx + thr + 2 * m)) return soft(__LINE__);     // (the rewrite's target, also allocated now rather than in front of a later answer)
    T* pdv[2] = {static_cast<T*>(ix.pd_v[0]), static_cast<T*>(ix.pd_v[1])}; T* piv[2] = {static_cast<T*>(ix.pi_v[0]), static_cast<T*>(ix.pi_v[1])};
    auto merge2 = [&](bool ids, ViewRun<T> A, uint64_t la, const T* bv, const uint32_t* bp, uint64_t lb, ViewRun<T> Z) {
      if (la + lb == 0) return;
      const uint32_t g = (uint32_t)((la + lb + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS);
      if (ids) hipLaunchKernelGGL((k_view_merge2<T, true>), dim3(g), dim3(256), 0, st, A, (uint32_t)la, bv, bp, (uint32_t)lb, (const uint64_t*)ix.ids, Z);
      else hipLaunchKernelGGL((k_view_merge2<T, false>), dim3(g), dim3(256), 0, st, A, (uint32_t)la, bv, bp, (uint32_t)lb, (const uint64_t*)nullptr, Z);
    };
    if (!have) {
      const int dc = ix.pcur, ic = ix.icur;
      (void)hipMemcpyAsync(pdv[dc], Dv, c * sizeof(T), hipMemcpyDeviceToDevice, st); (void)hipMemcpyAsync(ix.pd_p[dc], Dp, c * 4, hipMemcpyDeviceToDevice, st);
      ViewRun<T> none{nullptr, nullptr, nullptr}, Zi{piv[ic], ix.pi_p[ic], ix.pi_ids[ic]};
      merge2(true, none, 0, Iv, Ip, m, Zi);                             // (the inserted keys with their ids)
      if (hipGetLastError() != hipSuccess) return soft(__LINE__);       // (nothing here raises the error word, and everything that reads the patch is behind it on this stream: no synchronisation)
      ix.npd = c; ix.npi = m;
    } else {
      // 2. which deleted keys are pending inserted keys (they cancel), which are keys of main (they join pd)? one flag per key, two ordered selects by flag
      T* sel_v = kv[cur ^ 1]; uint32_t* sel_p = kp[cur ^ 1];              // the sort's other buffer: [0, c) keys of main, [c, 2c) pending inserted keys, behind them the flags
      uint8_t* flag = reinterpret_cast<uint8_t*>(sel_v + 2 * c);
      const int dc = ix.pcur, ic = ix.icur;
      unsigned long long hc[2] = {0, 0};
      if (c) {
        ctx->hres[HRES_SPLIT] = ctx->hres[HRES_SPLIT + 1] = ~0ull;
        (void)hipMemsetAsync(ix.pi_dead, 0, ix.npi, st);
        hipLaunchKernelGGL((k_view_flag_in<T>), dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, st, Dv, Dp, (uint32_t)c, (const T*)piv[ic], (const uint32_t*)ix.pi_p[ic], (uint32_t)ix.npi, flag, ix.pi_dead);
        SelGeom g = sel_geom<1>(c);
        for (uint32_t want = 0; want < 2; want++) {
          PredFlag PF{flag, want};
          EmitKeys<T> EK{Dv, Dp, sel_v + (want ? c : 0), sel_p + (want ? c : 0)};
          FinishCount FC{const_cast<unsigned long long*>(&ctx->hres[HRES_SPLIT + want])};
          hipLaunchKernelGGL((k_sel_count<PredFlag>), dim3(g.blocks), dim3(SEL_THREADS), 0, st, PF, c, g.tiles_per_block, ctx->block_counts);
          hipLaunchKernelGGL((k_sel_write<PredFlag, EmitKeys<T>, FinishCount>), dim3(g.blocks), dim3(SEL_THREADS), 0, st, PF, EK, FC, c, g.tiles_per_block, ctx->block_counts);
        }
        if (hipStreamSynchronize(st) != hipSuccess) return soft(__LINE__);
        hc[0] = ctx->hres[HRES_SPLIT]; hc[1] = ctx->hres[HRES_SPLIT + 1];
        if (hc[0] + hc[1] != c || hc[1] > ix.npi) return soft(__LINE__);
      }
      const uint64_t cX = hc[0], cI = hc[1];
      // pi' = pi - (deleted keys that were pending inserts) + inserted keys;  pd' = pd + (deleted keys of main): balanced two-run merges (k_view_merge2)
      int ia = ic; uint64_t na = ix.npi;
      if (cI) {                                                           // the cancelled inserts leave pi: an ordered select into the other set
        PredFlag PN{(const uint8_t*)ix.pi_dead, 0u};                           // (k_view_flag_in marked them while it looked the deleted keys up)
        EmitRun<T> ER{(const T*)piv[ic], (const uint32_t*)ix.pi_p[ic], (const uint64_t*)ix.pi_ids[ic], piv[ic ^ 1], ix.pi_p[ic ^ 1], ix.pi_ids[ic ^ 1]};
        FinishCount FC{&ctx->ds->view_tmp[0]};
        SelGeom g = sel_geom<1>(ix.npi);
        hipLaunchKernelGGL((k_sel_count<PredFlag>), dim3(g.blocks), dim3(SEL_THREADS), 0, st, PN, ix.npi, g.tiles_per_block, ctx->block_counts);
        hipLaunchKernelGGL((k_sel_write<PredFlag, EmitRun<T>, FinishCount>), dim3(g.blocks), dim3(SEL_THREADS), 0, st, PN, ER, FC, ix.npi, g.tiles_per_block, ctx->block_counts);
        ia = ic ^ 1; na = ix.npi - cI;
      }
      ViewRun<T> Ai{piv[ia], ix.pi_p[ia], ix.pi_ids[ia]}, Zi{piv[ia ^ 1], ix.pi_p[ia ^ 1], ix.pi_ids[ia ^ 1]};
      if (m) merge2(true, Ai, na, Iv, Ip, m, Zi);
      ViewRun<T> Ad{pdv[dc], ix.pd_p[dc], nullptr}, Zd{pdv[dc ^ 1], ix.pd_p[dc ^ 1], nullptr};
      if (cX) merge2(false, Ad, ix.npd, (const T*)sel_v, (const uint32_t*)sel_p, cX, Zd);
      if (hipGetLastError() != hipSuccess) return soft(__LINE__);       // (nothing here raises the error word, and everything that reads the patch is behind it on this stream: no synchronisation)
      ix.icur = m ? ia ^ 1 : ia; if (cX) ix.pcur = dc ^ 1;
      ix.npd += cX; ix.npi = na + m;
    }
    // 3. the pending patch has grown: main will be rewritten BEHIND the answer of the query that brought this refresh about (view_after_query), not in front of it
    if (ix.npd + ix.npi > thr) ix.rewrite_due = true;
  }
  ix.ord_patches++; ix.ord_patched_keys += ktot;
  ix.last_patch_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// Bring EVERY maintained index up to date from the change log (they share it), then forget the log. Per index: created rows of its field are
// appended in log order, then every logged row of the field gets its current value. One host sync at the end (appended counts, wide flags).
// An index whose value-ordered view is current goes on being current: the refresh captures the change run and the view is patched with it.
int refresh_from_log(bmx_ctx* ctx) {
  const bool dbg_t = std::getenv("BMX_VIEW_DEBUG") != nullptr;
  const auto dbg_t0 = std::chrono::steady_clock::now();
  auto dbg_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count(); };
  double dbg_sync = 0;
  const unsigned long long* n_dev = &ctx->ds->chg_n[ctx->chg_par];
  const uint64_t ub = ctx->chg_ub;
  struct Res { unsigned long long added; uint32_t wide; uint32_t changed; unsigned long long run; };     // (wide, changed: the two halves of one result word)
  std::vector<Res> res(ctx->indexes.size());
  std::vector<char> capture(ctx->indexes.size(), 0);
  // results of index k live in its own scratch words: part_totals[] is free between partitions (k < PART_MAX_SHARDS indexes are maintained)
  if (ctx->indexes.size() > IX_MAINTAINED_MAX) return fail(ctx, BMX_ERR_INTERNAL, "index maintenance with more indexes than result words");
  if (ub) {
    if (!ensure_hres(ctx)) return fail(ctx, BMX_ERR_NOMEM, "index maintenance: no page-locked memory for the result words");
    for (size_t k = 0; k < ctx->indexes.size(); k++) ctx->hres[HRES_RUN + k] = ~0ull;
    for (size_t k = 0; k < ctx->indexes.size(); k++) {
      Index& ix = ctx->indexes[k];
      // the view is current and can stay so: capture the change run (needs room for one entry per log entry)
      if (ctx->view_patching && ix.ordered_after && ix.s_val && ix.ord_content == ix.content && ix.ord_fits32 == ix.fits32 && ix.n && ix.n < 0xFFFFFFFFull && ub <= VIEW_PATCH_MAX_LOG) {
        if (ix.cl_cap < ub) {
          HIPCHK(hipStreamSynchronize(ctx->stream));
          for (void* q : {(void*)ix.cl_pos, (void*)ix.cl_old, (void*)ix.cl2_pos, (void*)ix.cl2_old}) if (q) (void)hipFree(q);
          ix.cl_pos = nullptr; ix.cl_old = nullptr; ix.cl2_pos = nullptr; ix.cl2_old = nullptr; ix.cl_cap = 0;
          const uint64_t cap = (ub + ub / 2 + (1u << 16) + 255) & ~255ull;
          if (hipMalloc(reinterpret_cast<void**>(&ix.cl_pos), cap * sizeof(uint32_t)) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&ix.cl_old), cap * sizeof(int64_t)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&ix.cl2_pos), cap * sizeof(uint32_t)) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&ix.cl2_old), cap * sizeof(int64_t)) == hipSuccess) ix.cl_cap = cap;
          else {
            (void)hipGetLastError();
            for (void* q : {(void*)ix.cl_pos, (void*)ix.cl_old, (void*)ix.cl2_pos, (void*)ix.cl2_old}) if (q) (void)hipFree(q);
            ix.cl_pos = nullptr; ix.cl_old = nullptr; ix.cl2_pos = nullptr; ix.cl2_old = nullptr;
          }
        }
        capture[k] = ix.cl_cap >= ub;
      }
      unsigned long long* d_added = &ctx->ds->part_totals[2 * k];
      uint32_t* d_wide = reinterpret_cast<uint32_t*>(&ctx->ds->part_totals[2 * k + 1]);
      HIPCHK(hipMemsetAsync(d_added, 0, 2 * sizeof(unsigned long long), ctx->stream));
      PredLogCreated P{ctx->chg, n_dev, ix.field, ctx->slot_pos};
      SelGeom g = sel_geom<PredLogCreated::E>(ub);
      hipLaunchKernelGGL((k_sel_count<PredLogCreated>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, ub, g.tiles_per_block, ctx->block_counts);
      LAUNCHCHK("k_sel_count(log)");
      EmitAppend Em{ctx->chg, ctx->slots, ix.ids, ix.v64, ix.v32, d_wide, ctx->slot_pos, ix.n, ix.cap};
      FinishCount Fin{d_added};
      hipLaunchKernelGGL((k_sel_write<PredLogCreated, EmitAppend, FinishCount>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, Em, Fin, ub, g.tiles_per_block,
                         ctx->block_counts);
      LAUNCHCHK("k_sel_write(log)");
      const uint32_t ublocks = (uint32_t)std::min<uint64_t>((ub + 255) / 256, 4096);
      hipLaunchKernelGGL(k_ix_update, dim3(ublocks), dim3(256), 0, ctx->stream, (const uint2*)ctx->chg, n_dev, (const Slot*)ctx->slots, ix.field, (const uint32_t*)ctx->slot_pos,
                         ix.v64, ix.v32, d_wide, capture[k] ? 2u : (ix.ordered_after ? 1u : 0u), ix.cl_pos, ix.cl_old, (uint64_t)ix.cl_cap);
      LAUNCHCHK("k_ix_update");
      if (capture[k]) {      // the change run without its holes, in log order (ordered select: no atomics), and its length
        PredChanged PC{ix.cl_pos, n_dev};
        SelGeom gc = sel_geom<PredChanged::E>(ub);
        hipLaunchKernelGGL((k_sel_count<PredChanged>), dim3(gc.blocks), dim3(SEL_THREADS), 0, ctx->stream, PC, ub, gc.tiles_per_block, ctx->block_counts);
        EmitChanged EC{ix.cl_pos, ix.cl_old, ix.cl2_pos, ix.cl2_old};
        FinishCount FC{const_cast<unsigned long long*>(&ctx->hres[HRES_RUN + k])};
        hipLaunchKernelGGL((k_sel_write<PredChanged, EmitChanged, FinishCount>), dim3(gc.blocks), dim3(SEL_THREADS), 0, ctx->stream, PC, EC, FC, ub, gc.tiles_per_block, ctx->block_counts);
        LAUNCHCHK("k_sel_write(change run)");
      }
    }
    // one copy of the indexes' (added, wide | changed) words into the mapped result words; the change runs' lengths were written there by their selects
    HIPCHK(hipMemcpyAsync(const_cast<unsigned long long*>(&ctx->hres[HRES_TOTALS]), ctx->ds->part_totals, 2 * ctx->indexes.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dbg_sync = dbg_us();
    for (size_t k = 0; k < ctx->indexes.size(); k++) {
      res[k].added = ctx->hres[HRES_TOTALS + 2 * k];
      const unsigned long long wc = ctx->hres[HRES_TOTALS + 2 * k + 1];
      res[k].wide = (uint32_t)wc; res[k].changed = (uint32_t)(wc >> 32);
      res[k].run = capture[k] ? ctx->hres[HRES_RUN + k] : 0;
    }
  }
  int rc = reset_chg_log(ctx);
  if (rc) return rc;
  for (size_t k = 0; k < ctx->indexes.size(); k++) {
    Index& ix = ctx->indexes[k];
    if (ub && ix.n + res[k].added > ix.cap) {
      // The appended rows did not fit. The entries it missed are gone with the log, so this index must never be refreshed from a LATER log:
      // without its positions it can only come back through build_index(), and the log stops until every index is fresh again.
      ix.version = ~0ull; ix.has_pos = false; ctx->chg_valid = false;
      continue;
    }
    if (ub) {
      const uint64_t n0 = ix.n;
      ix.n += res[k].added; if (res[k].wide) ix.fits32 = false;
      const bool moved = res[k].added || res[k].changed || !ix.ordered_after;     // (no view: nobody compared, nobody cares)
      if (moved) ix.content++;
      if (moved && capture[k] && ix.fits32 == ix.ord_fits32 && res[k].run <= ix.cl_cap && ix.n < 0xFFFFFFFFull) {
        const int prc = ix.ord_fits32 ? patch_view_t<int32_t>(ctx, ix, res[k].run, n0, res[k].added) : patch_view_t<int64_t>(ctx, ix, res[k].run, n0, res[k].added);
        if (prc == 0) ix.ord_content = ix.content;       // the view equals a fresh sort of the columns as they are now
      }
    }
    ix.version = ctx->version;
  }
  ctx->ix_incremental++;
  if (dbg_t) std::fprintf(stderr, "bmx: refresh from the log: columns up to date after %.1f us, patches done after %.1f us\n", dbg_sync, dbg_us());
  return BMX_OK;
}

int fresh_index(bmx_ctx* ctx, uint32_t field, Index** out) {
  Index* ix = find_index(ctx, field);
  if (!ix) {  // equals()/range() auto-create a missing index: src/bullet-query.js:194-196, 230-232
    if (ctx->indexes.size() >= IX_MAINTAINED_MAX) ctx->chg_valid = false;   // more indexes than the maintenance pass has result words for: they are rebuilt when stale
    ctx->indexes.emplace_back();
    ix = &ctx->indexes.back();
    ix->field = field;
  }
  if (ix->version != ctx->version) {
    // maintained: every index has its positions recorded, the log is complete, and it is shorter than a quarter of the table
    // (beyond that the rebuild's two sequential passes over the table are cheaper than the log's random accesses)
    bool inc = ctx->chg_valid && ix->has_pos && ctx->chg_ub <= std::max<uint64_t>(ctx->nslots / 8, 1u << 20);
    if (inc) for (auto& o : ctx->indexes) inc = inc && (o.has_pos || &o == ix);
    if (inc) {
      int rc = refresh_from_log(ctx);
      if (rc) return rc;
    }
    if (ix->version != ctx->version) {   // not maintained (or its appended rows did not fit): rebuild from the table
      ctx->chg_valid = false;
      int rc = build_index(ctx, ix);
      if (rc) return rc;
    }
  }
  *out = ix;
  return BMX_OK;
}

int ensure_scan_out(bmx_ctx* ctx, uint64_t n) {
  if (n <= ctx->scan_cap) return BMX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dev_free(ctx->scan_out);
  ctx->scan_cap = 0;
  int rc = dev_alloc(ctx, &ctx->scan_out, n);
  if (rc) return rc;
  ctx->scan_cap = n;
  return BMX_OK;
}

// scratch of the scans: one match bit per index row + one count per 8192-row block (+1 for the total)
int ensure_scan_scratch(bmx_ctx* ctx, uint64_t n) {
  uint64_t nb = (n + SCAN_BLOCK_ELEMS - 1) / SCAN_BLOCK_ELEMS;
  if (nb <= ctx->scan_blocks_cap) return BMX_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dev_free(ctx->scan_mask); dev_free(ctx->scan_counts);
  ctx->scan_blocks_cap = 0;
  uint64_t cap = nb + nb / 4 + 16;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->scan_mask, cap * (SCAN_BLOCK_ELEMS / 32))) || (rc = dev_alloc(ctx, &ctx->scan_counts, cap + 1))) return rc;
  ctx->scan_blocks_cap = cap;
  return BMX_OK;
}

// Run one predicate over an index and deliver ids (POS = false: u64 node ids gathered from the id column) or index positions (POS = true: u32,
// no gather) / the count according to `mem`. `out` is uint64_t* or uint32_t* accordingly.
// a value column above this size is read with nontemporal loads: it cannot stay in the 256 MiB Infinity Cache between two scans anyway (scan_kernels.h)
constexpr uint64_t SCAN_NT_BYTES = 256ull << 20;
constexpr uint32_t SCAN_NTX_DEFAULT = 0;      // EmitIds::ntx (profiles/r05_scan_nt_emit_ab.log)

// ---- value-ordered view (bmx.h bmx_index_set_ordered) ----
// Is the view of `ix` usable for the query at hand? A stale one is sorted again by the ordered_after-th query since the columns last changed — the
// queries in front of that one scan the column as ever (one sort of a 100M-row column costs what ~20 scans cost) —, and never while it cannot be had
// (no memory: the index goes on without it). Synchronous where it sorts.
bool ensure_ordered_view(bmx_ctx* ctx, Index* ix) {
  if (!ix->ordered_after || ix->n == 0 || ix->n > 0xFFFFFFFFull) return false;
  if (ix->ord_content == ix->content && ix->s_val) { view_before_use(ctx, ix, /*wait=*/false); return true; }   // (a finished rewrite becomes main; an unfinished one changes nothing)
  view_before_use(ctx, ix, /*wait=*/true);
  uint32_t after = ix->ordered_after;
  if (after == BMX_INDEX_ORDERED_AUTO) {
    // rent or buy: sort once the scans answered since the change have cost what a sort costs — then whatever the caller does next, at most twice
    // the best possible was spent. A scan moves the value column at ~6 TB/s (+ two launches); a sort costs what the last one cost (first time: 60 us per 10^6 rows).
    const double scan_us = 8.0 + (double)ix->n * (ix->fits32 ? 4.0 : 8.0) / 6.0e6;
    const double sort_us = ix->last_sort_us > 0 ? ix->last_sort_us : 200.0 + (double)ix->n * 0.00006;
    after = (uint32_t)std::min<double>(1.0e6, std::max<double>(2.0, std::ceil(sort_us / scan_us)));
  }
  if (ix->stale_content != ix->content) { ix->stale_content = ix->content; ix->stale_queries = 0; }   // the count starts with every change of the columns
  if (++ix->stale_queries < after) return false;
  const auto t_sort = std::chrono::steady_clock::now();
  ix->npd = ix->npi = 0;                          // a fresh sort of the columns: whatever patch was pending is in them
  const uint64_t n = ix->n;
  const size_t vb = ix->fits32 ? sizeof(int32_t) : sizeof(int64_t);
  auto give_up = [&]() { (void)hipGetLastError(); free_ordered_view(*ix); ix->stale_queries = 0; return false; };
  if (n > ix->ord_cap || ix->ord_fits32 != ix->fits32) {
    free_ordered_view(*ix);
    const uint64_t cap = n + n / 8 + 1024;
    if (hipMalloc(&ix->s_val, cap * vb) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&ix->s_pos), cap * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ix->s_ids), cap * sizeof(uint64_t)) != hipSuccess) return give_up();
    ix->ord_cap = cap; ix->ord_fits32 = ix->fits32;
  }
  if (ctx->view_own_sort && n && n < 0xFFFFFFFFull) {
    // A/B arm: the whole column through the patch path's sort — (value, position) keys, 4096-key tiles in LDS, then log2(n / 4096) merge-path passes between
    // the view's columns and a scratch pair; a tombstone is the column type's minimum and sorts in front like every other value
    void* tv = nullptr; uint32_t* tp = nullptr;
    if (hipMalloc(&tv, n * vb) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&tp), n * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); if (tv) (void)hipFree(tv); return give_up(); }
    const uint32_t gbo = (uint32_t)std::min<uint64_t>((n + 255) / 256, 8192);
    unsigned passes = 0; for (uint64_t L = VIEW_SORT_TILE; L < n; L *= 2) passes++;
    auto run = [&](auto tag) {
      using T = decltype(tag);
      const T* col = sizeof(T) == 4 ? reinterpret_cast<const T*>(ix->v32) : reinterpret_cast<const T*>(ix->v64);
      T* bufv[2] = {static_cast<T*>(ix->s_val), static_cast<T*>(tv)}; uint32_t* bufp[2] = {ix->s_pos, tp};
      int cur = passes & 1;                                             // the tile sort writes into the buffer from which `passes` swaps end in the view's columns
      uint32_t* iota = bufp[cur ^ 1];                                   // (the other position buffer is free until the first pass writes it)
      hipLaunchKernelGGL(k_iota_u32, dim3(gbo), dim3(256), 0, ctx->stream, iota, n);
      ViewSegs S{}; S.base[0] = 0; S.len[0] = (uint32_t)n; S.base[1] = (uint32_t)n; S.len[1] = 0;
      const uint32_t tiles = (uint32_t)((n + VIEW_SORT_TILE - 1) / VIEW_SORT_TILE);
      S.blk0[0] = 0; S.blk0[1] = tiles; S.blk0[2] = tiles;
      hipLaunchKernelGGL((k_view_tile_sort<T>), dim3(tiles), dim3(VIEW_SORT_THREADS), 0, ctx->stream, col, (const uint32_t*)iota, bufv[cur], bufp[cur], S);
      ViewSegs P = S; P.blk0[1] = (uint32_t)((n + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS); P.blk0[2] = P.blk0[1];
      for (uint64_t L = VIEW_SORT_TILE; L < n; L *= 2) {
        hipLaunchKernelGGL((k_view_merge_pass<T>), dim3(P.blk0[2]), dim3(256), 0, ctx->stream, (const T*)bufv[cur], (const uint32_t*)bufp[cur], bufv[cur ^ 1], bufp[cur ^ 1], P, (uint32_t)L);
        cur ^= 1;
      }
    };
    if (ix->fits32) run(int32_t{}); else run(int64_t{});
    hipLaunchKernelGGL(k_view_gather_ids, dim3(gbo), dim3(256), 0, ctx->stream, (const uint32_t*)ix->s_pos, (uint32_t)n, (const uint64_t*)ix->ids, ix->s_ids);
    hipError_t eo = hipGetLastError();
    if (eo == hipSuccess) eo = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tv); (void)hipFree(tp);
    if (eo != hipSuccess) return give_up();
    ix->ord_n = n; ix->ord_content = ix->content; ix->stale_queries = 0; ix->ord_sorts++;
    ix->last_sort_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_sort).count();
    return true;
  }
  // the column's value range decides how many bits the sort has to look at (csrc/ordered_sort.hip: keys rebased to min = 1, tombstones = 0)
  const uint32_t gb = (uint32_t)std::min<uint64_t>((n + 255) / 256, 8192);
  long long mm[2] = {INT64_MAX, INT64_MIN};
  long long* d_mm = reinterpret_cast<long long*>(ctx->ds->ord_ab);
  hipError_t e = hipMemcpyAsync(d_mm, mm, sizeof(mm), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    if (ix->fits32) hipLaunchKernelGGL((k_col_minmax<int32_t>), dim3(std::min<uint32_t>(gb, 2048)), dim3(256), 0, ctx->stream, (const int32_t*)ix->v32, n, d_mm);
    else hipLaunchKernelGGL((k_col_minmax<int64_t>), dim3(std::min<uint32_t>(gb, 2048)), dim3(256), 0, ctx->stream, (const int64_t*)ix->v64, n, d_mm);
    e = hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return give_up();
  if (mm[0] > mm[1]) { mm[0] = 0; mm[1] = 0; }                   // nothing but tombstones
  const unsigned long long span = (unsigned long long)mm[1] - (unsigned long long)mm[0] + 1ull;     // largest rebased key
  unsigned bits = 1; while (bits < 64 && (span >> bits)) bits++;
  bits = std::min<unsigned>(bits, ix->fits32 ? 32u : 64u);
  uint32_t* iota = nullptr; void* tmp = nullptr; size_t tmp_bytes = 0;
  e = ix->fits32 ? sort_pairs_i32(nullptr, &tmp_bytes, nullptr, 0, bits, nullptr, nullptr, nullptr, n, ctx->stream)
                 : sort_pairs_i64(nullptr, &tmp_bytes, nullptr, 0, bits, nullptr, nullptr, nullptr, n, ctx->stream);
  if (e != hipSuccess || hipMalloc(reinterpret_cast<void**>(&iota), n * sizeof(uint32_t)) != hipSuccess || hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)) != hipSuccess) {
    if (iota) (void)hipFree(iota);
    return give_up();
  }
  hipLaunchKernelGGL(k_iota_u32, dim3(gb), dim3(256), 0, ctx->stream, iota, n);
  e = ix->fits32 ? sort_pairs_i32(tmp, &tmp_bytes, ix->v32, (int32_t)mm[0], bits, static_cast<uint32_t*>(ix->s_val), iota, ix->s_pos, n, ctx->stream)
                 : sort_pairs_i64(tmp, &tmp_bytes, ix->v64, (int64_t)mm[0], bits, static_cast<uint64_t*>(ix->s_val), iota, ix->s_pos, n, ctx->stream);
  if (e == hipSuccess) {
    if (ix->fits32) hipLaunchKernelGGL((k_gather_ids<int32_t, uint32_t>), dim3(gb), dim3(256), 0, ctx->stream, (const uint64_t*)ix->ids, (const uint32_t*)ix->s_pos, ix->s_ids, n, static_cast<uint32_t*>(ix->s_val), (int32_t)mm[0]);
    else hipLaunchKernelGGL((k_gather_ids<int64_t, uint64_t>), dim3(gb), dim3(256), 0, ctx->stream, (const uint64_t*)ix->ids, (const uint32_t*)ix->s_pos, ix->s_ids, n, static_cast<uint64_t*>(ix->s_val), (int64_t)mm[0]);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // the scratch goes back below
  (void)hipFree(iota); (void)hipFree(tmp);
  if (e != hipSuccess) return give_up();
  ix->ord_n = n; ix->ord_content = ix->content; ix->stale_queries = 0; ix->ord_sorts++;
  ix->last_sort_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_sort).count();
  return true;
}
// the query itself: two searches + one contiguous copy; lo/hi are already clamped like the scans' (tombstones sort in front of every legal value)
template <bool POS, class OutT, class T>
void launch_ordered_t(bmx_ctx* ctx, const Index* ix, T l, T h, OutT* d_out, uint64_t d_cap, unsigned long long* d_n, const PredFilter* filter) {
  unsigned long long* ab = ctx->ds->ord_ab;
  const T* sv = static_cast<const T*>(ix->s_val);
  const bool pending = ix->npd + ix->npi > 0;           // the logical view = main - pd + pi (patch_view_t)
  const int q = ix->pcur, qi = ix->icur;
  const T* dv = static_cast<const T*>(ix->pd_v[q]); const T* iv = static_cast<const T*>(ix->pi_v[qi]);
  if (pending) hipLaunchKernelGGL((k_ordered_bounds_p<T>), dim3(1), dim3(384), 0, ctx->stream, sv, ix->ord_n, dv, ix->npd, iv, ix->npi, l, h, ab, d_n, filter ? 1u : 0u);
  else hipLaunchKernelGGL((k_ordered_bounds<T>), dim3(1), dim3(128), 0, ctx->stream, sv, ix->ord_n, l, h, ab, d_n, filter ? 1u : 0u);
  if (filter) {             // every candidate of the run is looked at whatever the caller can take: the count is the number of survivors
    if constexpr (!POS) {
      const uint32_t fb = (uint32_t)std::min<uint64_t>((ix->ord_n + ix->npi + 2047) / 2048, 4096);
      if (pending) hipLaunchKernelGGL((k_ordered_filter_p<T, PredFilter>), dim3(fb), dim3(256), 0, ctx->stream, sv, (const uint32_t*)ix->s_pos, (const uint64_t*)ix->s_ids, dv, (const uint32_t*)ix->pd_p[q],
                                      (const uint64_t*)ix->pi_ids[qi], (const unsigned long long*)ab, *filter, d_out, d_out ? d_cap : 0, d_n);
      else hipLaunchKernelGGL((k_ordered_filter<PredFilter>), dim3(fb), dim3(256), 0, ctx->stream, (const uint64_t*)ix->s_ids, (const unsigned long long*)ab, *filter, d_out, d_out ? d_cap : 0, d_n);
    }
    return;
  }
  if (!d_out || !d_cap) return;
  // the match count is the device's: a grid for the most the caller can take, whose workgroups beyond the matches leave at once
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((std::min<uint64_t>(d_cap, ix->ord_n + ix->npi) + 2047) / 2048, 8192);
  if (pending) {
    if constexpr (POS) hipLaunchKernelGGL((k_ordered_copy_p<T, uint32_t>), dim3(blocks), dim3(256), 0, ctx->stream, sv, (const uint32_t*)ix->s_pos, (const uint32_t*)ix->s_pos, dv, (const uint32_t*)ix->pd_p[q],
                                          (const uint32_t*)ix->pi_p[qi], (const unsigned long long*)ab, d_out, d_cap);
    else hipLaunchKernelGGL((k_ordered_copy_p<T, uint64_t>), dim3(blocks), dim3(256), 0, ctx->stream, sv, (const uint32_t*)ix->s_pos, (const uint64_t*)ix->s_ids, dv, (const uint32_t*)ix->pd_p[q],
                            (const uint64_t*)ix->pi_ids[qi], (const unsigned long long*)ab, d_out, d_cap);
  } else {
    if constexpr (POS) hipLaunchKernelGGL((k_ordered_copy<uint32_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)ix->s_pos, (const unsigned long long*)ab, d_out, d_cap);
    else hipLaunchKernelGGL((k_ordered_copy<uint64_t>), dim3(blocks), dim3(256), 0, ctx->stream, (const uint64_t*)ix->s_ids, (const unsigned long long*)ab, d_out, d_cap);
  }
}
template <bool POS, class OutT>
void launch_ordered(bmx_ctx* ctx, const Index* ix, int64_t lo, int64_t hi, OutT* d_out, uint64_t d_cap, unsigned long long* d_n, const PredFilter* filter = nullptr) {
  if (filter && !d_n) d_n = &ctx->ds->n_out;       // (the filter appends through a counter even when nobody asked for the count)
  if (ix->ord_fits32) {     // the view was sorted from the 4-byte column: bounds clamped into int32 like the scans' (an empty range stays empty)
    int64_t l = std::max<int64_t>(lo, (int64_t)INT32_MIN + 1), h = std::min<int64_t>(hi, INT32_MAX);
    if (lo > INT32_MAX || hi < INT32_MIN) { l = 1; h = 0; }
    launch_ordered_t<POS, OutT, int32_t>(ctx, ix, (int32_t)l, (int32_t)h, d_out, d_cap, d_n, filter);
  } else launch_ordered_t<POS, OutT, int64_t>(ctx, ix, lo, hi, d_out, d_cap, d_n, filter);
}

template <bool POS, class Pred>
int run_scan_t(bmx_ctx* ctx, const Pred& P, const Index* ix, void* out_v, uint64_t cap, uint64_t* n_out, int mem, bool ordered = false, int64_t olo = 0, int64_t ohi = 0) {
  using OutT = typename std::conditional<POS, uint32_t, uint64_t>::type;
  OutT* out_ids = static_cast<OutT*>(out_v);
  const bool host = mem == BMX_MEM_HOST;
  OutT* d_out = out_ids;
  uint64_t d_cap = cap;
  int rc;
  // small host-mode answers (count only, or room for at most SCAN_PIN_IDS ids) come back through mapped host memory: no download, one synchronisation
  constexpr uint64_t SCAN_PIN_IDS = 16384;
  static_assert(SCAN_PIN_IDS * 8 + 8 <= SMALL_OUT_BYTES, "pinned scan answer fits the small-call buffer");
  const bool pinned = host && !ctx->scan_defer && (!out_ids || std::min<uint64_t>(cap, ix->n) <= SCAN_PIN_IDS) && ensure_pinned(ctx);
  bool direct_host = false;
  if (host && out_ids) {
    d_cap = std::min<uint64_t>(cap, ix->n);
    if (pinned) d_out = reinterpret_cast<OutT*>(ctx->pin_out);
    else {
      // a caller's buffer in page-locked memory (bmx_host_alloc, hipHostMalloc, a registered range) is written by the kernels themselves: no staging copy behind the answer
      hipPointerAttribute_t at{};
      if (!ctx->scan_defer && hipPointerGetAttributes(&at, out_ids) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) { d_out = static_cast<OutT*>(at.devicePointer); direct_host = true; }
      else {
        (void)hipGetLastError();
        if ((rc = ensure_scan_out(ctx, std::max<uint64_t>(d_cap, 1)))) return rc;
        d_out = reinterpret_cast<OutT*>(ctx->scan_out);
      }
    }
  }
  if ((rc = ensure_scan_scratch(ctx, std::max<uint64_t>(ix->n, 1)))) return rc;
  const bool hres_n = host && !pinned && !ctx->scan_defer && !(ordered && std::is_same<Pred, PredFilter>::value) /* (that one counts with atomics) */ && ensure_hres(ctx);       // the count of a larger host-mode answer: a mapped result word
  unsigned long long* d_n = host ? (pinned ? reinterpret_cast<unsigned long long*>(ctx->pin_out + SCAN_PIN_IDS * 8) : hres_n ? const_cast<unsigned long long*>(&ctx->hres[HRES_SCAN_N]) : &ctx->ds->n_out)
                                 : reinterpret_cast<unsigned long long*>(n_out);
  const uint32_t nb = (uint32_t)((std::max<uint64_t>(ix->n, 1) + SCAN_BLOCK_ELEMS - 1) / SCAN_BLOCK_ELEMS);
  hipEvent_t* se = (ctx->prof_on && ctx->scan_prof_n < PROF_MAX_CALLS && !ctx->scan_ev.empty()) ? &ctx->scan_ev[3 * ctx->scan_prof_n] : nullptr;
  if (se) HIPCHK(hipEventRecord(se[0], ctx->stream));
  if (ordered) {
    if constexpr (std::is_same<Pred, PredFilter>::value) launch_ordered<POS>(ctx, ix, olo, ohi, d_out, d_cap, d_n, &P);
    else launch_ordered<POS>(ctx, ix, olo, ohi, d_out, d_cap, d_n);
    LAUNCHCHK("k_ordered_bounds / k_ordered_copy");
    if (se) HIPCHK(hipEventRecord(se[1], ctx->stream));
  } else if (d_out) {
    // pass 1: one read of the column -> match mask + block counts; pass 2: ids / positions from the mask
    hipLaunchKernelGGL((k_scan_mask<Pred, true>), dim3(nb), dim3(SEL_THREADS), 0, ctx->stream, P, ix->n, ctx->scan_mask, ctx->scan_counts);
    LAUNCHCHK("k_scan_mask");
    if (se) HIPCHK(hipEventRecord(se[1], ctx->stream));
    typename std::conditional<POS, EmitPos, EmitIds>::type Em;
    if constexpr (POS) Em = EmitPos{d_out, d_cap};
    else {
      const char* sm = std::getenv("BMX_SCAN_STREAM_MIN");      // measurement switch: matches per block from which the id column is streamed (0xFFFFFFFF: never)
      const char* nx = std::getenv("BMX_SCAN_NT");               // measurement switch: EmitIds::ntx
      Em = EmitIds{ix->ids, d_out, d_cap, ix->n * sizeof(uint64_t) > SCAN_NT_BYTES, sm ? (uint32_t)std::strtoul(sm, nullptr, 0) : SCAN_STREAM_MIN, nx ? (uint32_t)std::strtoul(nx, nullptr, 0) : SCAN_NTX_DEFAULT};
    }
    using EmT = decltype(Em);
    FinishCount Fin{d_n};
    if (nb > SCAN_SUB8_BLOCKS)   // large column: an eighth of the workgroups, each sums the counts in front of it once (no offsets launch)
      hipLaunchKernelGGL((k_scan_emit<EmT, FinishCount, 8>), dim3((nb + 7) / 8), dim3(SEL_THREADS), 0, ctx->stream, ctx->scan_mask, ctx->scan_counts, ix->n, nb, Em, Fin);
    else
      hipLaunchKernelGGL((k_scan_emit<EmT, FinishCount, 1>), dim3(nb), dim3(SEL_THREADS), 0, ctx->stream, ctx->scan_mask, ctx->scan_counts, ix->n, nb, Em, Fin);
    LAUNCHCHK("k_scan_emit");
  } else if (d_n) {
    hipLaunchKernelGGL((k_scan_mask<Pred, false>), dim3(nb), dim3(SEL_THREADS), 0, ctx->stream, P, ix->n, ctx->scan_mask, ctx->scan_counts);
    LAUNCHCHK("k_scan_mask(count)");
    if (se) HIPCHK(hipEventRecord(se[1], ctx->stream));
    hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(SEL_THREADS), 0, ctx->stream, ctx->scan_counts, nb, d_n);
    LAUNCHCHK("k_sum_counts");
  } else if (se) {
    HIPCHK(hipEventRecord(se[1], ctx->stream));
  }
  if (se) { HIPCHK(hipEventRecord(se[2], ctx->stream)); ctx->scan_prof_n++; }
  if (host && ctx->scan_defer) { ctx->scan_defer_cap = out_ids ? d_cap : 0; return BMX_OK; }   // the caller fetches with scan_collect()
  if (pinned) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const unsigned long long m = *d_n;
    if (out_ids && m) std::memcpy(out_ids, d_out, std::min<uint64_t>(m, d_cap) * sizeof(OutT));
    if (n_out) *n_out = m;
    return BMX_OK;
  }
  if (host) {
    unsigned long long m = 0;
    if (!hres_n) HIPCHK(hipMemcpyAsync(&m, &ctx->ds->n_out, sizeof(m), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (hres_n) m = ctx->hres[HRES_SCAN_N];
    if (out_ids && m && !direct_host) HIPCHK(hipMemcpy(out_ids, ctx->scan_out, std::min<uint64_t>(m, d_cap) * sizeof(OutT), hipMemcpyDeviceToHost));
    if (n_out) *n_out = m;
  }
  return BMX_OK;
}
template <class Pred>
int run_scan(bmx_ctx* ctx, const Pred& P, const Index* ix, uint64_t* out_ids, uint64_t cap, uint64_t* n_out, int mem) {
  return run_scan_t<false>(ctx, P, ix, out_ids, cap, n_out, mem);
}

// second half of a deferred host-mode scan: wait for the scan enqueued with ctx->scan_defer set, deliver the count and up to `cap` ids
int scan_collect(bmx_ctx* ctx, uint64_t* out_ids, uint64_t cap, uint64_t* n_out) {
  if (int erc = enter(ctx)) return erc;
  unsigned long long m = 0;
  HIPCHK(hipMemcpyAsync(&m, &ctx->ds->n_out, sizeof(m), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t k = std::min<uint64_t>(std::min<uint64_t>(m, ctx->scan_defer_cap), cap);
  if (out_ids && k) HIPCHK(hipMemcpy(out_ids, ctx->scan_out, k * 8, hipMemcpyDeviceToHost));
  if (n_out) *n_out = m;
  return BMX_OK;
}

template <bool POS>
int scan_range_impl_t(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, void* out, uint64_t cap, uint64_t* n_out, int mem) {
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  Index* ix;
  int rc = fresh_index(ctx, field, &ix);
  if (rc) return rc;
  const bool ordered = (n_out || out) && ensure_ordered_view(ctx, ix);     // (it was sorted from columns of the width they have now: a widened index has a new `content`)
  if (ix->fits32) {
    // every value fits int32: scan the 4-byte column with bounds clamped into int32 (an empty range stays empty). INT32_MIN itself is what a
    // tombstone looks like in this column and is never matched (a real -2^31 makes the index wide: scan_kernels.h v32_of)
    int64_t l = std::max<int64_t>(lo, (int64_t)INT32_MIN + 1), h = std::min<int64_t>(hi, INT32_MAX);
    if (lo > INT32_MAX || hi < INT32_MIN) { l = 1; h = 0; }
    PredRange32 P{ix->v32, (int32_t)l, (int32_t)h, ix->n * sizeof(int32_t) > SCAN_NT_BYTES};
    const int src = run_scan_t<POS>(ctx, P, ix, out, cap, n_out, mem, ordered, l, h);
    if (ordered && !src) view_after_query(ctx, ix);
    return src;
  }
  PredRange64 P{ix->v64, std::max<int64_t>(lo, -VAL_MAX), hi, ix->n * sizeof(int64_t) > SCAN_NT_BYTES};    // values live in +-(2^53-1): the clamp changes no answer and keeps tombstones (INT64_MIN) out
  const int src = run_scan_t<POS>(ctx, P, ix, out, cap, n_out, mem, ordered, P.lo, P.hi);
  if (ordered && !src) view_after_query(ctx, ix);
  return src;
}
int scan_range_impl(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap, uint64_t* n_out, int mem) {
  return scan_range_impl_t<false>(ctx, field, lo, hi, out_ids, cap, n_out, mem);
}

}  // namespace

extern "C" {

int bmx_abi_version(void) { return BMX_ABI_VERSION; }

const char* bmx_last_error(const bmx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

uint32_t bmx_owner_of(uint64_t id, uint32_t nshards) { return (uint32_t)(((unsigned __int128)owner_hash(id) * nshards) >> 64); }

// bmx.h "bmx_selfcheck". Synchronous for the caller, on a stream of its own (nothing else on the device is waited for); allocates and frees 128 KB + 4 words.
int bmx_selfcheck(int device, uint64_t* reads_out, uint64_t* torn_out, uint64_t* control_torn_out) {
  bmx_ctx* ctx = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, BMX_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(nullptr, BMX_ERR_INVALID, "bmx_selfcheck: device index out of range");
  HIPCHK(hipSetDevice(device));
  constexpr uint32_t NS = 4096, ITERS = 1500, BLOCKS = 1024;     // 4096 hot slots (128 KB: every line is written and read all the time), ~2*10^8 checked loads
  uint4* slots = nullptr; unsigned long long* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&slots), (size_t)NS * 32) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d), 4 * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError(); if (slots) (void)hipFree(slots);
    return fail(nullptr, BMX_ERR_NOMEM, "bmx_selfcheck: out of device memory");
  }
  unsigned long long h[4] = {0, 0, 0, 0};
  hipStream_t cs = nullptr;       // a stream of its own: no device-wide synchronisation, nothing of another context or library is waited for
  const char* nul = std::getenv("BMX_SELFCHECK_NULL_STREAM");     // measurement switch (round 5): the device's null stream, as rounds 1-4 used
  hipError_t e = (nul && nul[0] == '1') ? hipSuccess : hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMemsetAsync(slots, 0, (size_t)NS * 32, cs);
  if (e == hipSuccess) e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), cs);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_selfcheck_tear<false>, dim3(BLOCKS), dim3(256), 0, cs, slots, NS, ITERS, d, d + 1); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemsetAsync(slots, 0, (size_t)NS * 32, cs);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_selfcheck_tear<true>, dim3(BLOCKS), dim3(256), 0, cs, slots, NS, ITERS / 4, d + 2, d + 3); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, cs);
  if (e == hipSuccess) e = hipStreamSynchronize(cs);
  if (cs) (void)hipStreamDestroy(cs);
  (void)hipFree(slots); (void)hipFree(d);
  if (e != hipSuccess) return fail_hip(nullptr, e, "bmx_selfcheck");
  if (reads_out) *reads_out = h[1];
  if (torn_out) *torn_out = h[0];
  if (control_torn_out) *control_torn_out = h[2];
  if (h[0]) {
    char buf[240];
    snprintf(buf, sizeof(buf), "self-check failed on device %d: %llu of %llu aligned 16-byte loads saw HALF of an aligned 16-byte store; the merge kernel's exactness needs them indivisible (DESIGN.md section 4)",
             device, h[0], h[1]);
    return fail(nullptr, BMX_ERR_INTERNAL, buf);
  }
  return BMX_OK;
}

int bmx_create(int device, uint64_t capacity_rows, uint32_t flags, bmx_ctx** out) {
  return bmx_create_ex(device, capacity_rows, 0, flags, out);
}

int bmx_create_ex(int device, uint64_t capacity_rows, uint32_t max_load_pct, uint32_t flags, bmx_ctx** out) {
  if (!out || capacity_rows == 0) return fail(nullptr, BMX_ERR_INVALID, "bmx_create: bad arguments");
  *out = nullptr;
  // argument checks come before the device is looked at: they hold on any machine
  if (max_load_pct == 0) max_load_pct = BMX_DEFAULT_LOAD_PCT;
  if (max_load_pct < 5 || max_load_pct > 90) return fail(nullptr, BMX_ERR_INVALID, "bmx_create_ex: max_load_pct must be 5..90 (0 = default)");
  if (!slots_for(capacity_rows, max_load_pct))
    return fail(nullptr, BMX_ERR_INVALID, "bmx_create: table would need more than 2^32 slots (slot indices are 32-bit): shard the graph over more contexts");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(nullptr, BMX_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(nullptr, BMX_ERR_INVALID, "bmx_create: device index out of range");
  {  // Once per device and process, BEFORE any stream of this library exists: one empty launch on the device's NULL stream. Measured (profiles/r05_defer_timeline_*.txt,
     // r05_step_regression.log): the runtime gives the first stream that is USED in a process its first hardware queue, and a context stream that lands there does not
     // yield wave slots to the high-priority side stream — the deferred compaction then waits ~50 us per step for slots (k_seq_signal, one wave, 54 us; k_resolve_lists
     // 54 instead of 5 us; 141-147 us per step instead of 77-84). Rounds 1-4 used the null stream by accident (the self-check ran there), which is why the deferral
     // worked; with BMX_SKIP_SELFCHECK=1 it never did (round-4 tree, same box: 76.8 -> 140.9 us). The null stream takes that first queue and keeps it.
    static std::atomic<unsigned> touched[64];
    const char* no = std::getenv("BMX_NO_NULL_STREAM_TOUCH");
    if (device < 64 && !(no && no[0] == '1') && !touched[device].load()) {
      if (hipSetDevice(device) == hipSuccess) {
        hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, nullptr);
        (void)hipStreamSynchronize(nullptr);
      }
      (void)hipGetLastError();
      touched[device].store(1u);
    }
  }
  {  // once per device and process: the 16-byte load/store indivisibility the probe kernel relies on is checked on THIS box (BMX_SKIP_SELFCHECK=1 skips it)
    static std::atomic<unsigned> checked[64];
    const char* skip = std::getenv("BMX_SKIP_SELFCHECK");
    if (device < 64 && !(skip && skip[0] == '1') && !checked[device].load()) {
      int src = bmx_selfcheck(device, nullptr, nullptr, nullptr);
      if (src) return src;
      checked[device].store(1u);
    }
  }
  bmx_ctx* ctx = new (std::nothrow) bmx_ctx();
  if (!ctx) return fail(nullptr, BMX_ERR_NOMEM, "out of host memory");
  ctx->device = device;
  ctx->capacity_rows = capacity_rows;
  ctx->load_pct = max_load_pct;
  auto bail = [&](int rc) { std::string m = ctx->err; bmx_destroy(ctx); g_err = m; return rc; };
#define CR(call) do { hipError_t e2 = (call); if (e2 != hipSuccess) { fail_hip(ctx, e2, #call); return bail(BMX_ERR_HIP); } } while (0)
  CR(hipSetDevice(device));
  CR(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;
  CR(hipEventCreate(&ctx->ev0));
  CR(hipEventCreate(&ctx->ev1));
  for (int i = 0; i < 2; i++) { CR(hipEventCreateWithFlags(&ctx->stg[i].up, hipEventDisableTiming)); CR(hipEventCreateWithFlags(&ctx->stg[i].done, hipEventDisableTiming)); }
  const uint64_t nslots = slots_for(capacity_rows, max_load_pct);
  ctx->nslots = nslots;
  ctx->placement_tries_asked = (flags >> 8) & 0xFu;         // BMX_CTX_PLACEMENT_TRIES(n): 0 = the default
  if (ctx->placement_tries_asked > 8) ctx->placement_tries_asked = 8;
  int rc;
  if ((rc = alloc_table_tuned(ctx, nslots, &ctx->slots))) return bail(rc);
  if ((rc = dev_alloc(ctx, &ctx->ds, 1))) return bail(rc);
  for (int i = 0; i < 2; i++) {
    if ((rc = dev_alloc(ctx, &ctx->stg[i].n_out, 1)) || (rc = dev_alloc(ctx, &ctx->stg[i].stats, 1))) return bail(rc);
    CR(hipMemsetAsync(ctx->stg[i].n_out, 0, sizeof(unsigned long long), ctx->stream));
    CR(hipMemsetAsync(ctx->stg[i].stats, 0, sizeof(bmx_merge_stats), ctx->stream));
  }
  if ((rc = dev_alloc(ctx, &ctx->block_counts, SEL_MAX_BLOCKS))) return bail(rc);
  if ((rc = dev_alloc(ctx, &ctx->part_counts, PART_MAX_SHARDS * PART_BLOCKS))) return bail(rc);
  if ((rc = dev_alloc(ctx, &ctx->shard_ctr, bmx_ctx::WS_SETS * CTR_SHARDS * CTR_STRIDE))) return bail(rc);
  CR(hipMemsetAsync(ctx->shard_ctr, 0, bmx_ctx::WS_SETS * CTR_SHARDS * CTR_STRIDE * sizeof(unsigned long long), ctx->stream));
  // the row-count mirror is an optimisation: without mapped host memory the capacity guard simply synchronises as before
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->host_rows), 2 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
    ctx->host_rows[0] = 0; ctx->host_rows[1] = 0;
  } else { ctx->host_rows = nullptr; (void)hipGetLastError(); }
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->stg_tails), 2 * sizeof(SmallOut), hipHostMallocMapped) == hipSuccess) {
    std::memset(ctx->stg_tails, 0, 2 * sizeof(SmallOut));
    for (int i = 0; i < 2; i++) ctx->stg[i].tail = ctx->stg_tails + i;
  } else { ctx->stg_tails = nullptr; (void)hipGetLastError(); }
  ctx->fixed_capacity = (flags & BMX_CTX_FIXED_CAPACITY) != 0;
  ctx->defer_enabled = !launches_are_serialized();
  { const char* vp = std::getenv("BMX_VIEW_PATCH"); if (vp && vp[0] == '0' && !vp[1]) ctx->view_patching = false; }
  { const char* vf = std::getenv("BMX_TEST_VIEW_FAIL"); if (vf && (vf[0] == '1' || vf[0] == '2') && !vf[1]) ctx->view_test_fail = vf[0] - '0'; }
  { const char* vs = std::getenv("BMX_VIEW_SORT"); if (vs && std::strcmp(vs, "own") == 0) ctx->view_own_sort = true; }
  { const char* vp = std::getenv("BMX_VIEW_PENDING"); if (vp && vp[0] == '0' && !vp[1]) ctx->view_pending = false; }
  { const char* kw = std::getenv("BMX_K1_WAVES"); if (kw && kw[0] >= '3' && kw[0] <= '8' && kw[0] != '7' && !kw[1]) ctx->k1_waves = kw[0] - '0'; }
  CR(hipMemsetAsync(ctx->ds, 0, sizeof(DevScalars), ctx->stream));
  hipLaunchKernelGGL(k_init_slots, dim3(2048), dim3(256), 0, ctx->stream, ctx->slots, nslots);
  CR(hipGetLastError());
  CR(hipStreamSynchronize(ctx->stream));
#undef CR
  *out = ctx;
  return BMX_OK;
}

void bmx_destroy(bmx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)flush_pending(ctx); // a compaction that was only recorded writes the CALLER's winner list and count: it runs before anything is freed
  (void)hipGetLastError();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); if (!ctx->side_is_callers) (void)hipStreamDestroy(ctx->side); ctx->side = nullptr; }
  for (auto& ix : ctx->indexes) { dev_free(ix.ids); dev_free(ix.v64); dev_free(ix.v32); free_ordered_view(ix); }
  dev_free(ctx->slots); dev_free(ctx->ds); dev_free(ctx->next); dev_free(ctx->blk_info); dev_free(ctx->blk_follow); dev_free(ctx->shard_ctr);
  for (uint32_t h = 0; h < bmx_ctx::WS_SETS; h++) { dev_free(ctx->wflag[h]); dev_free(ctx->slot_of[h]); dev_free(ctx->fld_ws[h]); }
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (int i = 0; i < 2; i++) {
    bmx_ctx::Staging& S = ctx->stg[i];
    dev_free(S.id); dev_free(S.field); dev_free(S.ts); dev_free(S.val); dev_free(S.applied); dev_free(S.flags); dev_free(S.n_out); dev_free(S.stats);
    if (S.up) (void)hipEventDestroy(S.up);
    if (S.done) (void)hipEventDestroy(S.done);
  }
  dev_free(ctx->pr_id); dev_free(ctx->pr_field); dev_free(ctx->pr_ts); dev_free(ctx->pr_val); dev_free(ctx->pr_found);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->down_stream) { (void)hipStreamSynchronize(ctx->down_stream); (void)hipStreamDestroy(ctx->down_stream); }
  dev_free(ctx->slot_pos); dev_free(ctx->chg);
  for (int i = 0; i < 2; i++) { if (ctx->vk_v[i]) (void)hipFree(ctx->vk_v[i]); dev_free(ctx->vk_p[i]); }
  if (ctx->vk_sv) (void)hipFree(ctx->vk_sv); dev_free(ctx->vk_sp); dev_free(ctx->vk_d0); dev_free(ctx->vk_y0);
  if (ctx->hres) (void)hipHostFree(const_cast<unsigned long long*>(ctx->hres));
  if (ctx->view_err_host) (void)hipHostFree(ctx->view_err_host); if (ctx->view_ev) (void)hipEventDestroy(ctx->view_ev);
  if (ctx->host_rows) { (void)hipHostFree(ctx->host_rows); ctx->host_rows = nullptr; }
  if (ctx->stg_tails) { (void)hipHostFree(ctx->stg_tails); ctx->stg_tails = nullptr; ctx->stg[0].tail = ctx->stg[1].tail = nullptr; }
  if (ctx->pin_in) { (void)hipHostFree(ctx->pin_in); ctx->pin_in = nullptr; }
  if (ctx->pin_out) { (void)hipHostFree(ctx->pin_out); ctx->pin_out = nullptr; }
  dev_free(ctx->scan_out); dev_free(ctx->block_counts); dev_free(ctx->part_counts); dev_free(ctx->part_owner); dev_free(ctx->scan_mask); dev_free(ctx->scan_counts);
  for (auto ev : ctx->prof_ev) (void)hipEventDestroy(ev);
  for (auto ev : ctx->scan_ev) (void)hipEventDestroy(ev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int bmx_sync(bmx_ctx* ctx) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  return check_status(ctx);
}

int bmx_set_stream(bmx_ctx* ctx, void* s) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;        // a recorded compaction belongs on the old stream
  HIPCHK(hipStreamSynchronize(ctx->stream));   // work already enqueued on the old stream finishes first
  ctx->stream = s ? reinterpret_cast<hipStream_t>(s) : ctx->own_stream;
  return BMX_OK;
}
void* bmx_get_stream(bmx_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

int bmx_seq_signal(bmx_ctx* ctx, void* hip_stream, uint64_t* seq_dev, uint64_t value) {
  if (!ctx || !seq_dev) return fail(ctx, BMX_ERR_INVALID, "bmx_seq_signal: null context or sequence word");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream;
  // a signal on the context's own stream says "every merge enqueued before this is done, outputs included": a compaction that is only recorded
  // (or still on the side stream) is ordered in front of it (a consumer stream woken by the word read a stale n_applied)
  if (st == ctx->stream) { if (int frc = flush_pending(ctx)) return frc; }
  hipLaunchKernelGGL(k_seq_signal, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long*>(seq_dev), (unsigned long long)value);
  LAUNCHCHK("k_seq_signal");
  return BMX_OK;
}

int bmx_seq_wait(bmx_ctx* ctx, void* hip_stream, const uint64_t* seq_dev, uint64_t at_least) {
  if (!ctx || !seq_dev) return fail(ctx, BMX_ERR_INVALID, "bmx_seq_wait: null context or sequence word");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream;
  hipLaunchKernelGGL(k_seq_wait, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned long long*>(seq_dev), (unsigned long long)at_least,
                     &ctx->ds->status, ctx->ds->seq_diag);
  LAUNCHCHK("k_seq_wait");
  return BMX_OK;
}

int bmx_get_info(bmx_ctx* ctx, bmx_info* out) {
  if (!ctx || !out) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  int rc = refresh_rows(ctx);
  if (rc) return rc;
  out->capacity_rows = ctx->capacity_rows; out->n_slots = ctx->nslots; out->table_bytes = ctx->nslots * sizeof(Slot); out->n_rows = ctx->rows_ub;
  out->device = (uint32_t)ctx->device; out->abi_version = BMX_ABI_VERSION; out->n_indexes = (uint32_t)ctx->indexes.size(); out->epoch = ctx->epoch;
  return BMX_OK;
}

int bmx_row_count(bmx_ctx* ctx, uint64_t* n_out) {
  if (!ctx || !n_out) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  int rc = refresh_rows(ctx);
  if (rc) return rc;
  *n_out = ctx->rows_ub;
  return BMX_OK;
}

int bmx_reserve(bmx_ctx* ctx, uint64_t capacity_rows) {
  if (!ctx || capacity_rows == 0) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  return grow_table(ctx, capacity_rows);
}

int bmx_merge_batch(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                    int insert_mode, int mem, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (n && (!id || !field || !ts || !val)) return fail(ctx, BMX_ERR_INVALID, "null input column");
  if (!public_mode_ok(insert_mode)) return fail(ctx, BMX_ERR_INVALID, "bad insert_mode");
  HIPCHK(hipSetDevice(ctx->device));
  if (mem == BMX_MEM_DEVICE) return merge_core<false>(ctx, n, id, field, ts, val, nullptr, insert_mode, applied_idx, n_applied, flags, stats, /*defer=*/true);
  if (mem != BMX_MEM_HOST) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (n > MAX_BATCH) return fail(ctx, BMX_ERR_INVALID, "batch larger than 2^24 deltas: split it (sequential semantics are preserved)");
  return merge_host(ctx, n, id, field, ts, val, insert_mode, applied_idx, n_applied, flags, stats);
}

int bmx_merge_submit(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int insert_mode,
                     int want_flags, uint64_t* ticket) {
  if (!ctx || !ticket) return fail(ctx, BMX_ERR_INVALID, "bmx_merge_submit: null context or ticket");
  if (n && (!id || !field || !ts || !val)) return fail(ctx, BMX_ERR_INVALID, "null input column");
  if (n > MAX_BATCH) return fail(ctx, BMX_ERR_INVALID, "batch larger than 2^24 deltas: split it (sequential semantics are preserved)");
  if (!public_mode_ok(insert_mode)) return fail(ctx, BMX_ERR_INVALID, "bad insert_mode");
  if (int erc = enter(ctx)) return erc;
  return submit_host(ctx, n, id, field, ts, val, insert_mode, want_flags != 0, ticket, true);
}

int bmx_merge_collect(bmx_ctx* ctx, uint64_t ticket, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  return collect_host(ctx, ticket, applied_idx, n_applied, flags, stats);
}

int bmx_merge_records(bmx_ctx* ctx, uint64_t n, const bmx_delta_rec* recs, int insert_mode, uint32_t* applied_idx, uint64_t* n_applied,
                      uint8_t* flags, bmx_merge_stats* stats) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (n && !recs) return fail(ctx, BMX_ERR_INVALID, "null records");
  if (!public_mode_ok(insert_mode)) return fail(ctx, BMX_ERR_INVALID, "bad insert_mode");
  HIPCHK(hipSetDevice(ctx->device));
  return merge_core<true>(ctx, n, nullptr, nullptr, nullptr, nullptr, recs, insert_mode, applied_idx, n_applied, flags, stats, /*defer=*/true);
}

int bmx_merge_records_after(bmx_ctx* ctx, const uint64_t* wait_words_dev, uint32_t n_wait, uint64_t wait_at_least, uint64_t n, const bmx_delta_rec* recs,
                            int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (wait_words_dev && n_wait) {
    const bool done = ctx->tail_waited.n == n_wait && ctx->tail_waited.words == reinterpret_cast<const unsigned long long*>(wait_words_dev) && ctx->tail_waited.at_least >= wait_at_least;
    ctx->tail_waited = bmx_ctx::TailWait{};
    if (!done) { int wrc = bmx_seq_wait_all(ctx, nullptr, wait_words_dev, n_wait, wait_at_least); if (wrc) return wrc; }   // (done: the resolve kernel of the merge before waited for exactly this)
  }
  ctx->notify_armed = true;      // THIS merge reads a receive slab set: it (and no other merge of the context) tells the origins when the set is free again
  const int rc = bmx_merge_records(ctx, n, recs, insert_mode, applied_idx, n_applied, flags, stats);
  ctx->notify_armed = false;
  return rc;
}

int bmx_load_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (n && (!id || !field || !ts || !val)) return fail(ctx, BMX_ERR_INVALID, "null input column");
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (int erc = enter(ctx)) return erc;
  const uint64_t chunk = 1u << 22;
  for (uint64_t off = 0; off < n; off += chunk) {
    uint64_t m = std::min<uint64_t>(chunk, n - off);
    int rc;
    if (mem == BMX_MEM_DEVICE)
      rc = merge_core<false>(ctx, m, id + off, field + off, ts + off, val + off, nullptr, BMX_INSERT_DELTA, nullptr, nullptr, nullptr, nullptr);
    else
      rc = merge_host(ctx, m, id + off, field + off, ts + off, val + off, BMX_INSERT_DELTA, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  return mem == BMX_MEM_DEVICE ? BMX_OK : check_status(ctx);
}

int bmx_put_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (n && (!id || !field || !ts || !val)) return fail(ctx, BMX_ERR_INVALID, "null input column");
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (int erc = enter(ctx)) return erc;
  const uint64_t chunk = 1u << 22;
  for (uint64_t off = 0; off < n; off += chunk) {
    const uint64_t m = std::min<uint64_t>(chunk, n - off);
    int rc;
    if (mem == BMX_MEM_DEVICE)
      rc = merge_core<false>(ctx, m, id + off, field + off, ts + off, val + off, nullptr, BMX_INSERT_DELTA, nullptr, nullptr, nullptr, nullptr, false, /*force=*/true);
    else
      rc = merge_host(ctx, m, id + off, field + off, ts + off, val + off, MERGE_FORCE_INTERNAL, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  return mem == BMX_MEM_DEVICE ? BMX_OK : check_status(ctx);
}

int bmx_get_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, int64_t* ts, int64_t* val, uint8_t* found, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (n == 0) return BMX_OK;
  if (!id || !field || !ts || !val || !found || n > 0xFFFFFFFFull) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  if (mem == BMX_MEM_DEVICE) {
    hipLaunchKernelGGL(k_get_rows, dim3(blocks), dim3(256), 0, ctx->stream, ctx->slots, ctx->nslots, (uint32_t)n, id, field, ts, val, found);
    LAUNCHCHK("k_get_rows");
    return BMX_OK;
  }
  if (mem != BMX_MEM_HOST) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (n <= 8192 && ensure_pinned(ctx)) {   // few keys: keys and answers through mapped host memory, one launch and one synchronisation (25 us instead of 67)
    uint64_t* p_id = reinterpret_cast<uint64_t*>(ctx->pin_in);
    uint32_t* p_f = reinterpret_cast<uint32_t*>(ctx->pin_in + n * 8);
    int64_t* o_ts = reinterpret_cast<int64_t*>(ctx->pin_out);
    int64_t* o_val = reinterpret_cast<int64_t*>(ctx->pin_out + n * 8);
    uint8_t* o_found = ctx->pin_out + n * 16;
    std::memcpy(p_id, id, n * 8); std::memcpy(p_f, field, n * 4);
    hipLaunchKernelGGL(k_get_rows, dim3(blocks), dim3(256), 0, ctx->stream, ctx->slots, ctx->nslots, (uint32_t)n, (const uint64_t*)p_id, (const uint32_t*)p_f, o_ts, o_val, o_found);
    LAUNCHCHK("k_get_rows");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::memcpy(ts, o_ts, n * 8); std::memcpy(val, o_val, n * 8); std::memcpy(found, o_found, n);
    return BMX_OK;
  }
  int rc = ensure_point_read(ctx, n);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ctx->pr_id, id, n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->pr_field, field, n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_get_rows, dim3(blocks), dim3(256), 0, ctx->stream, ctx->slots, ctx->nslots, (uint32_t)n, ctx->pr_id, ctx->pr_field, ctx->pr_ts, ctx->pr_val, ctx->pr_found);
  LAUNCHCHK("k_get_rows");
  HIPCHK(hipMemcpyAsync(ts, ctx->pr_ts, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(val, ctx->pr_val, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(found, ctx->pr_found, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return BMX_OK;
}

int bmx_get_row(bmx_ctx* ctx, uint64_t id, uint32_t field, int64_t* ts, int64_t* val) {
  uint8_t found = 0;
  int64_t t = 0, v = 0;
  int rc = bmx_get_rows(ctx, 1, &id, &field, &t, &v, &found, BMX_MEM_HOST);
  if (rc) return rc;
  if (found) { if (ts) *ts = t; if (val) *val = v; }
  return found ? 1 : 0;
}

int bmx_dump_rows(bmx_ctx* ctx, uint64_t cap, uint64_t* id, uint32_t* field, int64_t* ts, int64_t* val, uint64_t* n_out, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (cap && (!id || !field || !ts || !val)) return fail(ctx, BMX_ERR_INVALID, "null output column");
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (int erc = enter(ctx)) return erc;
  const bool host = mem == BMX_MEM_HOST;
  uint64_t* d_id = id; uint32_t* d_f = field; int64_t* d_ts = ts; int64_t* d_val = val;
  int rc = BMX_OK;
  if (host && cap) {   // persistent (grow-only) device columns: no allocation per call
    if ((rc = ensure_point_read(ctx, cap))) return rc;
    d_id = ctx->pr_id; d_f = ctx->pr_field; d_ts = ctx->pr_ts; d_val = ctx->pr_val;
  }
  PredSlotAny P{ctx->slots};
  EmitRows Em{ctx->slots, cap, d_id, d_f, d_ts, d_val};
  FinishCount Fin{host ? &ctx->ds->n_out : reinterpret_cast<unsigned long long*>(n_out)};
  SelGeom g = sel_geom<PredSlotAny::E>(ctx->nslots);
  hipLaunchKernelGGL((k_sel_count<PredSlotAny>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, ctx->nslots, g.tiles_per_block, ctx->block_counts);
  hipLaunchKernelGGL((k_sel_write<PredSlotAny, EmitRows, FinishCount>), dim3(g.blocks), dim3(SEL_THREADS), 0, ctx->stream, P, Em, Fin, ctx->nslots,
                     g.tiles_per_block, ctx->block_counts);
  hipError_t e = hipGetLastError();
  if (host) {
    unsigned long long m = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&m, &ctx->ds->n_out, sizeof(m), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    uint64_t k = std::min<uint64_t>(m, cap);
    if (e == hipSuccess && k) {
      e = hipMemcpy(id, d_id, k * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(field, d_f, k * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(ts, d_ts, k * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(val, d_val, k * 8, hipMemcpyDeviceToHost);
    }
    if (n_out) *n_out = m;
  }
  if (e != hipSuccess) return fail_hip(ctx, e, "bmx_dump_rows");
  return BMX_OK;
}

int bmx_index_build(bmx_ctx* ctx, uint32_t field) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  Index* ix;
  return fresh_index(ctx, field, &ix);
}

int bmx_index_drop(bmx_ctx* ctx, uint32_t field) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < ctx->indexes.size(); i++)
    if (ctx->indexes[i].field == field) {
      dev_free(ctx->indexes[i].ids); dev_free(ctx->indexes[i].v64); dev_free(ctx->indexes[i].v32); free_ordered_view(ctx->indexes[i]);
      ctx->indexes.erase(ctx->indexes.begin() + (long)i);
      if (ctx->indexes.empty()) {   // nothing left to maintain: the merges stop logging and the maintenance memory goes back
        ctx->chg_valid = false; ctx->chg_ub = 0;
        dev_free(ctx->chg); ctx->chg_cap = 0;
        dev_free(ctx->slot_pos); ctx->slot_pos_n = 0;
      }
      return BMX_OK;
    }
  return fail(ctx, BMX_ERR_NO_INDEX, "no index on that field");
}

int bmx_index_size(bmx_ctx* ctx, uint32_t field, uint64_t* n_out) {
  if (!ctx || !n_out) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  Index* ix;
  int rc = fresh_index(ctx, field, &ix);
  if (rc) return rc;
  *n_out = ix->n;
  return BMX_OK;
}

int bmx_index_refresh_counts(bmx_ctx* ctx, uint64_t* full_builds, uint64_t* incremental_updates) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (full_builds) *full_builds = ctx->ix_full_builds;
  if (incremental_updates) *incremental_updates = ctx->ix_incremental;
  return BMX_OK;
}

int bmx_index_set_ordered(bmx_ctx* ctx, uint32_t field, uint32_t after_queries) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  Index* ix;
  int rc = fresh_index(ctx, field, &ix);      // (creates the index like a first query would)
  if (rc) return rc;
  ix->ordered_after = after_queries;
  ix->stale_queries = 0;
  if (!after_queries) { HIPCHK(hipStreamSynchronize(ctx->stream)); free_ordered_view(*ix); }
  return BMX_OK;
}
int bmx_index_ordered_info(bmx_ctx* ctx, uint32_t field, uint32_t* after_queries, int* valid_now, uint64_t* sorts) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  Index* ix = find_index(ctx, field);
  if (!ix) return fail(ctx, BMX_ERR_INVALID, "bmx_index_ordered_info: no index on this field");
  if (after_queries) *after_queries = ix->ordered_after;
  if (valid_now) *valid_now = ix->ordered_after && ix->s_val && ix->ord_content == ix->content && ix->version == ctx->version;
  if (sorts) *sorts = ix->ord_sorts;
  return BMX_OK;
}

int bmx_index_ordered_stats(bmx_ctx* ctx, uint32_t field, uint64_t* sorts, uint64_t* patches, uint64_t* keys_patched, double* last_sort_us, double* last_patch_us, uint64_t* rewrites, uint64_t* pending_keys) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  Index* ix = find_index(ctx, field);
  if (!ix) return fail(ctx, BMX_ERR_INVALID, "bmx_index_ordered_stats: no index on this field");
  if (sorts) *sorts = ix->ord_sorts;
  if (patches) *patches = ix->ord_patches;
  if (keys_patched) *keys_patched = ix->ord_patched_keys;
  if (last_sort_us) *last_sort_us = ix->last_sort_us;
  if (last_patch_us) *last_patch_us = ix->last_patch_us;
  if (rewrites) *rewrites = ix->ord_merges;
  if (pending_keys) *pending_keys = ix->npd + ix->npi;
  return BMX_OK;
}

int bmx_scan_range(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap, uint64_t* n_out, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  return scan_range_impl(ctx, field, lo, hi, out_ids, cap, n_out, mem);
}
int bmx_scan_equals(bmx_ctx* ctx, uint32_t field, int64_t value, uint64_t* out_ids, uint64_t cap, uint64_t* n_out, int mem) {
  return bmx_scan_range(ctx, field, value, value, out_ids, cap, n_out, mem);
}
int bmx_scan_count(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint64_t* n_out, int mem) {
  return bmx_scan_range(ctx, field, lo, hi, nullptr, 0, n_out, mem);
}

int bmx_scan_range_pos(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint32_t* out_pos, uint64_t cap, uint64_t* n_out, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  return scan_range_impl_t<true>(ctx, field, lo, hi, out_pos, cap, n_out, mem);
}

int bmx_index_ids(bmx_ctx* ctx, uint32_t field, uint64_t first, uint64_t count, uint64_t* out_ids, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (int erc = enter(ctx)) return erc;
  Index* ix;
  int rc = fresh_index(ctx, field, &ix);
  if (rc) return rc;
  if (first > ix->n || count > ix->n - first) return fail(ctx, BMX_ERR_INVALID, "bmx_index_ids: range beyond the index (bmx_index_size)");
  if (count == 0) return BMX_OK;
  if (!out_ids) return fail(ctx, BMX_ERR_INVALID, "null output");
  HIPCHK(hipMemcpyAsync(out_ids, ix->ids + first, count * sizeof(uint64_t), host_or_dev(mem), ctx->stream));
  if (mem == BMX_MEM_HOST) HIPCHK(hipStreamSynchronize(ctx->stream));
  return BMX_OK;
}

int bmx_scan_filter(bmx_ctx* ctx, uint32_t nterms, const bmx_term* terms, uint64_t* out_ids, uint64_t cap, uint64_t* n_out, int mem) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (nterms == 0 || nterms > MAX_TERMS || !terms) return fail(ctx, BMX_ERR_INVALID, "filter needs 1..8 terms");
  if (mem != BMX_MEM_HOST && mem != BMX_MEM_DEVICE) return fail(ctx, BMX_ERR_INVALID, "bad mem kind");
  if (int erc = enter(ctx)) return erc;
  Index* ix;
  int rc = fresh_index(ctx, terms[0].field, &ix);
  if (rc) return rc;
  PredFilter P;
  P.v = ix->v64; P.ids = ix->ids; P.slots = ctx->slots; P.nslots = ctx->nslots; P.nterms = nterms;
  for (uint32_t k = 0; k < nterms; k++) { P.t[k] = terms[k]; P.t[k].lo = std::max<int64_t>(terms[k].lo, -VAL_MAX); }   // tombstones (INT64_MIN) match no term
  // with a value-ordered view of the first term's index: its run is the candidate list, the other terms are probed for those ids only (no order)
  if ((n_out || out_ids) && ensure_ordered_view(ctx, ix)) {
    const int src = run_scan_t<false>(ctx, P, ix, out_ids, cap, n_out, mem, true, P.t[0].lo, P.t[0].hi);
    if (!src) view_after_query(ctx, ix);
    return src;
  }
  return run_scan(ctx, P, ix, out_ids, cap, n_out, mem);
}

static int partition_impl(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                          uint32_t nshards, uint64_t slab, bmx_delta_rec* recs_out, uint64_t* counts_out_dev, const PartOut* split = nullptr, uint32_t aux_base = 0) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (nshards == 0 || nshards > PART_MAX_SHARDS || n > 0xFFFFFFFFull || !counts_out_dev || slab * nshards > 0xFFFFFFFFull)
    return fail(ctx, BMX_ERR_INVALID, "bad arguments (1..16 shards)");
  if (n && (!id || !field || !ts || !val || (!recs_out && !split))) return fail(ctx, BMX_ERR_INVALID, "null pointer");
  PartOut po;
  if (split) po = *split; else std::memset(&po, 0, sizeof(po));
  po.split = split ? 1u : 0u; po.aux_base = aux_base;
  if (int erc = enter(ctx)) return erc;
  uint32_t per_block = (uint32_t)((n + PART_BLOCKS - 1) / PART_BLOCKS);
  per_block = std::max<uint32_t>(PART_TILE, (per_block + PART_TILE - 1) / PART_TILE * PART_TILE);
  if (n > ctx->part_owner_cap) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->part_owner); ctx->part_owner_cap = 0;
    int rc = dev_alloc(ctx, &ctx->part_owner, n + 256);
    if (rc) return rc;
    ctx->part_owner_cap = n;
  }
  hipLaunchKernelGGL(k_part_count, dim3(PART_BLOCKS), dim3(256), 0, ctx->stream, id, (uint32_t)n, nshards, per_block, ctx->part_counts, ctx->part_owner);
  LAUNCHCHK("k_part_count");
  hipLaunchKernelGGL(k_part_scatter, dim3(PART_BLOCKS), dim3(256), 0, ctx->stream, id, field, ts, val, (const uint8_t*)ctx->part_owner, (uint32_t)n, nshards, per_block, ctx->part_counts,
                     recs_out, reinterpret_cast<unsigned long long*>(counts_out_dev), (uint32_t)slab, &ctx->ds->status, po);
  LAUNCHCHK("k_part_scatter");
  return BMX_OK;
}

int bmx_partition_by_owner(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                           uint32_t nshards, bmx_delta_rec* recs_out, uint64_t* counts_out_dev) {
  return partition_impl(ctx, n, id, field, ts, val, nshards, 0, recs_out, counts_out_dev);
}

int bmx_partition_by_owner_slabs(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                                 uint32_t nshards, uint64_t slab_records, bmx_delta_rec* recs_out, uint64_t* counts_out_dev) {
  if (slab_records == 0) return fail(ctx, BMX_ERR_INVALID, "slab_records must be > 0");
  return partition_impl(ctx, n, id, field, ts, val, nshards, slab_records, recs_out, counts_out_dev);
}

/* ---- direct exchange between processes (one process per GPU): IPC-mapped receive slabs, arrival words, no collective on the data path ---- */
int bmx_ipc_alloc(bmx_ctx* ctx, uint64_t bytes, uint32_t flags, void** dev_ptr, uint8_t handle_out[64]) {
  if (!ctx || !dev_ptr || !handle_out || bytes == 0) return fail(ctx, BMX_ERR_INVALID, "bmx_ipc_alloc: bad arguments");
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle fits the 64-byte carrier");
  if (int erc = enter(ctx)) return erc;
  void* p = nullptr;
  // BMX_IPC_UNCACHED: memory other GPUs store into while kernels here poll or read it must not be served from this GPU's L2 (a line cached
  // before the peer's store would stay stale: the L2 is only coherent for this GPU's own writes)
  hipError_t e = (flags & BMX_IPC_UNCACHED) ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) : hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? BMX_ERR_NOMEM : BMX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  e = hipMemset(p, 0, bytes);
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return fail_hip(ctx, e, "hipIpcGetMemHandle"); }
  std::memset(handle_out, 0, 64);
  std::memcpy(handle_out, &h, sizeof(h));
  *dev_ptr = p;
  return BMX_OK;
}
int bmx_ipc_open(bmx_ctx* ctx, const uint8_t handle[64], int peer_device, void** dev_ptr) {
  if (!ctx || !handle || !dev_ptr) return fail(ctx, BMX_ERR_INVALID, "bmx_ipc_open: bad arguments");
  if (int erc = enter(ctx)) return erc;
  if (peer_device >= 0 && peer_device != ctx->device) {       // peer access first: the mapping below is only usable from this GPU with it
    int can = 0;
    HIPCHK(hipDeviceCanAccessPeer(&can, ctx->device, peer_device));
    if (!can) return fail(ctx, BMX_ERR_HIP, "bmx_ipc_open: no peer access between the two GPUs");
    hipError_t pe = hipDeviceEnablePeerAccess(peer_device, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail_hip(ctx, pe, "hipDeviceEnablePeerAccess");
    (void)hipGetLastError();
  }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *dev_ptr = p;
  return BMX_OK;
}
int bmx_ipc_close(bmx_ctx* ctx, void* dev_ptr) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (!dev_ptr) return BMX_OK;
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipIpcCloseMemHandle(dev_ptr));
  return BMX_OK;
}
int bmx_ipc_free(bmx_ctx* ctx, void* dev_ptr) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (!dev_ptr) return BMX_OK;
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipFree(dev_ptr));
  return BMX_OK;
}

int bmx_host_alloc(uint64_t bytes, void** host_ptr) {
  if (!host_ptr || bytes == 0) return fail(nullptr, BMX_ERR_INVALID, "bmx_host_alloc: bad arguments");
  *host_ptr = nullptr;
  hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); *host_ptr = nullptr; return fail(nullptr, e == hipErrorOutOfMemory ? BMX_ERR_NOMEM : BMX_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return BMX_OK;
}
int bmx_host_free(void* host_ptr) {
  if (!host_ptr) return BMX_OK;
  hipError_t e = hipHostFree(host_ptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, BMX_ERR_HIP, std::string("hipHostFree: ") + hipGetErrorString(e)); }
  return BMX_OK;
}

int bmx_partition_scatter(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                          uint32_t nshards, uint64_t slab_records, void* const* dst, uint64_t* counts_out_dev, uint64_t* const* arrive_words, uint64_t arrive_value,
                          const uint64_t* wait_words_dev, uint32_t n_wait, uint64_t wait_at_least) {
  if (!ctx || !dst || slab_records == 0 || nshards == 0 || nshards > PART_MAX_SHARDS) return fail(ctx, BMX_ERR_INVALID, "bmx_partition_scatter: bad arguments (1..16 shards, slab_records > 0)");
  // A/B switch, OFF by default: BMX_PART_WAIT_FOLD=1 folds the wait into the scatter pass (every workgroup polls before its first store) instead of a one-wave launch in
  // front of the count pass. Measured (profiles/r05_sharded_ab.log): 95.0 / 96.9 against 102.1 / 96.4 us per step — inside the run-to-run spread — and one run with the
  // deferred compaction beside it took 75 ms per step: 1024 spinning workgroups hold the LDS and wave slots the kernel that frees the slabs needs. A one-wave wait cannot do that.
  static const bool fold_wait = [] { const char* v = std::getenv("BMX_PART_WAIT_FOLD"); return v && v[0] == '1' && !v[1]; }();
  if (wait_words_dev && n_wait && (!fold_wait || n_wait > 64)) { int wrc = bmx_seq_wait_all(ctx, nullptr, wait_words_dev, n_wait, wait_at_least); if (wrc) return wrc; }
  PartOut po; std::memset(&po, 0, sizeof(po));
  if (wait_words_dev && n_wait && fold_wait && n_wait <= 64) {
    po.wait_words = reinterpret_cast<const unsigned long long*>(wait_words_dev); po.n_wait = n_wait; po.wait_at_least = wait_at_least; po.wait_diag = ctx->ds->seq_diag;
  }
  for (uint32_t g = 0; g < nshards; g++) {
    if (!dst[g]) return fail(ctx, BMX_ERR_INVALID, "bmx_partition_scatter: null destination slab");
    po.base[g] = static_cast<bmx_delta_rec*>(dst[g]);
  }
  int rc = partition_impl(ctx, n, id, field, ts, val, nshards, slab_records, nullptr, counts_out_dev, &po, 0);
  if (rc || !arrive_words) return rc;
  // the arrival words, from a launch of their own behind the scatter: its kernel boundary is the release (every record is stored and written
  // back, peer memory included) — fences inside the scatter's 1024 workgroups write the whole L2 back a thousand times (measured: +100 us)
  SeqPtrs w; std::memset(&w, 0, sizeof(w));
  for (uint32_t g = 0; g < nshards; g++) w.p[g] = reinterpret_cast<unsigned long long*>(arrive_words[g]);
  hipLaunchKernelGGL(k_seq_signal_multi, dim3(1), dim3(64), 0, ctx->stream, w, nshards, (unsigned long long)arrive_value);
  LAUNCHCHK("k_seq_signal_multi");
  return BMX_OK;
}

int bmx_seq_wait_all(bmx_ctx* ctx, void* hip_stream, const uint64_t* words_dev, uint32_t nwords, uint64_t at_least) {
  if (!ctx || !words_dev || nwords == 0 || nwords > 64) return fail(ctx, BMX_ERR_INVALID, "bmx_seq_wait_all: 1..64 words");
  HIPCHK(hipSetDevice(ctx->device));      // (no flush: a wait in front of the next merge must not pull the recorded compaction onto this stream)
  hipStream_t st = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream;
  hipLaunchKernelGGL(k_seq_wait_all, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned long long*>(words_dev), nwords, (unsigned long long)at_least,
                     &ctx->ds->status, ctx->ds->seq_diag);
  LAUNCHCHK("k_seq_wait_all");
  return BMX_OK;
}

int bmx_merge_tail_wait(bmx_ctx* ctx, const uint64_t* words_dev, uint32_t nwords, uint64_t at_least) {
  if (!ctx || nwords > 64 || (nwords && !words_dev)) return fail(ctx, BMX_ERR_INVALID, "bmx_merge_tail_wait: at most 64 words");
  ctx->tail_armed.words = reinterpret_cast<const unsigned long long*>(words_dev); ctx->tail_armed.n = nwords; ctx->tail_armed.at_least = at_least;
  return BMX_OK;
}

int bmx_merge_notify(bmx_ctx* ctx, uint64_t* const* words, uint32_t nwords) {
  if (!ctx || nwords > PART_MAX_SHARDS || (nwords && !words)) return fail(ctx, BMX_ERR_INVALID, "bmx_merge_notify: at most 16 words");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memset(&ctx->notify, 0, sizeof(ctx->notify));
  for (uint32_t k = 0; k < nwords; k++) ctx->notify.p[k] = reinterpret_cast<unsigned long long*>(words[k]);
  ctx->n_notify = nwords; ctx->notify_seq = 0;
  return BMX_OK;
}

int bmx_set_deferred_compaction(bmx_ctx* ctx, int on) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  ctx->defer_enabled = on != 0 && !launches_are_serialized();    // (never where kernels run one at a time: see launches_are_serialized)
  return BMX_OK;
}
int bmx_set_wait_limit(bmx_ctx* ctx, double seconds) {
  if (!ctx || !(seconds >= 0.001) || seconds > 3600.0) return fail(ctx, BMX_ERR_INVALID, "bmx_set_wait_limit: 0.001 .. 3600 seconds");
  HIPCHK(hipSetDevice(ctx->device));
  const unsigned long long ticks = (unsigned long long)(seconds * 1.0e8);      // wall_clock64(): 100 MHz
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_ticks), &ticks, sizeof(ticks), 0, hipMemcpyHostToDevice));
  return BMX_OK;
}
int bmx_set_side_stream(bmx_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;            // whatever is recorded or still on the old side stream is ordered into the context's stream first
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->side) { HIPCHK(hipStreamSynchronize(ctx->side)); if (!ctx->side_is_callers) (void)hipStreamDestroy(ctx->side); }
  ctx->side = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : nullptr;    // nullptr: the next deferring merge creates the context's own again
  ctx->side_is_callers = hip_stream != nullptr;
  return BMX_OK;
}
int bmx_merge_fence(bmx_ctx* ctx) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  return enter(ctx);
}
int bmx_get_deferred_counts(bmx_ctx* ctx, uint64_t* deferred, uint64_t* on_side_stream) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (deferred) *deferred = ctx->n_deferred;
  if (on_side_stream) *on_side_stream = ctx->n_side;
  return BMX_OK;
}

int bmx_set_probe_waves(bmx_ctx* ctx, int waves_per_simd) {
  if (!ctx || waves_per_simd < 3 || waves_per_simd > 8 || waves_per_simd == 7) return fail(ctx, BMX_ERR_INVALID, "bmx_set_probe_waves: 3, 4, 5, 6 or 8");
  ctx->k1_waves = waves_per_simd;
  return BMX_OK;
}

int bmx_get_placement(bmx_ctx* ctx, uint32_t* candidates, float* probe_us_chosen, float* probe_us_slowest) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (candidates) *candidates = ctx->placement_tries;
  if (probe_us_chosen) *probe_us_chosen = ctx->placement_us_best;
  if (probe_us_slowest) *probe_us_slowest = ctx->placement_us_worst;
  return BMX_OK;
}

int bmx_timer_start(bmx_ctx* ctx) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
  return BMX_OK;
}
int bmx_timer_stop(bmx_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;          // the last batch's compaction is part of what is timed
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(hipEventSynchronize(ctx->ev1));
  HIPCHK(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return BMX_OK;
}

int bmx_timer_mark(bmx_ctx* ctx) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;          // the last batch's compaction is part of what is timed
  HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
  return BMX_OK;
}
int bmx_timer_elapsed(bmx_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventSynchronize(ctx->ev1));
  HIPCHK(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return BMX_OK;
}

int bmx_profile_enable(bmx_ctx* ctx, int on) {
  if (!ctx) return fail(nullptr, BMX_ERR_INVALID, "null context");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (on && ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(4 * PROF_MAX_CALLS, nullptr);
    for (auto& ev : ctx->prof_ev) HIPCHK(hipEventCreate(&ev));
    ctx->scan_ev.resize(3 * PROF_MAX_CALLS, nullptr);
    for (auto& ev : ctx->scan_ev) HIPCHK(hipEventCreate(&ev));
  }
  ctx->prof_on = on != 0;
  ctx->prof_n = 0;
  ctx->scan_prof_n = 0;
  return BMX_OK;
}

int bmx_profile_read(bmx_ctx* ctx, float ms_out[3], uint32_t* n_calls) {
  if (!ctx || !ms_out || !n_calls) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double acc[3] = {0, 0, 0};
  for (uint32_t i = 0; i < ctx->prof_n; i++)
    for (int k = 0; k < 3; k++) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ctx->prof_ev[4 * i + k], ctx->prof_ev[4 * i + k + 1]));
      acc[k] += ms;
    }
  for (int k = 0; k < 3; k++) ms_out[k] = ctx->prof_n ? (float)(acc[k] / ctx->prof_n) : 0.f;
  *n_calls = ctx->prof_n;
  return BMX_OK;
}

int bmx_profile_read_scan(bmx_ctx* ctx, float ms_out[2], uint32_t* n_calls) {
  if (!ctx || !ms_out || !n_calls) return fail(ctx, BMX_ERR_INVALID, "bad arguments");
  if (int erc = enter(ctx)) return erc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double acc[2] = {0, 0};
  for (uint32_t i = 0; i < ctx->scan_prof_n; i++)
    for (int k = 0; k < 2; k++) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ctx->scan_ev[3 * i + k], ctx->scan_ev[3 * i + k + 1]));
      acc[k] += ms;
    }
  for (int k = 0; k < 2; k++) ms_out[k] = ctx->scan_prof_n ? (float)(acc[k] / ctx->scan_prof_n) : 0.f;
  *n_calls = ctx->scan_prof_n;
  return BMX_OK;
}

}  // extern "C"

#include "bmx_vc.inc"
#include "bmx_comm.inc"
