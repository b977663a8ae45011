/*
 * bmx.h — C ABI of libbmx.so: the MI355X (gfx950) CRDT-merge and index-scan engine that sits
 * behind Bullet's `crt` and `query` plug points.
 *
 * This is the drop-in boundary. The reference (KORandi/bullet-js) is pure JavaScript with no FFI;
 * the entry points below are what an N-API addon binds (bullet-js_amd/napi/bmx_napi.cc) so that
 *   bullet.crt   = new GpuCRT(bullet)     replaces src/bullet-crt.js    (called at src/bullet.js:141-142)
 *   bullet.query = new GpuQuery(bullet)   replaces src/bullet-query.js  (called at src/bullet.js:313-389)
 * Each function cites the reference interface it replaces. Plain pointers and sizes only; no C++,
 * torch or HIP types cross this boundary. Little-endian, 64-bit sizes.
 *
 * Data model: one resident row per (node-id hash u64, field hash u32) holding a
 * scalar clock `ts` (int64, 0 <= ts <= 2^53-1: a single-component vector clock {w: ts}) and an integer
 * value `val` (int64, |val| <= 2^53-1: exact in a JS number). Rows whose clocks name several writers live in a second kind of
 * table (bmx_vc_*, "N4" below: up to 8 known writers). Strings, objects and clocks outside those contracts stay on the host
 * (GpuCRT's single-op path).
 *
 * Reserved key values (never produced by the host hash functions): id 0xFFFFFFFFFFFFFFFF, field 0xFFFFFFFF.
 *
 * Threading: a context is bound to one GPU and one HIP stream and is not re-entrant.
 * Errors: every call returns BMX_OK (0) or a negative code; bmx_last_error() gives the text. The
 * library never aborts the process (reference error policy: src/bullet.js:230-234).
 */
#ifndef BMX_H
#define BMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history. 1: rounds 1-3 (during which BMX_MERGE_BUCKETED 0x800 / BMX_CTX_BUCKETED_MERGE were removed and ~20 entry points added without a bump).
 * 2: every insert_mode bit that is not documented below is refused with BMX_ERR_INVALID; deferred compaction (bmx_set_deferred_compaction,
 *    bmx_merge_fence, bmx_get_deferred_counts); bmx_selfcheck. A caller built against 1 that passes only documented bits keeps working.
 * 3: value-ordered index views (bmx_index_set_ordered, bmx_index_ordered_info): additions only; nothing a caller built against 2 uses has changed.
 * 4: a current view stays current under writes (patched from the change log; bmx_index_ordered_stats), BMX_CTX_PLACEMENT_TRIES, bmx_seq_signal on the context's
 *    stream orders a recorded compaction in front of it: additions and one strengthening; nothing a caller built against 3 uses has changed. */
#define BMX_ABI_VERSION 4

/* status codes */
#define BMX_OK             0
#define BMX_ERR_INVALID   -1  /* bad argument */
#define BMX_ERR_HIP       -2  /* HIP runtime failure (text in bmx_last_error) */
#define BMX_ERR_FULL      -3  /* resident table cannot take the batch (capacity_rows exceeded) */
#define BMX_ERR_NOMEM     -4
#define BMX_ERR_RANGE     -5  /* a delta carried a reserved key, ts < 0 or > 2^53-1, or |val| > 2^53-1 */
#define BMX_ERR_INTERNAL  -6  /* device-side protocol fault (bounded spin expired) */
#define BMX_ERR_NO_DEVICE -7
#define BMX_ERR_NO_INDEX  -8  /* scan on a field with no index and auto-build disabled */
#define BMX_ERR_OVERFLOW  -9  /* a fixed-size exchange slab was too small (bmx_partition_by_owner_slabs): records were dropped */

/* where the caller's buffers live */
#define BMX_MEM_HOST   0      /* host pointers: the call copies in/out and is synchronous */
#define BMX_MEM_DEVICE 1      /* device pointers (same GPU): the call only enqueues work; outputs (including counts) are valid
                                 after bmx_sync(), or — for work the caller enqueues on the context's stream itself — behind
                                 bmx_merge_fence() (see "deferred compaction" below). Input arrays may be overwritten in stream
                                 order as soon as the call returns. */

/* what an absent key's first write stores as its clock */
#define BMX_INSERT_REFERENCE 0 /* ts := 2, exactly as src/bullet-crt.js:172-185 (+ :33-60) does */
#define BMX_INSERT_DELTA     1 /* ts := incoming ts (true last-writer-wins); used by bmx_load_rows */

/* Optional bit OR-ed into `insert_mode`: the caller guarantees that no two deltas of the batch share a key
 * (e.g. the JS host de-duplicated the batch while hashing paths). The per-row claim atomic is then skipped.
 * With duplicate keys in such a batch the result is unspecified (never unsafe): use it only when the
 * guarantee holds. The default path makes no assumption and is exact for any batch. */
#define BMX_MERGE_UNIQUE_KEYS 0x100

/* Optional bit OR-ed into `insert_mode`: exact SEQUENTIAL decision flags for every delta even when keys repeat inside the
 * batch (SURVEY §8(a) batch semantics (C)): flags[j] is what resolve() would have returned for delta j in the reference's
 * one-by-one loop. Final state and applied_idx are the same as without it; the batch takes roughly twice as long because
 * no delta is applied optimistically. Cannot be combined with BMX_MERGE_UNIQUE_KEYS. */
#define BMX_MERGE_STRICT_FLAGS 0x200

/* applied_idx[k] carries bit 31 (BMX_APPLIED_CREATED) when winner k is the delta that CREATED its row: the row then stores the insert rule's
 * clock ({id: 2} with BMX_INSERT_REFERENCE, src/bullet-crt.js:172-185), not the delta's own ts — what the host writes into meta[path].vectorClock
 * (src/bullet.js:196-201) without reading the row back. Mask with BMX_APPLIED_INDEX to get the index. */
#define BMX_MERGE_MARK_CREATED 0x1000
#define BMX_APPLIED_CREATED 0x80000000u
#define BMX_APPLIED_INDEX   0x00FFFFFFu

/* per-delta decision flags (bits of `flags[j]`), the booleans of resolve()'s decision record
 * src/bullet-crt.js:174-184. Exact for batches without duplicate keys (stats.n_conflicts == 0); with duplicates they are
 * relative to the state each delta observed, unless BMX_MERGE_STRICT_FLAGS is set (then always exact). */
#define BMX_FLAG_INCOMING   1u
#define BMX_FLAG_CURRENT    2u
#define BMX_FLAG_HISTORICAL 4u

typedef struct bmx_ctx bmx_ctx;

/* 32-byte delta record: the wire format of the sharded exchange and an alternative input layout */
typedef struct bmx_delta_rec {
  uint64_t id;
  uint32_t field;
  uint32_t aux;     /* carried through untouched */
  int64_t ts;
  int64_t val;
} bmx_delta_rec;

typedef struct bmx_merge_stats {
  uint64_t n_applied;    /* final winners: keys whose stored (ts,val) changed */
  uint64_t n_conflicts;  /* deltas that claimed a row another delta of the batch had claimed before them. 0 for a batch without duplicate keys; exact for
                          * rows created by the batch; otherwise a lower bound that can differ from run to run: a duplicate that is already below the
                          * value it sees drops out uncounted, and which of its predecessors' values it sees is a matter of timing */
  uint64_t n_rows;       /* resident rows after the batch */
  uint64_t reserved;
} bmx_merge_stats;

/* one term of a declarative filter: lo <= value(field) <= hi */
typedef struct bmx_term {
  uint32_t field;
  uint32_t reserved;
  int64_t lo, hi;
} bmx_term;

typedef struct bmx_info {
  uint64_t capacity_rows, n_slots, table_bytes, n_rows;
  uint32_t device, abi_version, n_indexes, epoch;
} bmx_info;

/* bmx_create flags */
#define BMX_CTX_FIXED_CAPACITY 2u /* never grow: a batch that would exceed capacity_rows fails with BMX_ERR_FULL.
                                     Default: the table is rehashed into one twice as large (synchronous, on device). */

/* Table placement tries (see bmx_get_placement): OR BMX_CTX_PLACEMENT_TRIES(n), n = 1..8, into the flags; 0 / absent = the default (4 at create,
 * 3 at a growth). Peak device memory of bmx_create: tries x the table (at most half of the free memory, whatever was asked); of a growth: the old table + tries x
 * the new one. 1 = take the first allocation (no transient memory, no timing). */
#define BMX_CTX_PLACEMENT_TRIES(n) (((uint32_t)(n) & 0xFu) << 8)

/* ---- lifetime -------------------------------------------------------------------------------
 * Replaces `new BulletCRT(bullet)` src/bullet-crt.js:6-16 / `new BulletQuery(bullet)`
 * src/bullet-query.js:2-7 for the device-resident part of the state (reference state:
 * bullet.store/meta src/bullet.js:28-31). capacity_rows bounds the resident rows. */
int bmx_create(int device, uint64_t capacity_rows, uint32_t flags, bmx_ctx** out);
/* Same with the table's maximum load factor (percent, 5..90; 0 = BMX_DEFAULT_LOAD_PCT): the table gets capacity_rows * 100 / max_load_pct
 * slots of 32 bytes. A 128-byte line is a bucket of four slots, so high load factors stay cheap to probe while the table shrinks
 * (more of it sits in the 256 MiB Infinity Cache); measured sweep: DESIGN.md §5. Tables that would need more than 2^32 slots
 * are refused with BMX_ERR_INVALID (slot indices in the per-batch workspace are 32-bit): shard the graph instead. */
#define BMX_DEFAULT_LOAD_PCT 50
int bmx_create_ex(int device, uint64_t capacity_rows, uint32_t max_load_pct, uint32_t flags, bmx_ctx** out);
void bmx_destroy(bmx_ctx* ctx);                     /* reference: Bullet.close() src/bullet.js:288-304 */
const char* bmx_last_error(const bmx_ctx* ctx);     /* ctx may be NULL for bmx_create failures */
int bmx_abi_version(void);
/* The merge kernel's exactness argument rests on one property the ISA does not promise: an aligned 16-byte load never
 * observes half of an aligned 16-byte store to the same address — the (ts,val) pair of a row. bmx_selfcheck hammers exactly that for a few
 * milliseconds on `device` (writers on half the workgroups, ~2*10^8 checked loads on the others, plain and L2-direct) and returns BMX_ERR_INTERNAL if one
 * torn pair is seen; *control_torn is the same run with the halves stored separately, which MUST tear (it proves the detector can see one).
 * bmx_create runs it once per device and process before it builds anything (BMX_SKIP_SELFCHECK=1 in the environment skips that). The reference
 * needs no counterpart: its writes are single-threaded JS assignments (src/bullet.js:184-201). Outputs may be NULL. */
int bmx_selfcheck(int device, uint64_t* reads, uint64_t* torn, uint64_t* control_torn);
int bmx_get_info(bmx_ctx* ctx, bmx_info* out);
/* Table placement. Where a large table lands in device memory decides ~10 % of the merge kernel's time (the same kernels on the same rows: 68-72 us per
 * 1M-delta launch on some allocations of a 1.4 GB table, 77-80 us on others made in the same process — for the allocation's lifetime). bmx_create and every
 * growth therefore allocate a table of >= 256 MB several times (four at create, three at a growth — BMX_CTX_PLACEMENT_TRIES —, and never more than half of the free device memory together), time each candidate with the merge kernel's own request mix (2^20 random slot reads, head
 * exchanges and 16-byte stores; ~0.25 ms per candidate) and keep the fastest (BMX_TABLE_PLACEMENT_TRIES=1 in the environment: take the first). This call
 * reports what was seen for the current table: number of candidates (0: not tuned), probe time of the chosen and of the slowest one, in us. The reference has
 * no counterpart (its store is a JS object, src/bullet.js:28). */
int bmx_get_placement(bmx_ctx* ctx, uint32_t* candidates, float* probe_us_chosen, float* probe_us_slowest);
/* Resident waves per SIMD the probe kernel may take: 8 (default: all), 6, 5, 4 or 3. The kernel is bound by memory-side requests in flight — a CU's miss
 * queue is full with far fewer waves — so fewer cost it nothing measurable, and kernels that are meant to run BESIDE it (the owner partition of the next
 * batch on the exchange stream, the deferred compaction) find wave slots on every CU at once instead of waiting for its one-wave workgroups to retire:
 * in the sharded pipeline k_part_count runs 17-33 us beside a 5-wave probe kernel and 58-72 us beside the full one (profiles/r04_sharded_timeline.log).
 * BMX_K1_WAVES in the environment sets the default of new contexts. */
int bmx_set_probe_waves(bmx_ctx* ctx, int waves_per_simd);
int bmx_sync(bmx_ctx* ctx);                         /* wait for the stream; returns a sticky device error if any */
int bmx_set_stream(bmx_ctx* ctx, void* hip_stream); /* run on the caller's hipStream_t (NULL = context's own) */
void* bmx_get_stream(bmx_ctx* ctx);
/* Cross-stream ordering for BMX_MEM_DEVICE callers without command-processor markers (no reference counterpart: the reference
 * is single threaded). Measured on MI355X: an event record, or a wait on a not-yet-complete event, between two kernels of one
 * stream holds that stream for ~9-10 us (bench_micro/event_gap*.hip); a 64-bit sequence word in device memory plus two tiny
 * kernels costs a launch (~2.5 us) and nothing on the other stream.
 *   bmx_seq_signal: enqueue on `hip_stream` a store of `value` to *seq_dev (device-scope release: everything enqueued on that
 *                   stream before it is visible to whoever observes the value).
 *   bmx_seq_wait:   enqueue on `hip_stream` a one-wave kernel that returns once *seq_dev >= at_least. The signal it waits for
 *                   must already be enqueued (on any stream) when this is called, otherwise the wait can never be satisfied; it
 *                   gives up after ~60 s and raises the context's sticky device error (bmx_sync then reports it).
 * seq_dev is 8-byte aligned device memory owned by the caller, zeroed before first use. Same GPU only. */
/* bmx_set_wait_limit: how long every device-side wait of this library on the context's DEVICE (all contexts of the process on that GPU) polls before it gives up:
 * default ~60 s. Once one wait of a context has given up, the waits still queued on that context return at once (the sticky error is reported by bmx_sync as ever).
 * The sharded bench sets 20 s: a peer that is 20 s late for a 100-us step is gone, and the ranks must reach their agreed fallback inside the collective time-out. */
int bmx_set_wait_limit(bmx_ctx* ctx, double seconds);
int bmx_seq_signal(bmx_ctx* ctx, void* hip_stream, uint64_t* seq_dev, uint64_t value);
int bmx_seq_wait(bmx_ctx* ctx, void* hip_stream, const uint64_t* seq_dev, uint64_t at_least);

/* ---- merge ----------------------------------------------------------------------------------
 * bmx_load_rows: bulk preload of resident rows with their clocks (what storage load / initial sync
 * does through setData: src/bullet-file-storage.js:96-163). True LWW against anything already there.
 *
 * bmx_merge_batch: one batch of deltas through conflict resolution. Replaces calling
 * crt.handleUpdate()/resolve() once per entry (src/bullet-crt.js:164-279, 329-385) in the
 * sequential loop of src/bullet-network-sync.js:551-569, with the same final state and the same
 * final winner per key as that loop (deltas taken in index order):
 *   - final (ts,val) of a key = lexicographic max over its resident row and its deltas,
 *   - applied_idx = ascending indices j of the deltas whose value is finally stored, one per changed
 *     key (the smallest index attaining the key's maximum; nothing if the key did not change),
 *   - an absent key is created by its smallest-index delta with ts := 2 (BMX_INSERT_REFERENCE).
 * n <= 2^24 per call. applied_idx (capacity n), n_applied, flags (capacity n) and stats may be NULL.
 * With BMX_MEM_DEVICE, n_applied and stats are device pointers too. */
int bmx_load_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                  const int64_t* val, int mem);
int bmx_merge_batch(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                    const int64_t* val, int insert_mode, int mem, uint32_t* applied_idx, uint64_t* n_applied,
                    uint8_t* flags, bmx_merge_stats* stats);
/* Pipelined form of bmx_merge_batch(BMX_MEM_HOST) for a host that streams batches (the sync replay of
 * src/bullet-network-sync.js:551-569 with chunks of 10^5..10^6 entries): bmx_merge_submit uploads the batch and enqueues its merge,
 * bmx_merge_collect returns that batch's winners / flags / stats. Up to TWO batches may be in flight, so the upload of batch b+1
 * (PCIe, the expensive part of a host batch) runs while the GPU merges batch b; results arrive one batch late. Batches are applied
 * in submission order; collect in the same order. The input arrays may be reused as soon as bmx_merge_submit returns.
 * bmx_merge_batch(BMX_MEM_HOST) is exactly submit + collect. want_flags = 0 skips the per-delta flag transfer. */
int bmx_merge_submit(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int insert_mode,
                     int want_flags, uint64_t* ticket);
int bmx_merge_collect(bmx_ctx* ctx, uint64_t ticket, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats);
/* Page-locked host memory for the arrays handed to BMX_MEM_HOST calls (inputs and outputs): copies to and from it run at the link's rate and
 * without the runtime's own staging, where pageable memory costs an extra pass (a 1M-delta host batch: 0.76 ms against 1.2 ms on MI355X, and
 * ~17 ms the first time a fresh pageable array is seen). A BMX_MEM_HOST scan whose out buffer lies in such memory (or in any page-locked range the runtime
 * knows) has its answer written there by the kernels themselves — no staging copy behind the answer. Usable with every context of the process, on any device. Nothing requires it:
 * any host pointer is accepted everywhere. (The reference has no counterpart: its batches are JS objects, src/bullet-network-sync.js:551-569.) */
int bmx_host_alloc(uint64_t bytes, void** host_ptr);
int bmx_host_free(void* host_ptr);
/* same, deltas as 32-byte records (device pointers only): the receive side of the sharded exchange */
int bmx_merge_records(bmx_ctx* ctx, uint64_t n, const bmx_delta_rec* recs, int insert_mode, uint32_t* applied_idx,
                      uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats);

/* Deferred compaction (BMX_MEM_DEVICE merges of >= 65536 deltas on the default path; on by default). The reference filters the winners of a
 * chunk while it applies it (src/bullet-crt.js:383, src/bullet.js:144-146); here that filter is the last of three launches
 * (k_compact_winners -> applied_idx / n_applied / stats) and reads nothing but per-batch workspace. When batch b + 1 follows batch b directly,
 * the compaction of b runs on a second, high-priority stream UNDER the probe kernel of b + 1 instead of in front of it (ordering by two words in
 * device memory, no event markers); with any other bmx_* call in between it runs on the context's stream as before. Consequences for a caller:
 *   - outputs of a device merge are valid after bmx_sync() (as ever) or, in stream order, behind any other call on the context that enqueues
 *     or reads something — every entry point except the ones listed next (bmx_destroy included: it launches a compaction that was only recorded and
 *     waits for it). bmx_seq_signal on the context's own stream counts: it publishes "everything before this is done", so it orders the compaction
 *     in front of the signal. NOT ordering points, because they neither publish nor read a merge's outputs: another device merge, bmx_seq_wait,
 *     bmx_seq_wait_all, bmx_seq_signal on a stream that is not the context's, bmx_merge_tail_wait, bmx_set_probe_waves, bmx_timer_elapsed,
 *     bmx_get_deferred_counts, bmx_get_placement, bmx_index_refresh_counts, bmx_index_ordered_info, bmx_last_error;
 *   - a caller that enqueues its own work on the context's stream (bmx_set_stream) and reads applied_idx / n_applied / stats there without
 *     bmx_sync() calls bmx_merge_fence() first: it only enqueues, and orders the stream behind every compaction;
 *   - bmx_set_deferred_compaction(ctx, 0) restores strict stream order for every launch (the communicator does this for its shards).
 * The side stream's compaction is released by a one-wave kernel that waits for the next probe kernel to START, so it needs kernels of two
 * streams to run side by side. A process in which every dispatch is serialised — rocprofv3 counter collection (--pmc; ROCPROF_COUNTER_COLLECTION in
 * the environment), HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL, or BMX_NO_DEFERRED_COMPACTION=1 — never defers (bmx_set_deferred_compaction(ctx, 1)
 * is ignored there); kernel traces without counters (rocprofv3 --kernel-trace --stats) do not serialise and keep it on.
 * bmx_get_deferred_counts: merges whose compaction was deferred / of those, how many actually ran on the side stream. */
int bmx_set_deferred_compaction(bmx_ctx* ctx, int on);
/* The stream the deferred compactions run on. Default (NULL): a high-priority stream the context creates. A caller that already drives a second stream beside
 * the merges (the sharded pipeline's exchange stream: bullet-js_amd/bmx/sharded.py) may hand that one in, so that no third hardware queue competes with the two
 * it has; the compaction of batch b then runs on it behind whatever the caller enqueued there before the merge of batch b + 1. Synchronises both streams once. */
int bmx_set_side_stream(bmx_ctx* ctx, void* hip_stream);
int bmx_merge_fence(bmx_ctx* ctx);
int bmx_get_deferred_counts(bmx_ctx* ctx, uint64_t* deferred, uint64_t* on_side_stream);

/* bmx_put_rows: rows whose outcome was decided ELSEWHERE are stored as given (no comparison with the resident row; absent rows are created
 * with the given ts). This is how the host keeps the device table in step with writes it resolved itself — single puts, values or clocks the
 * device cannot compare (src/bullet-crt.js:329-385 run on the host), and node-level writes: a dominating object REPLACES the node, so the
 * fields it no longer carries are removed (src/bullet-crt.js:236-248), and `deleted` sync entries become setData(path, null)
 * (src/bullet-network-sync.js:553-555). val == BMX_VAL_DELETED leaves a TOMBSTONE: the key keeps its slot and clock, but the row is no data —
 * index builds, scans, filters and dumps skip it exactly as _addToIndex skips null values (src/bullet-query.js:83-85); bmx_get_rows reports it
 * found with val == BMX_VAL_DELETED; a later merge or put of a real value brings it back (against a tombstone any delta with ts >= the
 * tombstone's wins). Keys must be unique within one call (the caller keeps the last write per key). Maintained indexes follow through the
 * change log like after any merge. */
#define BMX_VAL_DELETED INT64_MIN
int bmx_put_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int mem);

/* ---- point reads ----------------------------------------------------------------------------
 * Replaces bullet._getData(path) + bullet.meta[path].vectorClock (src/bullet.js:115-129,
 * src/bullet-crt.js:331-333) for device-resident rows. found[i] = 1/0. */
int bmx_get_rows(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, int64_t* ts, int64_t* val,
                 uint8_t* found, int mem);
int bmx_get_row(bmx_ctx* ctx, uint64_t id, uint32_t field, int64_t* ts, int64_t* val); /* 1 found, 0 absent, <0 error */
/* all resident rows that hold data (tombstones are left out), unordered; *n_out = their number even if cap is smaller (checkpoint hook:
 * src/bullet-network-sync.js:592-664 _collectFullSyncData, which skips deleted entries the same way) */
int bmx_dump_rows(bmx_ctx* ctx, uint64_t cap, uint64_t* id, uint32_t* field, int64_t* ts, int64_t* val,
                  uint64_t* n_out, int mem);
int bmx_row_count(bmx_ctx* ctx, uint64_t* n_out);   /* keys holding a slot: rows + tombstones (what counts against capacity_rows) */
/* Make room for at least capacity_rows resident rows (no-op if already there): allocates a new table, re-inserts every
 * row on the device, frees the old one. Synchronous. The reference's store is a JS object that simply grows
 * (src/bullet.js:28); this is the device-side equivalent. */
int bmx_reserve(bmx_ctx* ctx, uint64_t capacity_rows);

/* ---- index + scans --------------------------------------------------------------------------
 * bmx_index_build replaces BulletQuery.index(path, field) / _buildIndex (src/bullet-query.js:30-73):
 * it materialises dense columns (node id, value) of the rows carrying `field`. Scans always see the
 * FRESH index state (a stale index is rebuilt before the scan), which is the state in which the
 * reference's range()/equals() equal a ground-truth scan (SURVEY §8(a) "Scan parity target").
 *
 * bmx_scan_range replaces range(path, field, min, max) src/bullet-query.js:221-261: lo <= val <= hi,
 * both inclusive. bmx_scan_equals replaces equals() :186-210, bmx_scan_count replaces count() :293-313.
 * Results are node ids in index-column order unless the index has a value-ordered view (below) (deterministic for a given history of calls: table order for the rows
 * present at the last full build, then creation order); the host mirror maps ids back to paths and can reproduce the
 * reference's first-seen-value order.
 *
 * Maintenance (the device-side _updateIndices, src/bullet-query.js:82-110): while an index exists, every merge on the default path
 * logs its winners' rows; the next scan applies that log to the dense columns (created rows appended, changed rows overwritten)
 * instead of rebuilding them from the table. A full rebuild still happens after a table growth, after merges that do not log
 * (BMX_MERGE_UNIQUE_KEYS, BMX_MERGE_STRICT_FLAGS), when the log outgrows an eighth of the table, with more than 8
 * indexes, or for tables of 2^31 slots and more. bmx_index_refresh_counts reports how often each happened.
 * out_ids may be NULL (count only). *n_out = number of matches even if cap is smaller. */
int bmx_index_build(bmx_ctx* ctx, uint32_t field);
int bmx_index_drop(bmx_ctx* ctx, uint32_t field);
int bmx_index_size(bmx_ctx* ctx, uint32_t field, uint64_t* n_out);   /* positions in the index columns (rows of the field, tombstoned ones included: they keep their position and match nothing) */
int bmx_index_refresh_counts(bmx_ctx* ctx, uint64_t* full_builds, uint64_t* incremental_updates);
/* Value-ordered view of an index (opt-in per index; ABI 3). The reference keeps an index as a Map keyed by VALUE (src/bullet-query.js:30-73): equals() is one
 * lookup (:186-210), range() walks the distinct values (:221-261) — neither ever touches a child that does not match. The dense columns above cost one pass
 * over ALL rows per query instead. bmx_index_set_ordered(ctx, field, N >= 1) gives the index the reference's shape: a second copy of its columns sorted by
 * (value, position). While that view is current, bmx_scan_range / _equals / _count / _range_pos on the field are two k-ary searches plus one contiguous copy —
 * O(log R + matches), nothing read that is not part of the answer — and deliver the matches in (value, position) order instead of position order (the set
 * is the same; the host mirror can reproduce the reference's first-seen-value order from either).
 * Currency (ABI 4: the device-side _updateIndices, src/bullet-query.js:139-176 — the reference moves a path from the bucket of its old value to the bucket of
 * its new one on every write): a view that is current STAYS current under writes. The refresh that brings the dense columns up to date from the merges' change
 * log also captures the change run — (position, old value) of every row whose value really changed, plus the appended rows — and sorts it (hand-written LDS tile
 * sort + merge-path passes, csrc/view_kernels.h). What happens to the sorted run depends on its size against the view (R rows):
 *   - more than R/4 keys: the view's main run is rewritten at once, in front of the answer: one streaming pass main - deleted + inserted, R * (w + 12) bytes
 *     read and written (w = 4 or 8);
 *   - otherwise the run joins the view's PENDING PATCH (pd: keys deleted from main, pi: keys inserted, each sorted; a deleted key that is a pending insert cancels
 *     it). Queries answer from main - pd + pi: five k-ary searches instead of two, and the copy skips / appends the patch's keys of the answer's range. When the
 *     patch has grown beyond R/16 keys, main is rewritten with it BEHIND the answer of the query that found it so: the rewrite is enqueued after that answer, reads
 *     main and the patch and writes the second set of columns; the next call that touches the view looks at its completion event and error word and swaps the sets
 *     (or, after a failure, goes on with main and the patch, which were never written). A run of R/16 ... R/4 keys makes the rewrite due at once: the same
 *     work as a rewrite in front of the answer, but behind it (10^7 rows, 1M-delta merges: every answer is followed by a 0.23-ms rewrite). At 10^8 rows and 1M-delta merges on the field: every first query after a
 *     merge pays the run's sort and the join (0.53-0.81 ms with its answer in host memory), every fourth is followed by a 1.3-1.8 ms rewrite that nobody waits for
 *     unless the next merge + query arrive within that time.
 * All of it is paid by the first query after any number of merges on the field, never by the merges; merges on other fields, merges that lose and rewrites of
 * the same value cost nothing. ORDER of a view's answers: survivors of main in (value, position) order, then the range's pending inserts in (value, position)
 * order — the same SET as ever, as one sorted run only while the patch is empty; the order may differ between two calls (a rewrite in between). Memory: a second
 * set of the view's columns from the first patch on (the two sets swap), i.e. 32 or 40 bytes per row in all, the pending patch (two sets, R/16 + two runs of keys:
 * ~5 bytes per row), + 12 bytes per logged winner for the change run. BMX_VIEW_PENDING=0 in the environment switches the pending patch off (every patch rewrites
 * main; A/B switch). The view goes STALE (and is sorted again from scratch, below) only when it cannot be patched: after an index rebuild (table growth, merges
 * that do not log, a log longer than an eighth of the table or 2^24 entries), when a value stops fitting int32 (the index switches columns), without memory for
 * the second set or the patch, or with BMX_VIEW_PATCH=0 in the environment (A/B switch).
 * A stale (or new) view is sorted (one radix sort of the column + one gather: milliseconds for 10^8 rows, csrc/ordered_sort.hip) by the N-th query since the
 * columns last changed; the N - 1 queries before it scan the column as ever. N = 0 switches the view off and frees it. bmx_scan_filter takes its candidates from
 * the view of its FIRST term's index when there is one: the other terms are probed for the ids of one run only (survivors then come in no particular order).
 * If the memory cannot be had the index silently goes on without the view. bmx_index_ordered_info: N, whether the view would answer the next query, and how many
 * sorts have run; bmx_index_ordered_stats: sorts, patches, keys moved by the patches (deleted + inserted), what the last sort / the last patch cost the
 * caller in microseconds (the patch: the host's time in it — a run that joins the pending patch is not waited for, the query behind it on the stream is), rewrites of
 * main that have completed, and the keys in the pending patch now (outputs may be NULL). */
#define BMX_INDEX_ORDERED_AUTO 0xFFFFFFFFu   /* after_queries chosen by the engine: sort once the scans since the change have cost what the sort costs (rent-or-buy:
                                               * never more than twice the cheapest schedule, whatever comes next): ~70 queries on 10^8 int32 rows, ~25 on 10^7 */
int bmx_index_set_ordered(bmx_ctx* ctx, uint32_t field, uint32_t after_queries);
int bmx_index_ordered_info(bmx_ctx* ctx, uint32_t field, uint32_t* after_queries, int* valid_now, uint64_t* sorts);
int bmx_index_ordered_stats(bmx_ctx* ctx, uint32_t field, uint64_t* sorts, uint64_t* patches, uint64_t* keys_patched, double* last_sort_us, double* last_patch_us,
                            uint64_t* rewrites, uint64_t* pending_keys);
int bmx_scan_range(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap,
                   uint64_t* n_out, int mem);
int bmx_scan_equals(bmx_ctx* ctx, uint32_t field, int64_t value, uint64_t* out_ids, uint64_t cap, uint64_t* n_out,
                    int mem);
int bmx_scan_count(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint64_t* n_out, int mem);
/* Position output (same query as bmx_scan_range, src/bullet-query.js:221-261): out_pos[k] = the k-th match's POSITION in the index columns
 * (u32, ascending) instead of its node id. The emit pass then never touches the id column — from ~6 % selectivity on, gathering ids reads
 * most lines of that column and bounds the whole scan (profiles/traffic_scan.json). For callers that mirror something per index row
 * (the JS host keeps child paths by position: GpuQuery): fetch the id column once with bmx_index_ids, map positions thereafter.
 * Positions are stable while the index is maintained from the change log (rows only get appended); a full rebuild
 * (bmx_index_refresh_counts: full_builds changes) renumbers them. */
int bmx_scan_range_pos(bmx_ctx* ctx, uint32_t field, int64_t lo, int64_t hi, uint32_t* out_pos, uint64_t cap,
                       uint64_t* n_out, int mem);
/* ids[first .. first+count) of the index columns of `field` (brought up to date first), in position order. */
int bmx_index_ids(bmx_ctx* ctx, uint32_t field, uint64_t first, uint64_t count, uint64_t* out_ids, int mem);
/* Declarative subset of filter(path, fn) src/bullet-query.js:270-283: fn = AND of range terms over
 * fields of the same node. Arbitrary JS predicates stay on the host. */
int bmx_scan_filter(bmx_ctx* ctx, uint32_t nterms, const bmx_term* terms, uint64_t* out_ids, uint64_t cap,
                    uint64_t* n_out, int mem);

/* ---- sharding (one context per GPU; rows owned by bmx_owner_of(id, nshards)) -----------------
 * Replaces the gossip fan-out of src/bullet-network.js:378-418 inside one node: instead of every
 * peer merging every delta, each delta is routed to the shard that owns its node id.
 * bmx_partition_by_owner: stable partition of a delta batch into `nshards` contiguous runs of
 * 32-byte records in recs_out (device, capacity n), counts[nshards] (device or host per `mem`
 * of the counts pointer: always device here; copy it yourself or use bmx_copy_counts). */
uint32_t bmx_owner_of(uint64_t id, uint32_t nshards);
int bmx_partition_by_owner(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                           const int64_t* val, uint32_t nshards, bmx_delta_rec* recs_out, uint64_t* counts_out_dev);
/* Same, into fixed-size slabs: shard g's records go to recs_out[g*slab_records ...] and the rest of each slab
 * is padding (id = 0xFFFFFFFFFFFFFFFF, skipped by bmx_merge_records), so the exchange can use equal splits and
 * needs no host round trip for counts. recs_out has nshards*slab_records records. counts_out_dev[g] is the TRUE
 * count: if it exceeds slab_records the surplus records of that shard were NOT written and the caller must
 * re-route the batch with bmx_partition_by_owner (re-delivery is harmless for existing rows: their state is a lexmax; a re-delivered
 * first write of an absent key meets the row its first delivery created, as in the reference when a sync chunk arrives twice). The overflow is also
 * sticky on the context: the next bmx_sync() returns BMX_ERR_OVERFLOW, so a pipelined caller cannot miss it. */
int bmx_partition_by_owner_slabs(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                                 const int64_t* val, uint32_t nshards, uint64_t slab_records, bmx_delta_rec* recs_out,
                                 uint64_t* counts_out_dev);

/* ---- one process, N shards (bmx_comm_*) --------------------------------------------------------
 * The multi-GPU entry of the product surface (SURVEY §8(b) sketch bmx_comm_create / bmx_merge_batch_sharded; §8(e)): one handle owns
 * N contexts, one per GPU of the node — or several per GPU ("logical shards": devices[] may repeat a device, which is how a one-GPU
 * machine exercises N = 2, 4, 8). Replaces the gossip fan-out of src/bullet-network.js:378-418: a delta is merged once, by the shard
 * that owns its node id, instead of once by every peer. A `Bullet` instance in Node owns all shards through the N-API binding of
 * these calls (commCreate / commMerge / commScanRange ...). Not re-entrant. GPUs of one communicator must be peer-accessible.
 *
 * bmx_comm_merge: a HOST batch, same contract as bmx_merge_batch(BMX_MEM_HOST) on one context: final rows = the reference's sequential
 *   loop over the batch, applied_idx = ascending indices (into the caller's batch) of the deltas whose value is finally stored.
 *   (The batch is cut into N slices; slice i is partitioned by owner on shard i; runs are copied device-to-device to the owners in
 *   origin order, so every shard sees its deltas in ascending batch order.) stats->n_rows = rows over all shards.
 * bmx_comm_load_rows: bulk preload (true LWW), routed the same way.
 * bmx_comm_merge_dev: every shard ORIGINATES a device-resident batch (n[i] deltas, pointers on shard i's GPU): its owner partition
 *   writes fixed-size slabs of slab_records records (0 = mean run + 12.5 % + 64) straight into the owners' receive buffers — peer-
 *   mapped stores over xGMI, no copy kernel, no host round trip — and every shard merges its N slabs. Enqueue-only: call
 *   bmx_comm_sync() before reading results. A slab that was too small makes that sync return BMX_ERR_OVERFLOW (records were
 *   dropped: re-send the step through bmx_comm_merge; re-delivery is harmless for existing rows, whose state is a lexmax). Order inside a shard: origin 0's run, origin 1's, ...
 * bmx_comm_shard_result: device pointers to what shard g merged in the last bmx_comm_merge_dev step and to its winners (positions in
 *   that record array), valid after bmx_comm_sync().
 * bmx_comm_scan_*: replaces range()/equals()/count()/declarative filter() (src/bullet-query.js:186-313) on the sharded graph: every
 *   shard scans its own rows, results are concatenated in shard order. No collective. The scans of all shards are enqueued before the
 *   first result is fetched, so N GPUs scan at the same time.
 * Every bmx_comm_* call restores the calling thread's current HIP device before it returns (single-context calls leave the context's
 * device current). */
typedef struct bmx_comm bmx_comm;
int bmx_comm_create(uint32_t nshards, const int* devices, uint64_t capacity_rows_per_shard, uint32_t flags, bmx_comm** out);
void bmx_comm_destroy(bmx_comm* comm);
const char* bmx_comm_last_error(const bmx_comm* comm);
uint32_t bmx_comm_nshards(const bmx_comm* comm);
bmx_ctx* bmx_comm_shard(bmx_comm* comm, uint32_t shard);   /* borrowed: point reads, index handling, info of one shard */
int bmx_comm_sync(bmx_comm* comm);
int bmx_comm_load_rows(bmx_comm* comm, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val);
/* bmx_put_rows over the shards: every row goes to the shard that owns its node (host buffers). */
int bmx_comm_put_rows(bmx_comm* comm, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val);
int bmx_comm_merge(bmx_comm* comm, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, int insert_mode,
                   uint32_t* applied_idx, uint64_t* n_applied, bmx_merge_stats* stats);
int bmx_comm_merge_dev(bmx_comm* comm, const uint64_t* n, const uint64_t* const* id, const uint32_t* const* field, const int64_t* const* ts,
                       const int64_t* const* val, int insert_mode, uint64_t slab_records);
int bmx_comm_shard_result(bmx_comm* comm, uint32_t shard, const bmx_delta_rec** recs_dev, const uint32_t** applied_dev, const uint64_t** n_applied_dev,
                          const bmx_merge_stats** stats_dev, uint64_t* n_records);
int bmx_comm_row_count(bmx_comm* comm, uint64_t* n_out);
int bmx_comm_get_rows(bmx_comm* comm, uint64_t n, const uint64_t* id, const uint32_t* field, int64_t* ts, int64_t* val, uint8_t* found);
int bmx_comm_dump_rows(bmx_comm* comm, uint64_t cap, uint64_t* id, uint32_t* field, int64_t* ts, int64_t* val, uint64_t* n_out);
int bmx_comm_index_build(bmx_comm* comm, uint32_t field);
int bmx_comm_index_set_ordered(bmx_comm* comm, uint32_t field, uint32_t after_queries);   /* bmx_index_set_ordered on every shard (results: shard by shard, each in (value, position) order) */
int bmx_comm_scan_range(bmx_comm* comm, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap, uint64_t* n_out);
int bmx_comm_scan_equals(bmx_comm* comm, uint32_t field, int64_t value, uint64_t* out_ids, uint64_t cap, uint64_t* n_out);
int bmx_comm_scan_count(bmx_comm* comm, uint32_t field, int64_t lo, int64_t hi, uint64_t* n_out);
int bmx_comm_scan_filter(bmx_comm* comm, uint32_t nterms, const bmx_term* terms, uint64_t* out_ids, uint64_t cap, uint64_t* n_out);

/* ---- N4: fixed-K multi-writer vector clocks -------------------------------------
 * A second kind of table for rows whose clocks have up to 8 writers: {writer_0: c0, ..., writer_{K-1}: c_{K-1}} with small
 * non-negative integer components and integer values. Replaces resolve() in full for such rows — dominance test
 * (compareVectorClocks src/bullet-crt.js:68-95, missing component = 0), component-wise max merge (:103-114), the
 * "identical clocks -> value comparison" branch (:200-233), and the "concurrent" branch whose value is
 * compare(in,cur) >= 0 ? in : cur (:266-278, :133-135) — including the quirk that a first write stores the one-key clock
 * {local: 2} (:172-185). Clocks that name all K writers in the table's order need nothing more (the plain entry points); clocks that name
 * only SOME of them, or in another order — the reference's clocks are JS objects: a missing key counts as 0 in the dominance test (:76-79), two
 * clocks are "identical" only when their JSON texts are (:200-203: same keys, same order, same counters), a merged clock lists the incoming
 * clock's keys first and then the stored clock's other keys (:103-114) — carry a KEY SET word per clock (the *_ks entry points): eight 4-bit
 * writer indices in the object's key order, 0xF = end (BMX_VC_KEYSET_NONE = the clock {}); components of writers a clock does not name must
 * be 0. A row stores its clock's key set next to the counters and bmx_vc_get_rows_ks returns it, so the host rebuilds the very object the
 * reference would hold. Writers outside the table's K stay on the host.
 * Deltas of one key are applied in index order (the result is order dependent for concurrent clocks), so batches are
 * exact for any duplication. flags[j] additionally carries BMX_FLAG_CONCURRENT. updated_idx = ascending indices of the
 * last delta per key that caused a store (doUpdate: src/bullet-crt.js:383). These entry points take host buffers (synchronous). capacity_rows
 * is the initial size: the table grows by a device-side rehash whenever a batch could push the load factor above 0.5. */
#define BMX_FLAG_CONCURRENT 8u
#define BMX_VC_MAX_WRITERS 8
#define BMX_VC_ABSENT 0
#define BMX_VC_DENSE  1   /* the row's clock was loaded or merged at least once (its keys: the row's key set; all K writers for the plain entry points) */
#define BMX_VC_SPARSE 2   /* the row still carries the one-key clock {local: 2} of its first write */
#define BMX_VC_KEYSET_NONE 0xFFFFFFFFu
/* key set of a clock whose keys are writers w[0..count) in that order; bmx_vc_keyset_dense(K) = all K writers in the table's order (pure functions) */
uint32_t bmx_vc_keyset(const uint8_t* w, uint32_t count);
uint32_t bmx_vc_keyset_dense(uint32_t k_writers);
typedef struct bmx_vc bmx_vc;
int bmx_vc_create(int device, uint64_t capacity_rows, uint32_t k_writers, uint32_t local_writer, bmx_vc** out);
void bmx_vc_destroy(bmx_vc* t);
const char* bmx_vc_last_error(const bmx_vc* t);
int bmx_vc_load_rows(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks /* n*K */,
                     const int64_t* val);
int bmx_vc_merge_batch(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks /* n*K */,
                       const int64_t* val, uint32_t* updated_idx, uint64_t* n_updated, uint8_t* flags);
int bmx_vc_get_rows(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, uint32_t* clocks_out /* n*K */,
                    int64_t* val_out, uint8_t* state_out);
/* The same with a key set per clock (keysets: n words, NULL = every clock names all K writers in order; keysets_out may be NULL).
 * A malformed key set (index >= K, a writer twice, a non-zero component of an unnamed writer) is BMX_ERR_RANGE. */
int bmx_vc_load_rows_ks(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks /* n*K */, const uint32_t* keysets,
                        const int64_t* val);
int bmx_vc_merge_batch_ks(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks /* n*K */, const uint32_t* keysets,
                          const int64_t* val, uint32_t* updated_idx, uint64_t* n_updated, uint8_t* flags);
int bmx_vc_get_rows_ks(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, uint32_t* clocks_out /* n*K */, uint32_t* keysets_out,
                       int64_t* val_out, uint8_t* state_out);
int bmx_vc_row_count(bmx_vc* t, uint64_t* n_out);
/* range()/equals()/count() (src/bullet-query.js:186-313) over the rows of this table: node ids of the rows of `field` with lo <= val <= hi, in
 * table order; out_ids may be NULL (count only); *n_out = matches even if cap is smaller. Scans the table itself (64 B per slot), host buffers. */
int bmx_vc_scan_range(bmx_vc* t, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap, uint64_t* n_out);
/* Device-pointer form (all pointers are device memory; enqueue-only on the table's stream, like BMX_MEM_DEVICE for the scalar table):
 * n_updated is a device uint64; updated_idx (capacity n) and flags (n) may be NULL. Domain errors and protocol faults are sticky and
 * reported by bmx_vc_sync. bmx_vc_set_stream: run on the caller's hipStream_t (NULL = the table's own). */
int bmx_vc_merge_batch_dev(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const int64_t* val,
                           uint32_t* updated_idx, uint64_t* n_updated, uint8_t* flags);
int bmx_vc_merge_batch_ks_dev(bmx_vc* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const uint32_t* keysets,
                              const int64_t* val, uint32_t* updated_idx, uint64_t* n_updated, uint8_t* flags);
int bmx_vc_set_stream(bmx_vc* t, void* hip_stream);
int bmx_vc_sync(bmx_vc* t);

/* ---- direct exchange between PROCESSES, one per GPU (what `bench.py --gpus N` runs; replaces the gossip fan-out of src/bullet-network.js:378-418
 * inside a node, like the all-to-all it supersedes): every rank owns receive slabs other ranks write into.
 *   bmx_ipc_alloc  device memory (zeroed; flags: BMX_IPC_UNCACHED) + a 64-byte handle another process opens with bmx_ipc_open (peer_device: the GPU it lives on as THIS
 *                  process numbers it, or -1 for "the same GPU"); bmx_ipc_close / bmx_ipc_free undo them (the stream is synchronised first).
 *   bmx_partition_scatter  bmx_partition_by_owner_slabs with one destination per shard: slab g is written to dst[g] — a pointer into shard g's
 *                  receive memory, this process's or a mapped one — and, from a one-wave launch behind the scatter (its kernel boundary is the
 *                  release), arrive_words[g] (a word in shard g's memory, may be NULL) is set to arrive_value. No copy kernel, no collective,
 *                  no second stream.
 *                  wait_words_dev (may be NULL): bmx_seq_wait_all on these words first (the owners have freed the slab set). One host call per route.
 *   bmx_merge_records_after  bmx_seq_wait_all on the words (the origins' slabs have arrived), then bmx_merge_records: one host call per merge.
 *   bmx_seq_wait_all  one-wave kernel on the stream: returns once every one of nwords consecutive words (this GPU's memory) is >= at_least;
 *                  expires like bmx_seq_wait.
 *   bmx_merge_notify  from now on every bmx_merge_records_after of this context — the merges that read a receive slab set, and only those — stores,
 *                  as its last act, the number of such merges finished since this call into each of the given words (other ranks' memory:
 *                  "your slab set k is free again"). Other merges on the same context (a bmx_merge_batch in between) do not move the
 *                  words. nwords = 0 switches it off. */
#define BMX_IPC_UNCACHED 1u   /* never cached in this GPU's L2: for memory that peers store into (receive slabs, arrival / free words) */
int bmx_ipc_alloc(bmx_ctx* ctx, uint64_t bytes, uint32_t flags, void** dev_ptr, uint8_t handle_out[64]);
int bmx_ipc_open(bmx_ctx* ctx, const uint8_t handle[64], int peer_device, void** dev_ptr);
int bmx_ipc_close(bmx_ctx* ctx, void* dev_ptr);
int bmx_ipc_free(bmx_ctx* ctx, void* dev_ptr);
int bmx_partition_scatter(bmx_ctx* ctx, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val,
                          uint32_t nshards, uint64_t slab_records, void* const* dst, uint64_t* counts_out_dev, uint64_t* const* arrive_words,
                          uint64_t arrive_value, const uint64_t* wait_words_dev, uint32_t n_wait, uint64_t wait_at_least);
int bmx_merge_records_after(bmx_ctx* ctx, const uint64_t* wait_words_dev, uint32_t n_wait, uint64_t wait_at_least, uint64_t n, const bmx_delta_rec* recs,
                            int insert_mode, uint32_t* applied_idx, uint64_t* n_applied, uint8_t* flags, bmx_merge_stats* stats);
int bmx_seq_wait_all(bmx_ctx* ctx, void* hip_stream, const uint64_t* words_dev, uint32_t nwords, uint64_t at_least);
/* bmx_merge_tail_wait: the wait of the NEXT batch, folded into THIS merge. Armed before a merge call, it makes that merge's resolve kernel (one wave
 * of it) return only once the nwords words are >= at_least; a following bmx_merge_records_after that asks for the same (or a weaker) wait on the same
 * words then launches no wait kernel: its probe kernel follows the resolve kernel directly (one launch and its boundary less per step of the
 * pipeline). Arm it only when the batch those words belong to has been routed already — like every wait here it expires after ~60 s otherwise.
 * One shot: it applies to the next merge call only (and is dropped by merges on paths that have no resolve kernel). */
int bmx_merge_tail_wait(bmx_ctx* ctx, const uint64_t* words_dev, uint32_t nwords, uint64_t at_least);
int bmx_merge_notify(bmx_ctx* ctx, uint64_t* const* words, uint32_t nwords);

/* ---- timing helpers (HIP events on the context's stream; used by bench.py) ------------------- */
int bmx_timer_start(bmx_ctx* ctx);
int bmx_timer_stop(bmx_ctx* ctx, float* ms_out);    /* synchronises on the stop event */
int bmx_timer_mark(bmx_ctx* ctx);                   /* bmx_timer_stop in two halves: record the stop event now (enqueue only) ... */
int bmx_timer_elapsed(bmx_ctx* ctx, float* ms_out); /* ... and read the time between start and mark later (synchronises on the mark) */
/* Per-kernel timing of the merge: while enabled, every merge call brackets its three stages with HIP
 * events (up to 64 calls are kept). bmx_profile_read synchronises and returns the AVERAGE milliseconds
 * per call of: [0] k_probe_apply, [1] k_resolve_lists, [2] winner compaction, and the number of calls averaged */
int bmx_profile_enable(bmx_ctx* ctx, int on);
int bmx_profile_read(bmx_ctx* ctx, float ms_out[3], uint32_t* n_calls);
/* Same for the scans issued while profiling was enabled: AVERAGE milliseconds per scan call of [0] the mask pass (k_scan_mask: the one
 * read of the value column), [1] everything after it (offset scan + k_scan_emit, or the count reduction). */
int bmx_profile_read_scan(bmx_ctx* ctx, float ms_out[2], uint32_t* n_calls);

#ifdef __cplusplus
}
#endif
#endif /* BMX_H */
