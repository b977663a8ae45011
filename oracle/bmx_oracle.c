/*
 * bmx_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's conflict-resolution
 * and index-scan semantics for the scalar-clock contract.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported baseline. The product path
 * (libbmx.so, HIP) never links or calls it.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks every function here
 * against the tests/golden JSON fixtures, which oracle/gen_golden.js produced by running the
 * real reference (KORandi/bullet-js src/bullet-crt.js, src/bullet-query.js) under
 * Node on seeded inputs.
 *
 * Reference lines restated:
 *   compare (default 3-way)              src/bullet-crt.js:11-15
 *   compareVectorClocks, scalar form     src/bullet-crt.js:68-95
 *   mergeVectorClocks, scalar form       src/bullet-crt.js:103-114
 *   resolve decision table               src/bullet-crt.js:164-279
 *   caller's store rule (doUpdate)       src/bullet-crt.js:383, src/bullet.js:144-148
 *   sequential batch loop                src/bullet-network-sync.js:551-569
 *   index build / equals / range / count src/bullet-query.js:53-73, 186-210, 221-261, 293-313
 *   filter (full scan, predicate)        src/bullet-query.js:270-283
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FLAG_INCOMING   1u
#define ORC_FLAG_CURRENT    2u
#define ORC_FLAG_HISTORICAL 4u

#define ORC_INSERT_REFERENCE 0 /* stored clock of a first write is {localId: 2}: src/bullet-crt.js:172-185 + :33-60 */
#define ORC_INSERT_DELTA     1 /* stored clock is the incoming one (true LWW); not what the reference does */
#define ORC_VAL_DELETED INT64_MIN /* tombstone value (== BMX_VAL_DELETED): outside the value domain |val| <= 2^53-1 */

typedef struct {
  uint64_t id;
  uint32_t field;
  uint32_t stamp;   /* batch stamp of last_j */
  int64_t ts, val;
  uint32_t last_j;  /* index of the last applied delta of the current batch */
} orc_row;

typedef struct orc {
  orc_row* rows;      /* insertion order */
  uint64_t n, cap;
  uint32_t* slots;    /* open addressing: row index + 1, 0 = empty */
  uint64_t nslots;    /* power of two */
  uint32_t stamp;
} orc_t;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static uint64_t key_hash(uint64_t id, uint32_t field) { return mix64(id ^ ((uint64_t)field * 0x9E3779B97F4A7C15ULL)); }

static void rehash(orc_t* t, uint64_t nslots) {
  free(t->slots);
  t->slots = (uint32_t*)calloc(nslots, sizeof(uint32_t));
  t->nslots = nslots;
  for (uint64_t i = 0; i < t->n; i++) {
    uint64_t s = key_hash(t->rows[i].id, t->rows[i].field) & (nslots - 1);
    while (t->slots[s]) s = (s + 1) & (nslots - 1);
    t->slots[s] = (uint32_t)(i + 1);
  }
}

orc_t* orc_create(void) {
  orc_t* t = (orc_t*)calloc(1, sizeof(orc_t));
  t->cap = 1024;
  t->rows = (orc_row*)malloc(t->cap * sizeof(orc_row));
  rehash(t, 4096);
  return t;
}
void orc_destroy(orc_t* t) { if (t) { free(t->rows); free(t->slots); free(t); } }
uint64_t orc_size(const orc_t* t) { return t->n; }

static orc_row* find(orc_t* t, uint64_t id, uint32_t field) {
  uint64_t s = key_hash(id, field) & (t->nslots - 1);
  while (t->slots[s]) {
    orc_row* r = &t->rows[t->slots[s] - 1];
    if (r->id == id && r->field == field) return r;
    s = (s + 1) & (t->nslots - 1);
  }
  return NULL;
}
static orc_row* append(orc_t* t, uint64_t id, uint32_t field) {
  if (t->n == t->cap) { t->cap *= 2; t->rows = (orc_row*)realloc(t->rows, t->cap * sizeof(orc_row)); }
  if ((t->n + 1) * 2 > t->nslots) rehash(t, t->nslots * 2);
  orc_row* r = &t->rows[t->n];
  r->id = id; r->field = field; r->stamp = 0; r->last_j = 0; r->ts = 0; r->val = 0;
  uint64_t s = key_hash(id, field) & (t->nslots - 1);
  while (t->slots[s]) s = (s + 1) & (t->nslots - 1);
  t->slots[s] = (uint32_t)(++t->n);
  return r;
}

/* Direct preload of resident rows: the harness state is set without a merge. Also the restatement of bmx_put_rows: rows whose
 * outcome was decided elsewhere are stored as given; val == ORC_VAL_DELETED leaves a tombstone (the reference's setData(path, null) for a
 * deleted entry, src/bullet-network-sync.js:553-555: the key keeps its clock, _addToIndex skips the null value, src/bullet-query.js:83-85). */
void orc_load_rows(orc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val) {
  for (uint64_t i = 0; i < n; i++) {
    orc_row* r = find(t, id[i], field[i]);
    if (!r) r = append(t, id[i], field[i]);
    r->ts = ts[i]; r->val = val[i];
  }
}

/* default compare: ===→0, <→-1, else +1      src/bullet-crt.js:11-15 */
static int cmp3(int64_t a, int64_t b) { return a == b ? 0 : (a < b ? -1 : 1); }

/*
 * One delta through resolve() with clocks {w:a} (incoming) and {w:r->ts} (current).
 * Returns the decision flags; applies the caller's store rule.
 */
static unsigned resolve_scalar(orc_t* t, uint64_t id, uint32_t field, int64_t a, int64_t v, int insert_mode, orc_row** out_row) {
  orc_row* r = find(t, id, field);
  if (!r) {                                   /* "no current state"  :172-185 */
    r = append(t, id, field);
    r->ts = (insert_mode == ORC_INSERT_REFERENCE) ? 2 : a;   /* createVectorClock {id:1} then increment → 2 */
    r->val = v;
    *out_row = r;
    return ORC_FLAG_INCOMING;
  }
  *out_row = r;
  int c = cmp3(a, r->ts);                     /* compareVectorClocks, one component  :68-95 */
  /* mergedClock = max(a, cur) is stored in crt.vectorClocks even when incoming loses (:192-197);
     for one component that equals cur.ts whenever incoming does not win, so nothing to do. */
  if (c == 0) {                               /* identical clocks → value comparison  :200-233 */
    int vc = cmp3(v, r->val);
    if (vc == 0) return 0;                    /* identical clocks and values: all flags false */
    if (vc > 0) { r->val = v; return ORC_FLAG_INCOMING; }
    return ORC_FLAG_CURRENT;
  }
  if (c > 0) { r->ts = a; r->val = v; return ORC_FLAG_INCOMING; }   /* incoming dominates  :236-248 */
  return ORC_FLAG_CURRENT | ORC_FLAG_HISTORICAL;                     /* current dominates   :251-263 */
}

/*
 * Sequential batch merge, deltas applied in index order.
 * flags[j] (optional): decision flags of delta j.
 * winners (optional, capacity n): ascending indices of the final winner per changed key =
 *   the last applied delta of that key (== smallest index attaining the key's final lexmax).
 * Returns the number of winners.
 */
/* created (optional): created[j] = 1 iff delta j took the "no current state" branch (src/bullet-crt.js:172-185): its row stores the insert
 * rule's clock, not its own — what BMX_MERGE_MARK_CREATED reports for winners. */
uint64_t orc_merge_batch_marked(orc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                                const int64_t* val, int insert_mode, uint8_t* flags, uint32_t* winners, uint8_t* created) {
  t->stamp++;
  uint8_t* applied = (uint8_t*)calloc(n ? n : 1, 1);
  for (uint64_t j = 0; j < n; j++) {
    orc_row* r;
    const uint64_t rows_before = t->n;
    unsigned f = resolve_scalar(t, id[j], field[j], ts[j], val[j], insert_mode, &r);
    if (flags) flags[j] = (uint8_t)f;
    if (created) created[j] = t->n != rows_before;
    if (f & ORC_FLAG_INCOMING) {
      if (r->stamp == t->stamp) applied[r->last_j] = 0;
      r->stamp = t->stamp; r->last_j = (uint32_t)j; applied[j] = 1;
    }
  }
  uint64_t w = 0;
  for (uint64_t j = 0; j < n; j++) if (applied[j]) { if (winners) winners[w] = (uint32_t)j; w++; }
  free(applied);
  return w;
}
uint64_t orc_merge_batch(orc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                         const int64_t* val, int insert_mode, uint8_t* flags, uint32_t* winners) {
  return orc_merge_batch_marked(t, n, id, field, ts, val, insert_mode, flags, winners, NULL);
}

int orc_get_row(orc_t* t, uint64_t id, uint32_t field, int64_t* ts, int64_t* val) {
  orc_row* r = find(t, id, field);
  if (!r) return 0;
  *ts = r->ts; *val = r->val;
  return 1;
}

uint64_t orc_dump_rows(const orc_t* t, uint64_t cap, uint64_t* id, uint32_t* field, int64_t* ts, int64_t* val) {
  uint64_t m = 0;
  for (uint64_t i = 0; i < t->n; i++) {
    if (t->rows[i].val == ORC_VAL_DELETED) continue;          /* tombstones are not data */
    if (m < cap) { id[m] = t->rows[i].id; field[m] = t->rows[i].field; ts[m] = t->rows[i].ts; val[m] = t->rows[i].val; }
    m++;
  }
  return m;
}

static uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
/* order-independent digest of the row set; same formula as oracle/gen_golden.js rowDigest() */
uint64_t orc_row_digest(uint64_t id, uint32_t field, int64_t ts, int64_t val) {
  uint64_t h = splitmix64((uint64_t)val);
  h = splitmix64(h ^ (uint64_t)ts);
  h = splitmix64(h ^ (uint64_t)field);
  h = splitmix64(h ^ id);
  return h;
}
uint64_t orc_digest(const orc_t* t) {
  uint64_t d = 0;
  for (uint64_t i = 0; i < t->n; i++)
    if (t->rows[i].val != ORC_VAL_DELETED) d += orc_row_digest(t->rows[i].id, t->rows[i].field, t->rows[i].ts, t->rows[i].val);
  return d;
}

/*
 * Index scan in the FRESH-index state (SURVEY §8(a) "Scan parity target"): the set the reference's
 * range()/equals() return right after index() equals a ground-truth scan of the field's rows with
 * lo <= val <= hi (both ends inclusive; src/bullet-query.js:248-253). equals(c) = range(c,c); count = |equals|.
 * Ids are returned in row insertion order (the reference's order is by first-seen value: see
 * reference_order() in the host mirror). out_ids may be NULL to count only. Returns the match count.
 */
uint64_t orc_scan_range(const orc_t* t, uint32_t field, int64_t lo, int64_t hi, uint64_t* out_ids, uint64_t cap) {
  uint64_t m = 0;
  for (uint64_t i = 0; i < t->n; i++) {
    const orc_row* r = &t->rows[i];
    if (r->field != field || r->val == ORC_VAL_DELETED) continue;   /* _addToIndex skips null: src/bullet-query.js:83-85 */
    if (r->val >= lo && r->val <= hi) { if (out_ids && m < cap) out_ids[m] = r->id; m++; }
  }
  return m;
}

/*
 * Declarative filter: nodes (ids) having, for every term k, a row (id, fields[k]) with lo[k] <= val <= hi[k].
 * Restates filter(path, fn) (src/bullet-query.js:270-283) for fn = AND of range terms over fields of one node.
 * Driven by the first term's rows in insertion order.
 */
uint64_t orc_scan_filter_and(orc_t* t, uint32_t nterms, const uint32_t* fields, const int64_t* lo, const int64_t* hi,
                             uint64_t* out_ids, uint64_t cap) {
  uint64_t m = 0;
  if (nterms == 0) return 0;
  for (uint64_t i = 0; i < t->n; i++) {
    const orc_row* r = &t->rows[i];
    if (r->field != fields[0] || r->val == ORC_VAL_DELETED || r->val < lo[0] || r->val > hi[0]) continue;
    int ok = 1;
    for (uint32_t k = 1; k < nterms && ok; k++) {
      orc_row* q = find(t, r->id, fields[k]);
      ok = q && q->val != ORC_VAL_DELETED && q->val >= lo[k] && q->val <= hi[k];
    }
    if (ok) { if (out_ids && m < cap) out_ids[m] = r->id; m++; }
  }
  return m;
}

/* owner shard of a node id: must match bmx_owner_of() in the product (include/bmx.h) */
uint32_t orc_owner_of(uint64_t id, uint32_t nshards) {
  uint64_t h = mix64(id * 0xD6E8FEB86659FD93ULL + 0x2545F4914F6CDD1DULL);
  return (uint32_t)(((unsigned __int128)h * nshards) >> 64);
}

/* ---- all-cores variant of the scalar merge, for bench.py's extra CPU baseline line only ------------------------
 * T independent tables; table k owns the keys with orc_owner_of(id, T) == k. Every thread walks the whole batch in index order
 * and applies the deltas it owns, so each key still sees its deltas sequentially: same final state as one table. */
#include <pthread.h>
typedef struct { orc_t* t; uint32_t k, T; uint64_t n; const uint64_t* id; const uint32_t* field; const int64_t* ts; const int64_t* val;
                 int mode; int load; uint64_t applied; } orc_mt_job;
static void* orc_mt_run(void* p) {
  orc_mt_job* j = (orc_mt_job*)p;
  uint64_t applied = 0;
  for (uint64_t i = 0; i < j->n; i++) {
    if (orc_owner_of(j->id[i], j->T) != j->k) continue;
    if (j->load) {
      orc_row* r = find(j->t, j->id[i], j->field[i]);
      if (!r) r = append(j->t, j->id[i], j->field[i]);
      r->ts = j->ts[i]; r->val = j->val[i];
    } else {
      orc_row* r;
      if (resolve_scalar(j->t, j->id[i], j->field[i], j->ts[i], j->val[i], j->mode, &r) & ORC_FLAG_INCOMING) applied++;
    }
  }
  j->applied = applied;
  return NULL;
}
/* tables: T handles from orc_create(); load != 0 preloads rows instead of merging. Returns the number of applied deltas. */
uint64_t orc_mt_batch(orc_t** tables, uint32_t T, uint64_t n, const uint64_t* id, const uint32_t* field, const int64_t* ts,
                      const int64_t* val, int insert_mode, int load) {
  pthread_t* th = (pthread_t*)malloc(T * sizeof(pthread_t));
  orc_mt_job* jobs = (orc_mt_job*)malloc(T * sizeof(orc_mt_job));
  for (uint32_t k = 0; k < T; k++) {
    orc_mt_job jb = {tables[k], k, T, n, id, field, ts, val, insert_mode, load, 0};
    jobs[k] = jb;
    pthread_create(&th[k], NULL, orc_mt_run, &jobs[k]);
  }
  uint64_t total = 0;
  for (uint32_t k = 0; k < T; k++) { pthread_join(th[k], NULL); total += jobs[k].applied; }
  free(th); free(jobs);
  return total;
}

/* ======================================================================================================================
 * N4: fixed-K multi-writer vector clocks, integer values. Restates resolve() for general clocks
 * (src/bullet-crt.js:164-279) with compareVectorClocks :68-95 (missing component = 0), mergeVectorClocks :103-114
 * (component-wise max) and mergeValues :122-153 for non-objects (compare(in,cur) >= 0 ? in : cur).
 * A clock is K counters plus its KEY SET: which writers the JS object names and in which order (eight 4-bit writer indices, 0xF = end).
 * A missing key counts as 0 in the dominance test (:76-79); two clocks are "identical" only if JSON.stringify says so (:200-203): same keys,
 * same order, same counters; a merged clock is {...incoming} followed by the stored clock's other keys (:103-114). The plain entry points
 * take "dense" clocks (all K writers, in order); a first write stores the ONE-key clock {local: 2} (:172-185), also reported as `sparse`.
 * Pinned by the g6 vc fixtures (dense clocks) and g11_vc_keysets_*.json (subsets, permutations, {}) under tests/golden (real reference, 3 writers).
 * ====================================================================================================================== */
#define ORC_VC_MAXK 8
#define ORC_FLAG_CONCURRENT 8u

typedef struct {
  uint64_t id; uint32_t field; uint32_t stamp; int64_t val; uint32_t clock[ORC_VC_MAXK]; uint32_t last_j; uint32_t ks; uint8_t sparse;
} orc_vrow;
typedef struct orc_vc {
  orc_vrow* rows; uint64_t n, cap; uint32_t* slots; uint64_t nslots; uint32_t K, local, stamp;
} orc_vc_t;

static void vc_rehash(orc_vc_t* t, uint64_t nslots) {
  free(t->slots);
  t->slots = (uint32_t*)calloc(nslots, sizeof(uint32_t));
  t->nslots = nslots;
  for (uint64_t i = 0; i < t->n; i++) {
    uint64_t s = key_hash(t->rows[i].id, t->rows[i].field) & (nslots - 1);
    while (t->slots[s]) s = (s + 1) & (nslots - 1);
    t->slots[s] = (uint32_t)(i + 1);
  }
}
orc_vc_t* orc_vc_create(uint32_t K, uint32_t local) {
  if (K == 0 || K > ORC_VC_MAXK || local >= K) return NULL;
  orc_vc_t* t = (orc_vc_t*)calloc(1, sizeof(orc_vc_t));
  t->K = K; t->local = local; t->cap = 1024;
  t->rows = (orc_vrow*)malloc(t->cap * sizeof(orc_vrow));
  vc_rehash(t, 4096);
  return t;
}
void orc_vc_destroy(orc_vc_t* t) { if (t) { free(t->rows); free(t->slots); free(t); } }
uint64_t orc_vc_size(const orc_vc_t* t) { return t->n; }
static orc_vrow* vc_find(orc_vc_t* t, uint64_t id, uint32_t field) {
  uint64_t s = key_hash(id, field) & (t->nslots - 1);
  while (t->slots[s]) {
    orc_vrow* r = &t->rows[t->slots[s] - 1];
    if (r->id == id && r->field == field) return r;
    s = (s + 1) & (t->nslots - 1);
  }
  return NULL;
}
static orc_vrow* vc_append(orc_vc_t* t, uint64_t id, uint32_t field) {
  if (t->n == t->cap) { t->cap *= 2; t->rows = (orc_vrow*)realloc(t->rows, t->cap * sizeof(orc_vrow)); }
  if ((t->n + 1) * 2 > t->nslots) vc_rehash(t, t->nslots * 2);
  orc_vrow* r = &t->rows[t->n];
  memset(r, 0, sizeof(*r));
  r->id = id; r->field = field;
  uint64_t s = key_hash(id, field) & (t->nslots - 1);
  while (t->slots[s]) s = (s + 1) & (t->nslots - 1);
  t->slots[s] = (uint32_t)(++t->n);
  return r;
}
#define ORC_KS_NONE 0xFFFFFFFFu
static uint32_t ks_dense(uint32_t K) { uint32_t ks = ORC_KS_NONE; for (uint32_t k = 0; k < K; k++) ks = (ks & ~(0xFu << (4 * k))) | (k << (4 * k)); return ks; }
/* key order of mergeVectorClocks(in, cur): the spread of `in`, then every key of `cur` that is new (:103-114) */
static uint32_t ks_merge(uint32_t in, uint32_t cur) {
  uint32_t keys[2 * ORC_VC_MAXK], nk = 0;
  for (int i = 0; i < ORC_VC_MAXK; i++) { uint32_t w = (in >> (4 * i)) & 0xFu; if (w == 0xFu) break; keys[nk++] = w; }
  for (int i = 0; i < ORC_VC_MAXK; i++) {
    uint32_t w = (cur >> (4 * i)) & 0xFu; if (w == 0xFu) break;
    int seen = 0; for (uint32_t x = 0; x < nk; x++) if (keys[x] == w) seen = 1;
    if (!seen) keys[nk++] = w;
  }
  uint32_t out = ORC_KS_NONE;
  for (uint32_t x = 0; x < nk && x < ORC_VC_MAXK; x++) out = (out & ~(0xFu << (4 * x))) | (keys[x] << (4 * x));
  return out;
}
/* preload (harness sets state directly): keysets NULL = dense */
void orc_vc_load_rows_ks(orc_vc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const uint32_t* keysets, const int64_t* val) {
  for (uint64_t i = 0; i < n; i++) {
    orc_vrow* r = vc_find(t, id[i], field[i]);
    if (!r) r = vc_append(t, id[i], field[i]);
    for (uint32_t k = 0; k < t->K; k++) r->clock[k] = clocks[i * t->K + k];
    r->val = val[i]; r->sparse = 0; r->ks = keysets ? keysets[i] : ks_dense(t->K);
  }
}
void orc_vc_load_rows(orc_vc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const int64_t* val) {
  orc_vc_load_rows_ks(t, n, id, field, clocks, NULL, val);
}
static unsigned vc_resolve(orc_vc_t* t, uint64_t id, uint32_t field, const uint32_t* c, uint32_t cks, int64_t v, orc_vrow** out) {
  orc_vrow* r = vc_find(t, id, field);
  if (!r) {                                  /* "no current state": stored clock is {local: 2}, the incoming clock is dropped */
    r = vc_append(t, id, field);
    r->clock[t->local] = 2; r->sparse = 1; r->val = v; r->ks = 0xFFFFFFF0u | t->local;
    *out = r;
    return ORC_FLAG_INCOMING;
  }
  *out = r;
  int in_ahead = 0, cur_ahead = 0, equal = 1;
  for (uint32_t k = 0; k < t->K; k++) {
    if (c[k] > r->clock[k]) in_ahead = 1; else if (r->clock[k] > c[k]) cur_ahead = 1;
    if (c[k] != r->clock[k]) equal = 0;
  }
  const int cmp = (in_ahead && cur_ahead) ? 0 : (in_ahead ? 1 : (cur_ahead ? -1 : 0));
  const int json_equal = equal && cks == r->ks;                /* same keys, same order, same counters */
  if (cmp == 0 && json_equal) {                                /* identical clocks: value comparison  :200-233 */
    int vc = cmp3(v, r->val);
    if (vc == 0) return 0;
    if (vc > 0) { r->val = v; return ORC_FLAG_INCOMING; }
    return ORC_FLAG_CURRENT;
  }
  if (cmp < 0) return ORC_FLAG_CURRENT | ORC_FLAG_HISTORICAL;  /* :251-263 (the row keeps its clock) */
  for (uint32_t k = 0; k < t->K; k++) if (c[k] > r->clock[k]) r->clock[k] = c[k];   /* merged clock is stored with the update */
  r->ks = ks_merge(cks, r->ks);
  r->sparse = 0;
  if (cmp > 0) { r->val = v; return ORC_FLAG_INCOMING; }       /* :236-248 */
  if (cmp3(v, r->val) >= 0) r->val = v;                        /* concurrent: mergeValues on non-objects  :266-278, :133-135 */
  return ORC_FLAG_CONCURRENT;
}
/* sequential batch; updated (optional, capacity n) = ascending indices of the last delta per key that caused a store
 * (doUpdate = incoming || no current || concurrent: src/bullet-crt.js:383). Returns their number. */
uint64_t orc_vc_merge_batch_ks(orc_vc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const uint32_t* keysets,
                               const int64_t* val, uint8_t* flags, uint32_t* updated) {
  t->stamp++;
  uint8_t* mark = (uint8_t*)calloc(n ? n : 1, 1);
  for (uint64_t j = 0; j < n; j++) {
    orc_vrow* r;
    unsigned f = vc_resolve(t, id[j], field[j], clocks + j * t->K, keysets ? keysets[j] : ks_dense(t->K), val[j], &r);
    if (flags) flags[j] = (uint8_t)f;
    if (f & (ORC_FLAG_INCOMING | ORC_FLAG_CONCURRENT)) {
      if (r->stamp == t->stamp) mark[r->last_j] = 0;
      r->stamp = t->stamp; r->last_j = (uint32_t)j; mark[j] = 1;
    }
  }
  uint64_t w = 0;
  for (uint64_t j = 0; j < n; j++) if (mark[j]) { if (updated) updated[w] = (uint32_t)j; w++; }
  free(mark);
  return w;
}
uint64_t orc_vc_merge_batch(orc_vc_t* t, uint64_t n, const uint64_t* id, const uint32_t* field, const uint32_t* clocks, const int64_t* val,
                            uint8_t* flags, uint32_t* updated) {
  return orc_vc_merge_batch_ks(t, n, id, field, clocks, NULL, val, flags, updated);
}
int orc_vc_get_row_ks(orc_vc_t* t, uint64_t id, uint32_t field, uint32_t* clock_out, uint32_t* ks_out, int64_t* val, int* sparse) {
  orc_vrow* r = vc_find(t, id, field);
  if (!r) return 0;
  for (uint32_t k = 0; k < t->K; k++) clock_out[k] = r->clock[k];
  *val = r->val; if (sparse) *sparse = r->sparse; if (ks_out) *ks_out = r->ks;
  return 1;
}
int orc_vc_get_row(orc_vc_t* t, uint64_t id, uint32_t field, uint32_t* clock_out, int64_t* val, int* sparse) {
  orc_vrow* r = vc_find(t, id, field);
  if (!r) return 0;
  for (uint32_t k = 0; k < t->K; k++) clock_out[k] = r->clock[k];
  *val = r->val; if (sparse) *sparse = r->sparse;
  return 1;
}
uint64_t orc_vc_dump_rows(const orc_vc_t* t, uint64_t cap, uint64_t* id, uint32_t* field, uint32_t* clocks, int64_t* val) {
  uint64_t m = t->n < cap ? t->n : cap;
  for (uint64_t i = 0; i < m; i++) {
    id[i] = t->rows[i].id; field[i] = t->rows[i].field; val[i] = t->rows[i].val;
    for (uint32_t k = 0; k < t->K; k++) clocks[i * t->K + k] = t->rows[i].clock[k];
  }
  return t->n;
}
