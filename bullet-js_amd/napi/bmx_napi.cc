// bmx_napi.cc — thin N-API shim over the C ABI (include/bmx.h). No logic lives here: it converts typed
// arrays to pointers, calls libbmx.so and throws a JS Error carrying bmx_last_error() on failure
// (reference error policy: the hot path reports, never aborts — src/bullet.js:230-234).
// Built by bullet-js_amd/Makefile into bullet-js_amd/bmx.node and loaded by js/native.js.
#include <node_api.h>
#include <cmath>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "bmx.h"

namespace {

#define NAPI_OK(call)                                           \
  do {                                                          \
    if ((call) != napi_ok) {                                    \
      napi_throw_error(env, nullptr, "N-API call failed: " #call); \
      return nullptr;                                           \
    }                                                           \
  } while (0)

napi_value throw_bmx(napi_env env, bmx_ctx* ctx, int rc) {
  std::string msg = "bmx error " + std::to_string(rc) + ": " + bmx_last_error(ctx);
  napi_value code, err, m;
  napi_create_string_utf8(env, msg.c_str(), NAPI_AUTO_LENGTH, &m);
  napi_create_error(env, nullptr, m, &err);
  napi_create_int32(env, rc, &code);
  napi_set_named_property(env, err, "code", code);
  napi_throw(env, err);
  return nullptr;
}

// A context is not re-entrant and the ORDER of merges matters (which delta creates a row decides its stored clock), so every
// operation on a handle takes a ticket when it is issued on the JS thread and runs when its turn comes: asynchronous merges
// (libuv workers) and synchronous calls execute in exactly the order JS issued them.
struct Handle {
  bmx_ctx* ctx = nullptr;
  std::mutex mu; std::condition_variable cv;
  uint64_t next_ticket = 0, serving = 0;
  uint64_t take() { std::lock_guard<std::mutex> g(mu); return next_ticket++; }
};
struct Turn {   // RAII: wait for the ticket's turn, release it on scope exit
  Handle* h; std::unique_lock<std::mutex> lk;
  Turn(Handle* hh, uint64_t ticket) : h(hh), lk(hh->mu) { h->cv.wait(lk, [&] { return h->serving == ticket; }); }
  explicit Turn(Handle* hh) : h(hh), lk(hh->mu) { const uint64_t t = h->next_ticket++; h->cv.wait(lk, [&] { return h->serving == t; }); }
  ~Turn() { h->serving++; lk.unlock(); h->cv.notify_all(); }
};

void finalize_handle(napi_env, void* data, void*) {
  Handle* h = static_cast<Handle*>(data);
  if (h->ctx) bmx_destroy(h->ctx);
  delete h;
}

bool get_handle(napi_env env, napi_value v, Handle** out) {
  void* p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !static_cast<Handle*>(p)->ctx) {
    napi_throw_error(env, nullptr, "bmx: invalid or closed engine handle");
    return false;
  }
  *out = static_cast<Handle*>(p);
  return true;
}

// typed array -> (pointer, element count); checks the element type
bool get_ta(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
  napi_typedarray_type t; napi_value ab; size_t off;
  if (napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
    napi_throw_type_error(env, nullptr, "bmx: wrong typed-array type (id BigUint64Array, field Uint32Array, ts/val BigInt64Array)");
    return false;
  }
  return true;
}

napi_value make_ta(napi_env env, napi_typedarray_type t, size_t elem, size_t n, void** data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, n * elem, data, &ab) != napi_ok) return nullptr;
  if (napi_create_typedarray(env, t, n, ab, 0, &ta) != napi_ok) return nullptr;
  return ta;
}

// number | bigint -> int64 (saturating; +-Infinity allowed, as range() accepts them: src/bullet-query.js:248-253)
bool get_i64(napi_env env, napi_value v, int64_t* out) {
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_bigint) {
    bool lossless;
    return napi_get_value_bigint_int64(env, v, out, &lossless) == napi_ok;
  }
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok || isnan(d)) { napi_throw_type_error(env, nullptr, "bmx: expected a number or bigint"); return false; }
  if (d >= 9.2e18) *out = INT64_MAX; else if (d <= -9.2e18) *out = INT64_MIN; else *out = (int64_t)d;
  return true;
}

void set_num(napi_env env, napi_value obj, const char* k, double v) { napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, k, n); }

#define ARGS(N)                                              \
  size_t argc = N; napi_value argv[N];                       \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr)); \
  if (argc < N) { napi_throw_type_error(env, nullptr, "bmx: missing arguments"); return nullptr; }

// ownersOf(id: BigUint64Array, nshards) -> Uint8Array: bmx_owner_of for every id (the host-side routing of small batches and of the K-writer table)
napi_value OwnersOf(napi_env env, napi_callback_info info) {
  ARGS(2);
  void* p; size_t n;
  if (!get_ta(env, argv[0], napi_biguint64_array, &p, &n)) return nullptr;
  uint32_t ns; NAPI_OK(napi_get_value_uint32(env, argv[1], &ns));
  if (ns == 0 || ns > 255) { napi_throw_range_error(env, nullptr, "bmx: 1..255 shards"); return nullptr; }
  void* o; napi_value out = make_ta(env, napi_uint8_array, 1, n, &o);
  const uint64_t* id = (const uint64_t*)p; uint8_t* ow = (uint8_t*)o;
  for (size_t i = 0; i < n; i++) ow[i] = (uint8_t)bmx_owner_of(id[i], ns);
  return out;
}

napi_value AbiVersion(napi_env env, napi_callback_info) { napi_value v; napi_create_int32(env, bmx_abi_version(), &v); return v; }

napi_value Create(napi_env env, napi_callback_info info) {
  ARGS(2);
  int32_t device; double cap;
  NAPI_OK(napi_get_value_int32(env, argv[0], &device));
  NAPI_OK(napi_get_value_double(env, argv[1], &cap));
  bmx_ctx* ctx = nullptr;
  int rc = bmx_create(device, (uint64_t)cap, 0, &ctx);
  if (rc) return throw_bmx(env, nullptr, rc);
  Handle* h = new Handle();
  h->ctx = ctx;
  napi_value ext;
  NAPI_OK(napi_create_external(env, h, finalize_handle, nullptr, &ext));
  return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  void* p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    Handle* h = static_cast<Handle*>(p);
    Turn turn(h);                                   // after every operation issued before the close
    if (h->ctx) { bmx_destroy(h->ctx); h->ctx = nullptr; }
  }
  return nullptr;
}

bool get_cols(napi_env env, napi_value* a, const uint64_t** id, const uint32_t** field, const int64_t** ts, const int64_t** val, size_t* n) {
  void *p0, *p1, *p2, *p3; size_t n0, n1, n2, n3;
  if (!get_ta(env, a[0], napi_biguint64_array, &p0, &n0) || !get_ta(env, a[1], napi_uint32_array, &p1, &n1) ||
      !get_ta(env, a[2], napi_bigint64_array, &p2, &n2) || !get_ta(env, a[3], napi_bigint64_array, &p3, &n3)) return false;
  if (n0 != n1 || n0 != n2 || n0 != n3) { napi_throw_range_error(env, nullptr, "bmx: column lengths differ"); return false; }
  *id = (const uint64_t*)p0; *field = (const uint32_t*)p1; *ts = (const int64_t*)p2; *val = (const int64_t*)p3; *n = n0;
  return true;
}

// mergeBatch(h, id, field, ts, val, mode) -> {applied: Uint32Array, flags: Uint8Array, nApplied, nConflicts, nRows}
napi_value MergeBatch(napi_env env, napi_callback_info info) {
  ARGS(6);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  int32_t mode; NAPI_OK(napi_get_value_int32(env, argv[5], &mode));
  std::vector<uint32_t> applied(n ? n : 1);
  void* fl = nullptr;
  napi_value flags = make_ta(env, napi_uint8_array, 1, n, &fl);
  uint64_t na = 0; bmx_merge_stats st; memset(&st, 0, sizeof(st));
  Turn turn(h);
  int rc = bmx_merge_batch(h->ctx, n, id, field, ts, val, mode, BMX_MEM_HOST, applied.data(), &na, (uint8_t*)fl, &st);
  if (rc) return throw_bmx(env, h->ctx, rc);
  void* ap = nullptr;
  napi_value ta = make_ta(env, napi_uint32_array, 4, (size_t)na, &ap);
  if (na) memcpy(ap, applied.data(), (size_t)na * 4);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "applied", ta);
  napi_set_named_property(env, out, "flags", flags);
  set_num(env, out, "nApplied", (double)st.n_applied); set_num(env, out, "nConflicts", (double)st.n_conflicts); set_num(env, out, "nRows", (double)st.n_rows);
  return out;
}

// ---- asynchronous merge: the H2D copy, kernels and D2H copy run on a libuv worker thread; resolves to the same object
// as mergeBatch. The input typed arrays are referenced until completion (they must not be mutated meanwhile).
struct MergeJob {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref refs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the four columns and the engine handle itself: the engine outlives the job
  Handle* h = nullptr;
  uint64_t ticket = 0;
  const uint64_t* id = nullptr; const uint32_t* field = nullptr; const int64_t* ts = nullptr; const int64_t* val = nullptr;
  size_t n = 0; int mode = 0;
  std::vector<uint32_t> applied; std::vector<uint8_t> flags;
  uint64_t na = 0; bmx_merge_stats st; int rc = 0; std::string err;
};

void merge_execute(napi_env, void* data) {
  MergeJob* j = static_cast<MergeJob*>(data);
  Turn turn(j->h, j->ticket);                       // merges apply in the order JS issued them, whatever worker picks them up
  if (!j->h->ctx) { j->rc = BMX_ERR_INVALID; j->err = "engine closed"; return; }
  j->rc = bmx_merge_batch(j->h->ctx, j->n, j->id, j->field, j->ts, j->val, j->mode, BMX_MEM_HOST, j->applied.data(), &j->na, j->flags.data(), &j->st);
  if (j->rc) j->err = bmx_last_error(j->h->ctx);
}

void merge_complete(napi_env env, napi_status, void* data) {
  MergeJob* j = static_cast<MergeJob*>(data);
  for (auto& r : j->refs) if (r) napi_delete_reference(env, r);
  if (j->rc) {
    napi_value msg, err, code;
    std::string m = "bmx error " + std::to_string(j->rc) + ": " + j->err;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_create_int32(env, j->rc, &code);
    napi_set_named_property(env, err, "code", code);
    napi_reject_deferred(env, j->deferred, err);
  } else {
    void *ap = nullptr, *fl = nullptr;
    napi_value ta = make_ta(env, napi_uint32_array, 4, (size_t)j->na, &ap);
    if (j->na) memcpy(ap, j->applied.data(), (size_t)j->na * 4);
    napi_value fa = make_ta(env, napi_uint8_array, 1, j->n, &fl);
    if (j->n) memcpy(fl, j->flags.data(), j->n);
    napi_value out;
    napi_create_object(env, &out);
    napi_set_named_property(env, out, "applied", ta);
    napi_set_named_property(env, out, "flags", fa);
    set_num(env, out, "nApplied", (double)j->st.n_applied); set_num(env, out, "nConflicts", (double)j->st.n_conflicts); set_num(env, out, "nRows", (double)j->st.n_rows);
    napi_resolve_deferred(env, j->deferred, out);
  }
  napi_delete_async_work(env, j->work);
  delete j;
}

// mergeBatchAsync(h, id, field, ts, val, mode) -> Promise<{applied, flags, nApplied, nConflicts, nRows}>
napi_value MergeBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(6);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  MergeJob* j = new MergeJob();
  j->h = h;
  if (!get_cols(env, argv + 1, &j->id, &j->field, &j->ts, &j->val, &j->n)) { delete j; return nullptr; }
  int32_t mode; if (napi_get_value_int32(env, argv[5], &mode) != napi_ok) { delete j; napi_throw_type_error(env, nullptr, "bmx: bad mode"); return nullptr; }
  j->mode = mode;
  j->applied.resize(j->n ? j->n : 1); j->flags.resize(j->n ? j->n : 1);
  memset(&j->st, 0, sizeof(j->st));
  napi_value promise, name;
  auto drop = [&](const char* what) {
    for (auto& r : j->refs) if (r) napi_delete_reference(env, r);
    if (j->work) napi_delete_async_work(env, j->work);
    delete j;
    napi_throw_error(env, nullptr, what);
    return (napi_value) nullptr;
  };
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) return drop("bmx: could not create a promise");
  for (int k = 0; k < 4; k++) napi_create_reference(env, argv[1 + k], 1, &j->refs[k]);
  napi_create_reference(env, argv[0], 1, &j->refs[4]);
  if (napi_create_string_utf8(env, "bmx.mergeBatchAsync", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, nullptr, name, merge_execute, merge_complete, j, &j->work) != napi_ok) return drop("bmx: could not create the async work item");
  // the ticket is taken last: a ticket that never runs would block every later operation on this handle
  j->ticket = h->take();
  if (napi_queue_async_work(env, j->work) != napi_ok) {
    { Turn skip(h, j->ticket); }                    // give the turn back
    return drop("bmx: could not queue the async work item");
  }
  return promise;
}

// hostColumns(n) -> {id: BigUint64Array, field: Uint32Array, ts: BigInt64Array, val: BigInt64Array} of n rows over ONE page-locked allocation
// (bmx_host_alloc): batches built in it upload at the link's rate, with no pinning of fresh pages by the runtime. Freed when the buffer is collected.
void finalize_host_buffer(napi_env, void* data, void*) { (void)bmx_host_free(data); }
napi_value HostColumns(napi_env env, napi_callback_info info) {
  ARGS(1);
  double dn;
  if (napi_get_value_double(env, argv[0], &dn) != napi_ok || !(dn >= 1) || dn > 16777216) { napi_throw_range_error(env, nullptr, "bmx: hostColumns(n) wants 1 <= n <= 2^24"); return nullptr; }
  const size_t n = (size_t)dn;
  void* mem = nullptr;
  int rc = bmx_host_alloc(28ull * n, &mem);
  if (rc) return throw_bmx(env, nullptr, rc);
  napi_value ab, out, id, field, ts, val;
  if (napi_create_external_arraybuffer(env, mem, 28 * n, finalize_host_buffer, nullptr, &ab) != napi_ok) { (void)bmx_host_free(mem); napi_throw_error(env, nullptr, "bmx: external ArrayBuffer refused"); return nullptr; }
  NAPI_OK(napi_create_typedarray(env, napi_biguint64_array, n, ab, 0, &id));
  NAPI_OK(napi_create_typedarray(env, napi_bigint64_array, n, ab, 8 * n, &ts));
  NAPI_OK(napi_create_typedarray(env, napi_bigint64_array, n, ab, 16 * n, &val));
  NAPI_OK(napi_create_typedarray(env, napi_uint32_array, n, ab, 24 * n, &field));
  NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "id", id); napi_set_named_property(env, out, "field", field);
  napi_set_named_property(env, out, "ts", ts); napi_set_named_property(env, out, "val", val);
  return out;
}

napi_value Reserve(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  double cap; NAPI_OK(napi_get_value_double(env, argv[1], &cap));
  Turn turn(h);
  int rc = bmx_reserve(h->ctx, (uint64_t)cap);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}

napi_value LoadRows(napi_env env, napi_callback_info info) {
  ARGS(5);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  int rc = bmx_load_rows(h->ctx, n, id, field, ts, val, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}

// putRows(h, id, field, ts, val): rows decided on the host, stored as given; val == -2^63 (BMX_VAL_DELETED) leaves a tombstone
napi_value PutRows(napi_env env, napi_callback_info info) {
  ARGS(5);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  int rc = bmx_put_rows(h->ctx, n, id, field, ts, val, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}

// getRows(h, id, field) -> {ts, val, found}
napi_value GetRows(napi_env env, napi_callback_info info) {
  ARGS(3);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  void *p0, *p1; size_t n0, n1;
  if (!get_ta(env, argv[1], napi_biguint64_array, &p0, &n0) || !get_ta(env, argv[2], napi_uint32_array, &p1, &n1)) return nullptr;
  if (n0 != n1) { napi_throw_range_error(env, nullptr, "bmx: column lengths differ"); return nullptr; }
  void *ts, *val, *found;
  napi_value a = make_ta(env, napi_bigint64_array, 8, n0, &ts), b = make_ta(env, napi_bigint64_array, 8, n0, &val), c = make_ta(env, napi_uint8_array, 1, n0, &found);
  int rc = bmx_get_rows(h->ctx, n0, (const uint64_t*)p0, (const uint32_t*)p1, (int64_t*)ts, (int64_t*)val, (uint8_t*)found, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "ts", a); napi_set_named_property(env, out, "val", b); napi_set_named_property(env, out, "found", c);
  return out;
}

napi_value RowCount(napi_env env, napi_callback_info info) {
  ARGS(1);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint64_t n = 0; int rc = bmx_row_count(h->ctx, &n);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value v; napi_create_double(env, (double)n, &v); return v;
}

napi_value DumpRows(napi_env env, napi_callback_info info) {
  ARGS(1);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint64_t n = 0; int rc = bmx_row_count(h->ctx, &n);
  if (rc) return throw_bmx(env, h->ctx, rc);
  std::vector<uint64_t> id(n ? n : 1); std::vector<uint32_t> f(n ? n : 1); std::vector<int64_t> ts(n ? n : 1), val(n ? n : 1);
  uint64_t m = 0;
  rc = bmx_dump_rows(h->ctx, n, id.data(), f.data(), ts.data(), val.data(), &m, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  if (m > n) m = n;                    // rows in use >= rows dumped: tombstones keep their slot and are not data
  void *pi, *pf, *pt, *pv;
  napi_value a = make_ta(env, napi_biguint64_array, 8, m, &pi), b = make_ta(env, napi_uint32_array, 4, m, &pf),
             c = make_ta(env, napi_bigint64_array, 8, m, &pt), d = make_ta(env, napi_bigint64_array, 8, m, &pv);
  if (m) { memcpy(pi, id.data(), m * 8); memcpy(pf, f.data(), m * 4); memcpy(pt, ts.data(), m * 8); memcpy(pv, val.data(), m * 8); }
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "id", a); napi_set_named_property(env, out, "field", b);
  napi_set_named_property(env, out, "ts", c); napi_set_named_property(env, out, "val", d);
  return out;
}

napi_value IndexBuild(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int rc = bmx_index_build(h->ctx, f);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}
/* indexSetOrdered(handle, field, afterQueries): value-ordered view of the index (bmx_index_set_ordered); 0 = off. -> {afterQueries, valid, sorts} */
napi_value IndexSetOrdered(napi_env env, napi_callback_info info) {
  ARGS(3);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f, n; NAPI_OK(napi_get_value_uint32(env, argv[1], &f)); NAPI_OK(napi_get_value_uint32(env, argv[2], &n));
  int rc = bmx_index_set_ordered(h->ctx, f, n);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}
napi_value IndexOrderedInfo(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  uint32_t after = 0; int valid = 0; uint64_t sorts = 0;
  int rc = bmx_index_ordered_info(h->ctx, f, &after, &valid, &sorts);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "afterQueries", after); set_num(env, out, "valid", valid); set_num(env, out, "sorts", (double)sorts);
  // round 5 (ABI 4): what kept the view current — patches from the change log instead of sorts (bmx_index_ordered_stats)
  uint64_t s2 = 0, patches = 0, keys = 0, rewrites = 0, pending = 0; double sort_us = 0, patch_us = 0;
  if (bmx_index_ordered_stats(h->ctx, f, &s2, &patches, &keys, &sort_us, &patch_us, &rewrites, &pending) == BMX_OK) {
    set_num(env, out, "patches", (double)patches); set_num(env, out, "keysPatched", (double)keys); set_num(env, out, "lastSortUs", sort_us); set_num(env, out, "lastPatchUs", patch_us);
    set_num(env, out, "rewrites", (double)rewrites); set_num(env, out, "pendingKeys", (double)pending);
  }
  return out;
}
napi_value IndexDrop(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int rc = bmx_index_drop(h->ctx, f);
  if (rc && rc != BMX_ERR_NO_INDEX) return throw_bmx(env, h->ctx, rc);
  return nullptr;
}
napi_value IndexSize(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  uint64_t n = 0; int rc = bmx_index_size(h->ctx, f, &n);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value v; napi_create_double(env, (double)n, &v); return v;
}
// indexRefreshCounts(h) -> {fullBuilds, incremental}: how often the indexes were rebuilt from the table / brought up to date from the change log
napi_value IndexRefreshCounts(napi_env env, napi_callback_info info) {
  ARGS(1);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);
  uint64_t a = 0, b = 0; int rc = bmx_index_refresh_counts(h->ctx, &a, &b);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "fullBuilds", (double)a); set_num(env, out, "incremental", (double)b);
  return out;
}

// scanRange(h, field, lo, hi) -> BigUint64Array of node ids
napi_value ScanRange(napi_env env, napi_callback_info info) {
  ARGS(4);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  uint64_t m = 0; int rc = bmx_scan_count(h->ctx, f, lo, hi, &m, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, m, &out);
  if (m) { uint64_t m2 = 0; rc = bmx_scan_range(h->ctx, f, lo, hi, (uint64_t*)out, m, &m2, BMX_MEM_HOST); if (rc) return throw_bmx(env, h->ctx, rc); }
  return ta;
}
napi_value ScanCount(napi_env env, napi_callback_info info) {
  ARGS(4);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  uint64_t m = 0; int rc = bmx_scan_count(h->ctx, f, lo, hi, &m, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value v; napi_create_double(env, (double)m, &v); return v;
}
// scanRangePos(h, field, lo, hi) -> Uint32Array of index positions (ascending): no id gather on the device, no id -> path lookup on the host
napi_value ScanRangePos(napi_env env, napi_callback_info info) {
  ARGS(4);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  uint64_t m = 0; int rc = bmx_scan_count(h->ctx, f, lo, hi, &m, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  void* out; napi_value ta = make_ta(env, napi_uint32_array, 4, m, &out);
  if (m) { uint64_t m2 = 0; rc = bmx_scan_range_pos(h->ctx, f, lo, hi, (uint32_t*)out, m, &m2, BMX_MEM_HOST); if (rc) return throw_bmx(env, h->ctx, rc); }
  return ta;
}
// indexIds(h, field, first, count) -> BigUint64Array: node ids of index positions [first, first + count)
napi_value IndexIds(napi_env env, napi_callback_info info) {
  ARGS(4);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  double first, count; NAPI_OK(napi_get_value_double(env, argv[2], &first)); NAPI_OK(napi_get_value_double(env, argv[3], &count));
  // validated BEFORE anything is allocated: a negative, fractional or NaN count cast to size_t is undefined behaviour or a huge allocation
  uint64_t size = 0; int src = bmx_index_size(h->ctx, f, &size);
  if (src) return throw_bmx(env, h->ctx, src);
  if (!(first >= 0 && count >= 0) || first != std::floor(first) || count != std::floor(count) || first > (double)size || count > (double)size - first) {
    napi_throw_range_error(env, nullptr, "bmx: indexIds(first, count) reaches outside the index");
    return nullptr;
  }
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, (size_t)count, &out);
  int rc = bmx_index_ids(h->ctx, f, (uint64_t)first, (uint64_t)count, (uint64_t*)out, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  return ta;
}
// scanFilter(h, [[field, lo, hi], ...]) -> BigUint64Array
napi_value ScanFilter(napi_env env, napi_callback_info info) {
  ARGS(2);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  uint32_t nt = 0; NAPI_OK(napi_get_array_length(env, argv[1], &nt));
  if (nt == 0 || nt > 8) { napi_throw_range_error(env, nullptr, "bmx: filter needs 1..8 terms"); return nullptr; }
  bmx_term terms[8];
  for (uint32_t k = 0; k < nt; k++) {
    napi_value t, e0, e1, e2;
    NAPI_OK(napi_get_element(env, argv[1], k, &t));
    NAPI_OK(napi_get_element(env, t, 0, &e0)); NAPI_OK(napi_get_element(env, t, 1, &e1)); NAPI_OK(napi_get_element(env, t, 2, &e2));
    NAPI_OK(napi_get_value_uint32(env, e0, &terms[k].field));
    terms[k].reserved = 0;
    if (!get_i64(env, e1, &terms[k].lo) || !get_i64(env, e2, &terms[k].hi)) return nullptr;
  }
  uint64_t cap = 0; int rc = bmx_index_size(h->ctx, terms[0].field, &cap);
  if (rc) return throw_bmx(env, h->ctx, rc);
  std::vector<uint64_t> tmp(cap ? cap : 1);
  uint64_t m = 0; rc = bmx_scan_filter(h->ctx, nt, terms, tmp.data(), cap, &m, BMX_MEM_HOST);
  if (rc) return throw_bmx(env, h->ctx, rc);
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, m, &out);
  if (m) memcpy(out, tmp.data(), m * 8);
  return ta;
}

napi_value Info(napi_env env, napi_callback_info info) {
  ARGS(1);
  Handle* h; if (!get_handle(env, argv[0], &h)) return nullptr;
  Turn turn(h);   // runs in issue order with the asynchronous merges
  bmx_info i; int rc = bmx_get_info(h->ctx, &i);
  if (rc) return throw_bmx(env, h->ctx, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  set_num(env, out, "capacityRows", (double)i.capacity_rows); set_num(env, out, "nSlots", (double)i.n_slots);
  set_num(env, out, "tableBytes", (double)i.table_bytes); set_num(env, out, "nRows", (double)i.n_rows);
  set_num(env, out, "device", i.device); set_num(env, out, "abiVersion", i.abi_version); set_num(env, out, "nIndexes", i.n_indexes);
  return out;
}

// ---- N4: vector-clock table (bmx_vc_*) -----------------------------------------------------------------------------
// operations on a table run in the order JS issued them, on whatever thread (the asynchronous merge runs on a libuv worker): tickets, as for the scalar engine
struct VcHandle {
  bmx_vc* t; uint32_t K;
  std::mutex mu; std::condition_variable cv;
  uint64_t next_ticket = 0, serving = 0;
  uint64_t take() { std::lock_guard<std::mutex> g(mu); return next_ticket++; }
};
struct VcTurn {
  VcHandle* h; std::unique_lock<std::mutex> lk;
  VcTurn(VcHandle* hh, uint64_t ticket) : h(hh), lk(hh->mu) { h->cv.wait(lk, [&] { return h->serving == ticket; }); }
  explicit VcTurn(VcHandle* hh) : h(hh), lk(hh->mu) { const uint64_t t = h->next_ticket++; h->cv.wait(lk, [&] { return h->serving == t; }); }
  ~VcTurn() { h->serving++; lk.unlock(); h->cv.notify_all(); }
};

void finalize_vc(napi_env, void* data, void*) {
  VcHandle* h = static_cast<VcHandle*>(data);
  if (h->t) bmx_vc_destroy(h->t);
  delete h;
}

bool get_vc(napi_env env, napi_value v, VcHandle** out) {
  void* p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !static_cast<VcHandle*>(p)->t) {
    napi_throw_error(env, nullptr, "bmx: invalid or closed vector-clock table handle");
    return false;
  }
  *out = static_cast<VcHandle*>(p);
  return true;
}

napi_value throw_vc(napi_env env, bmx_vc* t, int rc) {
  std::string msg = "bmx error " + std::to_string(rc) + ": " + bmx_vc_last_error(t);
  napi_value code, err, m;
  napi_create_string_utf8(env, msg.c_str(), NAPI_AUTO_LENGTH, &m);
  napi_create_error(env, nullptr, m, &err);
  napi_create_int32(env, rc, &code);
  napi_set_named_property(env, err, "code", code);
  napi_throw(env, err);
  return nullptr;
}

// vcCreate(device, capacityRows, kWriters, localWriter) -> handle
napi_value VcCreate(napi_env env, napi_callback_info info) {
  ARGS(4);
  int32_t device; double cap; uint32_t K, local;
  NAPI_OK(napi_get_value_int32(env, argv[0], &device));
  NAPI_OK(napi_get_value_double(env, argv[1], &cap));
  NAPI_OK(napi_get_value_uint32(env, argv[2], &K));
  NAPI_OK(napi_get_value_uint32(env, argv[3], &local));
  bmx_vc* t = nullptr;
  int rc = bmx_vc_create(device, (uint64_t)cap, K, local, &t);
  if (rc) return throw_vc(env, nullptr, rc);
  VcHandle* h = new VcHandle();
  h->t = t; h->K = K;
  napi_value ext;
  NAPI_OK(napi_create_external(env, h, finalize_vc, nullptr, &ext));
  return ext;
}

napi_value VcDestroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  void* p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    VcHandle* h = static_cast<VcHandle*>(p);
    VcTurn turn(h);
    if (h->t) { bmx_vc_destroy(h->t); h->t = nullptr; }
  }
  return nullptr;
}

// (id BigUint64Array, field Uint32Array, clocks Uint32Array[n*K], val BigInt64Array)
bool get_vc_cols(napi_env env, napi_value* a, uint32_t K, const uint64_t** id, const uint32_t** field, const uint32_t** clocks, const int64_t** val, size_t* n) {
  void *p0, *p1, *p2, *p3; size_t n0, n1, n2, n3;
  if (!get_ta(env, a[0], napi_biguint64_array, &p0, &n0) || !get_ta(env, a[1], napi_uint32_array, &p1, &n1) ||
      !get_ta(env, a[2], napi_uint32_array, &p2, &n2) || !get_ta(env, a[3], napi_bigint64_array, &p3, &n3)) return false;
  if (n0 != n1 || n0 != n3 || n2 != n0 * K) { napi_throw_range_error(env, nullptr, "bmx: column lengths differ (clocks must hold n*K counters)"); return false; }
  *id = (const uint64_t*)p0; *field = (const uint32_t*)p1; *clocks = (const uint32_t*)p2; *val = (const int64_t*)p3; *n = n0;
  return true;
}

// optional trailing argument: keysets Uint32Array[n] (which writers each clock names, in which order: include/bmx.h); absent / undefined = all K, in order
bool get_keysets(napi_env env, size_t argc, napi_value* argv, size_t at, size_t n, const uint32_t** ks) {
  *ks = nullptr;
  if (argc <= at) return true;
  napi_valuetype t; napi_typeof(env, argv[at], &t);
  if (t == napi_undefined || t == napi_null) return true;
  void* p; size_t m;
  if (!get_ta(env, argv[at], napi_uint32_array, &p, &m)) return false;
  if (m != n) { napi_throw_range_error(env, nullptr, "bmx: keysets must hold one word per row"); return false; }
  *ks = (const uint32_t*)p;
  return true;
}
#define ARGS_OPT(MIN, MAX)                                   \
  size_t argc = MAX; napi_value argv[MAX];                   \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr)); \
  if (argc < MIN) { napi_throw_type_error(env, nullptr, "bmx: missing arguments"); return nullptr; }

// vcLoadRows(h, id, field, clocks, val[, keysets])
napi_value VcLoadRows(napi_env env, napi_callback_info info) {
  ARGS_OPT(5, 6);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t *field, *clocks, *ks; const int64_t* val; size_t n;
  if (!get_vc_cols(env, argv + 1, h->K, &id, &field, &clocks, &val, &n) || !get_keysets(env, argc, argv, 5, n, &ks)) return nullptr;
  VcTurn turn(h);
  int rc = bmx_vc_load_rows_ks(h->t, n, id, field, clocks, ks, val);
  if (rc) return throw_vc(env, h->t, rc);
  return nullptr;
}

// vcMergeBatch(h, id, field, clocks, val[, keysets]) -> {updated: Uint32Array, flags: Uint8Array, nRows}
napi_value VcMergeBatch(napi_env env, napi_callback_info info) {
  ARGS_OPT(5, 6);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t *field, *clocks, *ks; const int64_t* val; size_t n;
  if (!get_vc_cols(env, argv + 1, h->K, &id, &field, &clocks, &val, &n) || !get_keysets(env, argc, argv, 5, n, &ks)) return nullptr;
  std::vector<uint32_t> upd(n ? n : 1);
  void* fl = nullptr;
  napi_value flags = make_ta(env, napi_uint8_array, 1, n, &fl);
  uint64_t nu = 0, rows = 0;
  VcTurn turn(h);
  int rc = bmx_vc_merge_batch_ks(h->t, n, id, field, clocks, ks, val, upd.data(), &nu, (uint8_t*)fl);
  if (rc) return throw_vc(env, h->t, rc);
  bmx_vc_row_count(h->t, &rows);
  void* up = nullptr;
  napi_value updated = make_ta(env, napi_uint32_array, 4, nu, &up);
  if (nu) memcpy(up, upd.data(), nu * 4);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "updated", updated);
  napi_set_named_property(env, out, "flags", flags);
  set_num(env, out, "nRows", (double)rows);
  return out;
}

// vcMergeBatchAsync(h, id, field, clocks, val[, keysets]) -> Promise<{updated, flags, nRows, rows: {clocks, keysets} of the updated rows}>: the upload,
// the kernels and the read-back of the updated rows' clocks run on a libuv worker thread, in issue order with every other operation on the table
// (reference seam: the sync loop src/bullet-network-sync.js:551-569 under general vector clocks, src/bullet-crt.js:68-153). The rows' clocks come back
// with the merge because a later merge — already in flight when this one is applied — would have moved them.
struct VcJob {
  napi_async_work work = nullptr; napi_deferred deferred = nullptr;
  napi_ref refs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  VcHandle* h = nullptr; uint64_t ticket = 0;
  const uint64_t* id = nullptr; const uint32_t *field = nullptr, *clocks = nullptr, *ks = nullptr; const int64_t* val = nullptr; size_t n = 0;
  std::vector<uint32_t> upd, rclocks, rks; std::vector<uint8_t> flags, rstate; std::vector<int64_t> rval;
  uint64_t nu = 0, rows = 0; int rc = 0; std::string err;
};
void vc_execute(napi_env, void* data) {
  VcJob* j = static_cast<VcJob*>(data);
  VcTurn turn(j->h, j->ticket);
  if (!j->h->t) { j->rc = BMX_ERR_INVALID; j->err = "table closed"; return; }
  j->rc = bmx_vc_merge_batch_ks(j->h->t, j->n, j->id, j->field, j->clocks, j->ks, j->val, j->upd.data(), &j->nu, j->flags.data());
  if (j->rc) { j->err = bmx_vc_last_error(j->h->t); return; }
  bmx_vc_row_count(j->h->t, &j->rows);
  if (j->nu) {
    std::vector<uint64_t> ids(j->nu); std::vector<uint32_t> fields(j->nu);
    for (uint64_t k = 0; k < j->nu; k++) { ids[k] = j->id[j->upd[k]]; fields[k] = j->field[j->upd[k]]; }
    j->rclocks.resize(j->nu * j->h->K); j->rks.resize(j->nu); j->rval.resize(j->nu); j->rstate.resize(j->nu);
    j->rc = bmx_vc_get_rows_ks(j->h->t, j->nu, ids.data(), fields.data(), j->rclocks.data(), j->rks.data(), j->rval.data(), j->rstate.data());
    if (j->rc) j->err = bmx_vc_last_error(j->h->t);
  }
}
void vc_complete(napi_env env, napi_status, void* data) {
  VcJob* j = static_cast<VcJob*>(data);
  for (auto& r : j->refs) if (r) napi_delete_reference(env, r);
  if (j->rc) {
    napi_value msg, err, code;
    std::string m = "bmx error " + std::to_string(j->rc) + ": " + j->err;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &err);
    napi_create_int32(env, j->rc, &code);
    napi_set_named_property(env, err, "code", code);
    napi_reject_deferred(env, j->deferred, err);
  } else {
    void *up = nullptr, *fl = nullptr, *c = nullptr, *ks = nullptr;
    napi_value updated = make_ta(env, napi_uint32_array, 4, (size_t)j->nu, &up);
    if (j->nu) memcpy(up, j->upd.data(), (size_t)j->nu * 4);
    napi_value flags = make_ta(env, napi_uint8_array, 1, j->n, &fl);
    if (j->n) memcpy(fl, j->flags.data(), j->n);
    napi_value clocks = make_ta(env, napi_uint32_array, 4, (size_t)j->nu * j->h->K, &c);
    napi_value keysets = make_ta(env, napi_uint32_array, 4, (size_t)j->nu, &ks);
    if (j->nu) { memcpy(c, j->rclocks.data(), (size_t)j->nu * j->h->K * 4); memcpy(ks, j->rks.data(), (size_t)j->nu * 4); }
    napi_value out, rows;
    napi_create_object(env, &out); napi_create_object(env, &rows);
    napi_set_named_property(env, rows, "clocks", clocks); napi_set_named_property(env, rows, "keysets", keysets);
    napi_set_named_property(env, out, "updated", updated); napi_set_named_property(env, out, "flags", flags); napi_set_named_property(env, out, "rows", rows);
    set_num(env, out, "nRows", (double)j->rows);
    napi_resolve_deferred(env, j->deferred, out);
  }
  napi_delete_async_work(env, j->work);
  delete j;
}
napi_value VcMergeBatchAsync(napi_env env, napi_callback_info info) {
  ARGS_OPT(5, 6);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  VcJob* j = new VcJob();
  j->h = h;
  if (!get_vc_cols(env, argv + 1, h->K, &j->id, &j->field, &j->clocks, &j->val, &j->n) || !get_keysets(env, argc, argv, 5, j->n, &j->ks)) { delete j; return nullptr; }
  j->upd.resize(j->n ? j->n : 1); j->flags.resize(j->n ? j->n : 1);
  napi_value promise, name;
  auto drop = [&](const char* what) {
    for (auto& r : j->refs) if (r) napi_delete_reference(env, r);
    if (j->work) napi_delete_async_work(env, j->work);
    delete j;
    napi_throw_error(env, nullptr, what);
    return (napi_value) nullptr;
  };
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) return drop("bmx: could not create a promise");
  for (size_t k = 0; k < argc && k < 6; k++) napi_create_reference(env, argv[k], 1, &j->refs[k]);     // the table handle and the columns outlive the job
  if (napi_create_string_utf8(env, "bmx.vcMergeBatchAsync", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, nullptr, name, vc_execute, vc_complete, j, &j->work) != napi_ok) return drop("bmx: could not create the async work item");
  j->ticket = h->take();                            // taken last: a ticket that never runs would block every later operation on this table
  if (napi_queue_async_work(env, j->work) != napi_ok) {
    { VcTurn skip(h, j->ticket); }
    return drop("bmx: could not queue the async work item");
  }
  return promise;
}

// vcGetRows(h, id, field) -> {clocks: Uint32Array[n*K], val: BigInt64Array, state: Uint8Array, keysets: Uint32Array}
napi_value VcGetRows(napi_env env, napi_callback_info info) {
  ARGS(3);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  void *p0, *p1; size_t n0, n1;
  if (!get_ta(env, argv[1], napi_biguint64_array, &p0, &n0) || !get_ta(env, argv[2], napi_uint32_array, &p1, &n1)) return nullptr;
  if (n0 != n1) { napi_throw_range_error(env, nullptr, "bmx: column lengths differ"); return nullptr; }
  void *c, *v, *st, *ks;
  napi_value clocks = make_ta(env, napi_uint32_array, 4, n0 * h->K, &c);
  napi_value val = make_ta(env, napi_bigint64_array, 8, n0, &v);
  napi_value state = make_ta(env, napi_uint8_array, 1, n0, &st);
  napi_value keysets = make_ta(env, napi_uint32_array, 4, n0, &ks);
  VcTurn turn(h);
  int rc = bmx_vc_get_rows_ks(h->t, n0, (const uint64_t*)p0, (const uint32_t*)p1, (uint32_t*)c, (uint32_t*)ks, (int64_t*)v, (uint8_t*)st);
  if (rc) return throw_vc(env, h->t, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "keysets", keysets);
  napi_set_named_property(env, out, "clocks", clocks);
  napi_set_named_property(env, out, "val", val);
  napi_set_named_property(env, out, "state", state);
  return out;
}

// vcScanRange(h, field, lo, hi) -> BigUint64Array of node ids (rows of `field` with lo <= val <= hi)
napi_value VcScanRange(napi_env env, napi_callback_info info) {
  ARGS(4);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  VcTurn turn(h);
  uint64_t m = 0; int rc = bmx_vc_scan_range(h->t, f, lo, hi, nullptr, 0, &m);
  if (rc) return throw_vc(env, h->t, rc);
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, m, &out);
  if (m) { uint64_t m2 = 0; rc = bmx_vc_scan_range(h->t, f, lo, hi, (uint64_t*)out, m, &m2); if (rc) return throw_vc(env, h->t, rc); }
  return ta;
}
napi_value VcRowCount(napi_env env, napi_callback_info info) {
  ARGS(1);
  VcHandle* h; if (!get_vc(env, argv[0], &h)) return nullptr;
  uint64_t n = 0;
  VcTurn turn(h);
  int rc = bmx_vc_row_count(h->t, &n);
  if (rc) return throw_vc(env, h->t, rc);
  napi_value v; napi_create_double(env, (double)n, &v);
  return v;
}

// ---- N shards in one process (bmx_comm_*): one JS object owns all GPUs of the node ------------------------------------
struct CommHandle { bmx_comm* c; std::mutex mu; };

void finalize_comm(napi_env, void* data, void*) {
  CommHandle* h = static_cast<CommHandle*>(data);
  if (h->c) bmx_comm_destroy(h->c);
  delete h;
}
bool get_comm(napi_env env, napi_value v, CommHandle** out) {
  void* p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !static_cast<CommHandle*>(p)->c) {
    napi_throw_error(env, nullptr, "bmx: invalid or closed communicator handle");
    return false;
  }
  *out = static_cast<CommHandle*>(p);
  return true;
}
napi_value throw_comm(napi_env env, bmx_comm* c, int rc) {
  std::string msg = "bmx error " + std::to_string(rc) + ": " + bmx_comm_last_error(c);
  napi_value code, err, m;
  napi_create_string_utf8(env, msg.c_str(), NAPI_AUTO_LENGTH, &m);
  napi_create_error(env, nullptr, m, &err);
  napi_create_int32(env, rc, &code);
  napi_set_named_property(env, err, "code", code);
  napi_throw(env, err);
  return nullptr;
}

// commCreate([device, device, ...], capacityRowsPerShard) -> handle   (a device may be listed several times: logical shards)
napi_value CommCreate(napi_env env, napi_callback_info info) {
  ARGS(2);
  uint32_t n = 0; NAPI_OK(napi_get_array_length(env, argv[0], &n));
  if (n == 0 || n > 16) { napi_throw_range_error(env, nullptr, "bmx: a communicator has 1..16 shards"); return nullptr; }
  std::vector<int> devs(n);
  for (uint32_t i = 0; i < n; i++) { napi_value e; NAPI_OK(napi_get_element(env, argv[0], i, &e)); int32_t d; NAPI_OK(napi_get_value_int32(env, e, &d)); devs[i] = d; }
  double cap; NAPI_OK(napi_get_value_double(env, argv[1], &cap));
  bmx_comm* c = nullptr;
  int rc = bmx_comm_create(n, devs.data(), (uint64_t)cap, 0, &c);
  if (rc) return throw_comm(env, nullptr, rc);
  CommHandle* h = new CommHandle(); h->c = c;
  napi_value ext;
  NAPI_OK(napi_create_external(env, h, finalize_comm, nullptr, &ext));
  return ext;
}
napi_value CommDestroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  void* p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    CommHandle* h = static_cast<CommHandle*>(p);
    std::lock_guard<std::mutex> g(h->mu);
    if (h->c) { bmx_comm_destroy(h->c); h->c = nullptr; }
  }
  return nullptr;
}
// commMergeBatch(h, id, field, ts, val, mode) -> {applied: Uint32Array (indices into this batch), nApplied, nConflicts, nRows}
napi_value CommMergeBatch(napi_env env, napi_callback_info info) {
  ARGS(6);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  int32_t mode; NAPI_OK(napi_get_value_int32(env, argv[5], &mode));
  std::vector<uint32_t> applied(n ? n : 1);
  uint64_t na = 0; bmx_merge_stats st; memset(&st, 0, sizeof(st));
  std::lock_guard<std::mutex> g(h->mu);
  int rc = bmx_comm_merge(h->c, n, id, field, ts, val, mode, applied.data(), &na, &st);
  if (rc) return throw_comm(env, h->c, rc);
  void* ap = nullptr;
  napi_value ta = make_ta(env, napi_uint32_array, 4, (size_t)na, &ap);
  if (na) memcpy(ap, applied.data(), (size_t)na * 4);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "applied", ta);
  set_num(env, out, "nApplied", (double)st.n_applied); set_num(env, out, "nConflicts", (double)st.n_conflicts); set_num(env, out, "nRows", (double)st.n_rows);
  return out;
}
napi_value CommLoadRows(napi_env env, napi_callback_info info) {
  ARGS(5);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  int rc = bmx_comm_load_rows(h->c, n, id, field, ts, val);
  if (rc) return throw_comm(env, h->c, rc);
  return nullptr;
}
napi_value CommPutRows(napi_env env, napi_callback_info info) {
  ARGS(5);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  const uint64_t* id; const uint32_t* field; const int64_t *ts, *val; size_t n;
  if (!get_cols(env, argv + 1, &id, &field, &ts, &val, &n)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  int rc = bmx_comm_put_rows(h->c, n, id, field, ts, val);
  if (rc) return throw_comm(env, h->c, rc);
  return nullptr;
}
napi_value CommGetRows(napi_env env, napi_callback_info info) {
  ARGS(3);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  void *p0, *p1; size_t n0, n1;
  if (!get_ta(env, argv[1], napi_biguint64_array, &p0, &n0) || !get_ta(env, argv[2], napi_uint32_array, &p1, &n1)) return nullptr;
  if (n0 != n1) { napi_throw_range_error(env, nullptr, "bmx: column lengths differ"); return nullptr; }
  void *ts, *val, *found;
  napi_value a = make_ta(env, napi_bigint64_array, 8, n0, &ts), b = make_ta(env, napi_bigint64_array, 8, n0, &val), c = make_ta(env, napi_uint8_array, 1, n0, &found);
  std::lock_guard<std::mutex> g(h->mu);
  int rc = bmx_comm_get_rows(h->c, n0, (const uint64_t*)p0, (const uint32_t*)p1, (int64_t*)ts, (int64_t*)val, (uint8_t*)found);
  if (rc) return throw_comm(env, h->c, rc);
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "ts", a); napi_set_named_property(env, out, "val", b); napi_set_named_property(env, out, "found", c);
  return out;
}
napi_value CommRowCount(napi_env env, napi_callback_info info) {
  ARGS(1);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t n = 0; int rc = bmx_comm_row_count(h->c, &n);
  if (rc) return throw_comm(env, h->c, rc);
  napi_value v; napi_create_double(env, (double)n, &v); return v;
}
napi_value CommDumpRows(napi_env env, napi_callback_info info) {
  ARGS(1);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t n = 0; int rc = bmx_comm_row_count(h->c, &n);
  if (rc) return throw_comm(env, h->c, rc);
  std::vector<uint64_t> id(n ? n : 1); std::vector<uint32_t> f(n ? n : 1); std::vector<int64_t> ts(n ? n : 1), val(n ? n : 1);
  uint64_t m = 0;
  rc = bmx_comm_dump_rows(h->c, n, id.data(), f.data(), ts.data(), val.data(), &m);
  if (rc) return throw_comm(env, h->c, rc);
  if (m > n) m = n;                    // tombstones keep their slot and are not dumped
  void *pi, *pf, *pt, *pv;
  napi_value a = make_ta(env, napi_biguint64_array, 8, m, &pi), b = make_ta(env, napi_uint32_array, 4, m, &pf),
             c = make_ta(env, napi_bigint64_array, 8, m, &pt), d = make_ta(env, napi_bigint64_array, 8, m, &pv);
  if (m) { memcpy(pi, id.data(), m * 8); memcpy(pf, f.data(), m * 4); memcpy(pt, ts.data(), m * 8); memcpy(pv, val.data(), m * 8); }
  napi_value out; NAPI_OK(napi_create_object(env, &out));
  napi_set_named_property(env, out, "id", a); napi_set_named_property(env, out, "field", b);
  napi_set_named_property(env, out, "ts", c); napi_set_named_property(env, out, "val", d);
  return out;
}
napi_value CommIndexSetOrdered(napi_env env, napi_callback_info info) {
  ARGS(3);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  uint32_t f, n; NAPI_OK(napi_get_value_uint32(env, argv[1], &f)); NAPI_OK(napi_get_value_uint32(env, argv[2], &n));
  int rc = bmx_comm_index_set_ordered(h->c, f, n);
  if (rc) return throw_comm(env, h->c, rc);
  return nullptr;
}
napi_value CommIndexBuild(napi_env env, napi_callback_info info) {
  ARGS(2);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  std::lock_guard<std::mutex> g(h->mu);
  int rc = bmx_comm_index_build(h->c, f);
  if (rc) return throw_comm(env, h->c, rc);
  return nullptr;
}
napi_value CommIndexDrop(napi_env env, napi_callback_info info) {
  ARGS(2);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  std::lock_guard<std::mutex> g(h->mu);
  for (uint32_t s = 0; s < bmx_comm_nshards(h->c); s++) (void)bmx_index_drop(bmx_comm_shard(h->c, s), f);
  return nullptr;
}
napi_value CommIndexSize(napi_env env, napi_callback_info info) {
  ARGS(2);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t tot = 0;
  for (uint32_t s = 0; s < bmx_comm_nshards(h->c); s++) {
    uint64_t n = 0; int rc = bmx_index_size(bmx_comm_shard(h->c, s), f, &n);
    if (rc) return throw_bmx(env, bmx_comm_shard(h->c, s), rc);
    tot += n;
  }
  napi_value v; napi_create_double(env, (double)tot, &v); return v;
}
napi_value CommScanRange(napi_env env, napi_callback_info info) {
  ARGS(4);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t m = 0; int rc = bmx_comm_scan_count(h->c, f, lo, hi, &m);
  if (rc) return throw_comm(env, h->c, rc);
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, m, &out);
  if (m) { uint64_t m2 = 0; rc = bmx_comm_scan_range(h->c, f, lo, hi, (uint64_t*)out, m, &m2); if (rc) return throw_comm(env, h->c, rc); }
  return ta;
}
napi_value CommScanCount(napi_env env, napi_callback_info info) {
  ARGS(4);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t f; NAPI_OK(napi_get_value_uint32(env, argv[1], &f));
  int64_t lo, hi; if (!get_i64(env, argv[2], &lo) || !get_i64(env, argv[3], &hi)) return nullptr;
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t m = 0; int rc = bmx_comm_scan_count(h->c, f, lo, hi, &m);
  if (rc) return throw_comm(env, h->c, rc);
  napi_value v; napi_create_double(env, (double)m, &v); return v;
}
napi_value CommScanFilter(napi_env env, napi_callback_info info) {
  ARGS(2);
  CommHandle* h; if (!get_comm(env, argv[0], &h)) return nullptr;
  uint32_t nt = 0; NAPI_OK(napi_get_array_length(env, argv[1], &nt));
  if (nt == 0 || nt > 8) { napi_throw_range_error(env, nullptr, "bmx: filter needs 1..8 terms"); return nullptr; }
  bmx_term terms[8];
  for (uint32_t k = 0; k < nt; k++) {
    napi_value t, e0, e1, e2;
    NAPI_OK(napi_get_element(env, argv[1], k, &t));
    NAPI_OK(napi_get_element(env, t, 0, &e0)); NAPI_OK(napi_get_element(env, t, 1, &e1)); NAPI_OK(napi_get_element(env, t, 2, &e2));
    NAPI_OK(napi_get_value_uint32(env, e0, &terms[k].field));
    terms[k].reserved = 0;
    if (!get_i64(env, e1, &terms[k].lo) || !get_i64(env, e2, &terms[k].hi)) return nullptr;
  }
  std::lock_guard<std::mutex> g(h->mu);
  uint64_t m = 0; int rc = bmx_comm_scan_filter(h->c, nt, terms, nullptr, 0, &m);
  if (rc) return throw_comm(env, h->c, rc);
  void* out; napi_value ta = make_ta(env, napi_biguint64_array, 8, m, &out);
  if (m) { uint64_t m2 = 0; rc = bmx_comm_scan_filter(h->c, nt, terms, (uint64_t*)out, m, &m2); if (rc) return throw_comm(env, h->c, rc); }
  return ta;
}

napi_value Init(napi_env env, napi_value exports) {
  struct { const char* name; napi_callback fn; } fns[] = {
      {"abiVersion", AbiVersion}, {"create", Create}, {"destroy", Destroy}, {"mergeBatch", MergeBatch}, {"mergeBatchAsync", MergeBatchAsync}, {"reserve", Reserve}, {"loadRows", LoadRows}, {"putRows", PutRows}, {"hostColumns", HostColumns}, {"scanRangePos", ScanRangePos}, {"indexIds", IndexIds}, {"commPutRows", CommPutRows},
      {"getRows", GetRows}, {"rowCount", RowCount}, {"dumpRows", DumpRows}, {"indexBuild", IndexBuild}, {"indexDrop", IndexDrop}, {"indexSetOrdered", IndexSetOrdered}, {"indexOrderedInfo", IndexOrderedInfo},
      {"indexSize", IndexSize}, {"indexRefreshCounts", IndexRefreshCounts}, {"scanRange", ScanRange}, {"scanCount", ScanCount}, {"scanFilter", ScanFilter}, {"info", Info},
      {"vcCreate", VcCreate}, {"vcDestroy", VcDestroy}, {"vcLoadRows", VcLoadRows}, {"vcMergeBatch", VcMergeBatch}, {"vcMergeBatchAsync", VcMergeBatchAsync}, {"vcGetRows", VcGetRows}, {"vcRowCount", VcRowCount}, {"vcScanRange", VcScanRange}, {"ownersOf", OwnersOf},
      {"commCreate", CommCreate}, {"commDestroy", CommDestroy}, {"commMergeBatch", CommMergeBatch}, {"commLoadRows", CommLoadRows}, {"commGetRows", CommGetRows},
      {"commRowCount", CommRowCount}, {"commDumpRows", CommDumpRows}, {"commIndexBuild", CommIndexBuild}, {"commIndexSetOrdered", CommIndexSetOrdered}, {"commIndexDrop", CommIndexDrop}, {"commIndexSize", CommIndexSize},
      {"commScanRange", CommScanRange}, {"commScanCount", CommScanCount}, {"commScanFilter", CommScanFilter}};
  for (auto& f : fns) {
    napi_value v;
    if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v) != napi_ok) return nullptr;
    napi_set_named_property(env, exports, f.name, v);
  }
  struct { const char* name; int v; } consts[] = {{"INSERT_REFERENCE", BMX_INSERT_REFERENCE}, {"INSERT_DELTA", BMX_INSERT_DELTA},
                                                  {"MERGE_UNIQUE_KEYS", BMX_MERGE_UNIQUE_KEYS}, {"MERGE_STRICT_FLAGS", BMX_MERGE_STRICT_FLAGS}, {"MERGE_MARK_CREATED", BMX_MERGE_MARK_CREATED}, {"FLAG_INCOMING", BMX_FLAG_INCOMING},
                                                  {"FLAG_CURRENT", BMX_FLAG_CURRENT}, {"FLAG_HISTORICAL", BMX_FLAG_HISTORICAL}, {"FLAG_CONCURRENT", BMX_FLAG_CONCURRENT},
                                                  {"VC_MAX_WRITERS", BMX_VC_MAX_WRITERS}, {"VC_ABSENT", BMX_VC_ABSENT}, {"VC_DENSE", BMX_VC_DENSE}, {"VC_SPARSE", BMX_VC_SPARSE}};
  for (auto& c : consts) { napi_value v; napi_create_int32(env, c.v, &v); napi_set_named_property(env, exports, c.name, v); }
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
