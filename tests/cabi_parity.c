/* cabi_parity.c — plain C against include/bmx.h: what a non-Python, non-JS host sees. Links libbmx.so (product) and
 * libbmx_oracle.so (checker, tests only). Built and run by tests/test_cabi_c_program.py on the GPU box.
 * Exit code 0 = every check passed. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bmx.h"

/* oracle prototypes (oracle/bmx_oracle.c) */
typedef struct orc orc_t;
orc_t* orc_create(void);
void orc_destroy(orc_t*);
void orc_load_rows(orc_t*, uint64_t, const uint64_t*, const uint32_t*, const int64_t*, const int64_t*);
uint64_t orc_merge_batch(orc_t*, uint64_t, const uint64_t*, const uint32_t*, const int64_t*, const int64_t*, int, uint8_t*, uint32_t*);
int orc_get_row(orc_t*, uint64_t, uint32_t, int64_t*, int64_t*);
uint64_t orc_scan_range(const orc_t*, uint32_t, int64_t, int64_t, uint64_t*, uint64_t);
uint64_t orc_size(const orc_t*);

static uint64_t sm(uint64_t x) { uint64_t z = x + 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #c, bmx_last_error(ctx)); return 1; } } while (0)

int main(void) {
  enum { R = 50000, D = 20000, F = 0x1234abcd };
  bmx_ctx* ctx = NULL;
  if (bmx_create(0, 4 * (R + 3 * D), 0, &ctx) != BMX_OK) { fprintf(stderr, "bmx_create: %s\n", bmx_last_error(NULL)); return 2; }
  orc_t* o = orc_create();
  uint64_t* id = malloc(sizeof(uint64_t) * R); uint32_t* f = malloc(sizeof(uint32_t) * R); int64_t* ts = malloc(sizeof(int64_t) * R); int64_t* val = malloc(sizeof(int64_t) * R);
  for (int i = 0; i < R; i++) { id[i] = sm(i + 1); f[i] = F; ts[i] = 1000 + (int64_t)(sm(i * 7 + 3) % 1000); val[i] = (int64_t)(sm(i * 11 + 5) % 2001) - 1000; }
  CHECK(bmx_load_rows(ctx, R, id, f, ts, val, BMX_MEM_HOST) == BMX_OK);
  orc_load_rows(o, R, id, f, ts, val);
  uint64_t nrows = 0; CHECK(bmx_row_count(ctx, &nrows) == BMX_OK && nrows == R);

  uint64_t* did = malloc(sizeof(uint64_t) * D); uint32_t* df = malloc(sizeof(uint32_t) * D); int64_t* dts = malloc(sizeof(int64_t) * D); int64_t* dval = malloc(sizeof(int64_t) * D);
  uint32_t* applied = malloc(sizeof(uint32_t) * D); uint32_t* want = malloc(sizeof(uint32_t) * D); uint8_t* flags = malloc(D);
  for (int b = 0; b < 3; b++) {
    for (int j = 0; j < D; j++) {
      uint64_t u = sm(1000003ULL * b + j);
      uint64_t row = (u % 100 < 15) ? R + (sm(u) % 5000) : ((u >> 8) % 100 < 30 ? sm(u) % 40 : sm(u ^ 99) % R);   /* inserts, hot keys, uniform hits */
      did[j] = sm(row + 1); df[j] = F; dts[j] = 1000 + 500 * b + (int64_t)(sm(u + 17) % 2000); dval[j] = (int64_t)(sm(u + 31) % 2001) - 1000;
    }
    uint64_t na = 0; bmx_merge_stats st;
    CHECK(bmx_merge_batch(ctx, D, did, df, dts, dval, BMX_INSERT_REFERENCE, BMX_MEM_HOST, applied, &na, flags, &st) == BMX_OK);
    uint64_t nw = orc_merge_batch(o, D, did, df, dts, dval, 0, NULL, want);
    CHECK(na == nw && st.n_applied == nw && st.n_rows == orc_size(o));
    CHECK(memcmp(applied, want, nw * sizeof(uint32_t)) == 0);
  }
  /* one more batch from page-locked arrays (bmx_host_alloc): same call, same answer */
  {
    void* pin = NULL; uint32_t* papp = NULL;
    CHECK(bmx_host_alloc(28ull * D, &pin) == BMX_OK && pin != NULL);
    CHECK(bmx_host_alloc(4ull * D, (void**)&papp) == BMX_OK);
    uint64_t* pid = (uint64_t*)pin; int64_t* pts = (int64_t*)((char*)pin + 8ull * D); int64_t* pval = (int64_t*)((char*)pin + 16ull * D); uint32_t* pf = (uint32_t*)((char*)pin + 24ull * D);
    for (int j = 0; j < D; j++) {
      uint64_t u = sm(7000003ULL + j);
      pid[j] = sm((u % 100 < 10 ? R + 5000 + (sm(u) % 3000) : sm(u ^ 5) % R) + 1); pf[j] = F; pts[j] = 3000 + (int64_t)(sm(u + 17) % 2000); pval[j] = (int64_t)(sm(u + 31) % 2001) - 1000;
    }
    uint64_t na = 0; bmx_merge_stats st;
    CHECK(bmx_merge_batch(ctx, D, pid, pf, pts, pval, BMX_INSERT_REFERENCE, BMX_MEM_HOST, papp, &na, NULL, &st) == BMX_OK);
    uint64_t nw = orc_merge_batch(o, D, pid, pf, pts, pval, 0, NULL, want);
    CHECK(na == nw && st.n_rows == orc_size(o) && memcmp(papp, want, nw * sizeof(uint32_t)) == 0);
    CHECK(bmx_host_free(pin) == BMX_OK && bmx_host_free(papp) == BMX_OK && bmx_host_free(NULL) == BMX_OK);
    CHECK(bmx_host_alloc(0, &pin) == BMX_ERR_INVALID);
  }
  /* point reads */
  for (int i = 0; i < 200; i++) {
    uint64_t k = sm((uint64_t)(i * 37 % (R + 5000)) + 1); int64_t t1, v1, t2, v2;
    int g = bmx_get_row(ctx, k, F, &t1, &v1), w = orc_get_row(o, k, F, &t2, &v2);
    CHECK(g == w && (!g || (t1 == t2 && v1 == v2)));
  }
  /* index scan */
  uint64_t cap = orc_size(o); uint64_t* a = malloc(8 * cap); uint64_t* b2 = malloc(8 * cap); uint64_t m = 0;
  CHECK(bmx_scan_range(ctx, F, -100, 250, a, cap, &m, BMX_MEM_HOST) == BMX_OK);
  uint64_t mw = orc_scan_range(o, F, -100, 250, b2, cap);
  CHECK(m == mw);
  qsort(a, m, 8, cmp_u64); qsort(b2, mw, 8, cmp_u64);
  CHECK(memcmp(a, b2, 8 * m) == 0);
  uint64_t c = 0; CHECK(bmx_scan_count(ctx, F, 7, 7, &c, BMX_MEM_HOST) == BMX_OK && c == orc_scan_range(o, F, 7, 7, NULL, 0));
  /* the same queries through the value-ordered view of the index (bmx_index_set_ordered): same sets, same counts */
  {
    uint32_t after = 0; int valid = 0; uint64_t sorts = 0;
    CHECK(bmx_index_set_ordered(ctx, F, 1) == BMX_OK);
    CHECK(bmx_scan_range(ctx, F, -100, 250, a, cap, &m, BMX_MEM_HOST) == BMX_OK && m == mw);
    qsort(a, m, 8, cmp_u64);
    CHECK(memcmp(a, b2, 8 * m) == 0);
    CHECK(bmx_index_ordered_info(ctx, F, &after, &valid, &sorts) == BMX_OK && after == 1 && valid == 1 && sorts == 1);
    CHECK(bmx_scan_count(ctx, F, 7, 7, &c, BMX_MEM_HOST) == BMX_OK && c == orc_scan_range(o, F, 7, 7, NULL, 0));
    CHECK(bmx_scan_count(ctx, F, 5, 4, &c, BMX_MEM_HOST) == BMX_OK && c == 0);
    CHECK(bmx_index_set_ordered(ctx, F, 0) == BMX_OK);
    CHECK(bmx_index_ordered_info(ctx, F, &after, &valid, &sorts) == BMX_OK && after == 0 && valid == 0);
    CHECK(bmx_index_ordered_info(ctx, F + 12345, &after, &valid, &sorts) == BMX_ERR_INVALID);
  }
  /* errors are codes + text, never aborts */
  uint64_t badid = ~0ULL; uint32_t bf = F; int64_t bt = 1, bv = 1;
  CHECK(bmx_merge_batch(ctx, 1, &badid, &bf, &bt, &bv, BMX_INSERT_REFERENCE, BMX_MEM_HOST, NULL, NULL, NULL, NULL) == BMX_ERR_RANGE);
  CHECK(strlen(bmx_last_error(ctx)) > 0);
  CHECK(bmx_merge_batch(ctx, 1, &badid, &bf, &bt, &bv, 77, BMX_MEM_HOST, NULL, NULL, NULL, NULL) == BMX_ERR_INVALID);
  /* K-writer vector clocks with key sets: {b:1,a:2} merges with {w:2} into {b:1,a:2,w:2}; then {a:2,b:1,w:2} — equal counters, another key ORDER: not identical
     (JSON.stringify) -> concurrent, and the stored clock lists the incoming clock's keys first (src/bullet-crt.js:103-114, 200-203) */
  {
    bmx_vc* t = NULL;
    CHECK(bmx_vc_create(0, 1024, 3, 2, &t) == BMX_OK);
    const uint8_t ba[2] = {1, 0}, ab[2] = {0, 1};
    uint64_t vid[3] = {sm(1), sm(1), sm(1)}; uint32_t vf[3] = {F, F, F};
    uint32_t vclk[9] = {5, 5, 5, /* first sight: dropped, {w:2} stored */ 2, 1, 0, /* {b:1,a:2}: concurrent with {w:2} -> {b:1,a:2,w:2} */ 2, 1, 2 /* {a:2,b:1,w:2}: the same counters */};
    uint32_t vks[3] = {bmx_vc_keyset_dense(3), bmx_vc_keyset(ba, 2), bmx_vc_keyset_dense(3)};
    (void)ab;
    int64_t vval[3] = {10, 11, 12};
    uint32_t upd[3]; uint64_t nu = 0; uint8_t vfl[3];
    CHECK(bmx_vc_merge_batch_ks(t, 3, vid, vf, vclk, vks, vval, upd, &nu, vfl) == BMX_OK);
    CHECK(vfl[0] == BMX_FLAG_INCOMING && vfl[1] == BMX_FLAG_CONCURRENT && vfl[2] == BMX_FLAG_CONCURRENT && nu == 1 && upd[0] == 2);
    uint32_t oc[3], oks = 0; int64_t ov = 0; uint8_t os = 0;
    CHECK(bmx_vc_get_rows_ks(t, 1, vid, vf, oc, &oks, &ov, &os) == BMX_OK);
    const uint8_t abw[3] = {0, 1, 2};
    CHECK(oc[0] == 2 && oc[1] == 1 && oc[2] == 2 && ov == 12 && oks == bmx_vc_keyset(abw, 3) && os == BMX_VC_DENSE);   /* {a:2, b:1, w:2} */
    uint32_t bad = bmx_vc_keyset(ba, 2); uint32_t badclk[3] = {0, 0, 7};                                                /* a counter for a writer the key set does not name */
    CHECK(bmx_vc_merge_batch_ks(t, 1, vid, vf, badclk, &bad, vval, NULL, NULL, NULL) == BMX_ERR_RANGE);
    bmx_vc_destroy(t);
  }
  bmx_destroy(ctx); orc_destroy(o);
  printf("cabi_parity ok: %d rows, 3 x %d deltas, scan %llu matches\n", R, D, (unsigned long long)m);
  return 0;
}
