"use strict";
/*
 * e2e_rate.js — end-to-end rate of the JS host on this box (bench.py's `js_host` key): sync-chunk entries in, winners out.
 *   mergeEntries          : [{path, data: {f: int}, vectorClock: {w: ts}}] -> path hashing + one clock-row delta per entry + GPU merge + winners' value rows
 *   mergeEntriesPipelined : the same through mergeEntriesAsync, two chunks in flight (chunk b + 1 is packed while chunk b is on the GPU)
 *   mergeBatch            : typed columns -> GPU merge (what is left when the host keeps its keys hashed)
 * Resident graph R keys, B batches of D entries (10 % new keys). One thread. Needs an MI355X.
 * Usage: node e2e_rate.js [R] [D] [B]   -> one JSON line
 */
const { GpuCRT, hash } = require("..");
const R = parseInt(process.argv[2] || "1000000", 10);
const D = parseInt(process.argv[3] || "200000", 10);
const B = parseInt(process.argv[4] || "5", 10);
let s = 12345;
const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: 2 * (R + B * D) });
const g = crt.graph;
{
  // resident nodes: a clock row (what entries are resolved against) and the value row of their field
  const cols = new hash.Columns(2 * R);
  const f = g.keys.fieldOf("n", "f"), fc = g.keys.fieldOf("n", hash.NODE_CLOCK);
  for (let i = 0; i < R; i++) {
    const id = g.keys.idOf("n/k" + i), ts = 1000000 + (rnd() % 1000000);
    cols.set(2 * i, id, fc, ts, 0); cols.set(2 * i + 1, id, f, ts, (rnd() % 2001) - 1000);
  }
  g.loadRows(cols);
}
const mkBatches = (salt, count = B, viaJson = false) => {
  const out = [];
  for (let b = 0; b < count; b++) {
    const entries = new Array(D);
    for (let j = 0; j < D; j++) {
      const ins = rnd() % 100 < 10;
      const clock = { w: 1000000 + (rnd() % 2000000) };
      entries[j] = { path: "n/k" + (ins ? R + salt * B * D + b * D + j : rnd() % R), data: { f: (rnd() % 2001) - 1000 }, vectorClock: clock };
    }
    out.push(viaJson ? JSON.parse(JSON.stringify(entries)) : entries);      // viaJson: the object shapes a parsed network message has
  }
  return out;
};
const ONLY = process.argv[5] === "only" ? process.argv[6] : null;      // "only apply" / "only vector": just that section (a 4M-entry run of every section does not fit node 12's heap)
const batches = ONLY ? [] : mkBatches(0), batches2 = ONLY ? [] : mkBatches(1);
if (!ONLY) { // warm the addon, the JIT and the pool of page-locked column sets with one chunk of the timed size: resident nodes under a clock below everything stored
  // (all historical: no row, no dictionary entry and no store state changes)
  const warm = new Array(D);
  for (let j = 0; j < D; j++) warm[j] = { path: "n/k" + (j % R), data: { f: j & 1023 }, vectorClock: { w: 1 } };
  const r0 = crt.mergeEntries(warm);
  if (r0.nApplied !== 0) throw new Error("warm-up chunk was supposed to be historical");
}
let applied = 0;
let t0 = process.hrtime.bigint();
for (const entries of batches) applied += crt.mergeEntries(entries).nApplied;
const dtEntries = Number(process.hrtime.bigint() - t0) / 1e9;
(async () => {
let dtPipe = 0;
if (!ONLY) {
await crt.mergeEntriesAsync(batches[0].slice(1000, 2000));      // the asynchronous path's first call (worker start-up, JIT)
t0 = process.hrtime.bigint();
const pr = await crt.mergeEntriesPipelined(batches2);
applied += pr.nApplied;
dtPipe = Number(process.hrtime.bigint() - t0) / 1e9;
}
// the same amount of work with the keys already hashed (typed columns in, winners out)
const colsB = [];
for (let b = 0; b < B && !ONLY; b++) {
  const cols = g.takeColumns(D);                 // page-locked column sets (the way a host builds typed batches: INTEGRATION.md)
  const f = g.keys.fieldOf("n", "f");
  for (let j = 0; j < D; j++) cols.set(j, g.keys.idOf(batches[b][j].path), f, 3000000 + (rnd() % 1000000), (rnd() % 2001) - 1000);
  colsB.push(cols.slice(D));
}
t0 = process.hrtime.bigint();
for (const cols of colsB) applied += crt.mergeBatch(cols).nApplied;
const dtCols = Number(process.hrtime.bigint() - t0) / 1e9;
// the whole ingestion seam with the store kept (attach(..., {batchSync}) -> processSyncEntries, apply: true: store, meta, op log, listeners, value rows),
// against the reference's loop body entry by entry through the host resolver (src/bullet-network-sync.js:551-569)
let applied_path = null, lazy_path = null;
if (process.argv[5] === "lazy" || process.argv[6] === "lazy") {
  // the same seam with the store following LAZILY (attach(..., {batchSync: {lazyStore}}), lazy-store.js): what the chunks cost while they arrive, and what the fold
  // that any read of the store triggers costs afterwards (here: all at once, right behind the last chunk)
  batches.length = 0; batches2.length = 0; colsB.length = 0;
  const { attach } = require("..");
  const MiniBullet = require("./mini-bullet");
  const ab = new MiniBullet("w");
  const h = attach(ab, { capacityRows: 2 * (R + B * D), batchSync: { lazyStore: { idleSlice: 0 } } });
  for (let r0 = 0; r0 < R; r0 += 250000) {
    const seed = new Array(Math.min(250000, R - r0));
    for (let i = 0; i < seed.length; i++) seed[i] = { path: "n/k" + (r0 + i), data: { f: (r0 + i) & 1023 }, vectorClock: { w: 5 } };
    h.sync.processSyncEntries(seed);
  }
  void ab.store;                                                   // the resident graph is in the store before the clock starts
  const chunks = mkBatches(3, B, true);
  h.sync.processSyncEntries(chunks[0].slice(0, 1000)); void ab.store;
  t0 = process.hrtime.bigint();
  for (const c of chunks) h.sync.processSyncEntries(c);
  const dtL = Number(process.hrtime.bigint() - t0) / 1e9;
  const pending = h.lazyStore.n;
  t0 = process.hrtime.bigint();
  const nodes = Object.keys(ab.store.n || {}).length;              // the first read folds every recorded winner in
  const dtF = Number(process.hrtime.bigint() - t0) / 1e9;
  lazy_path = { batchSync_lazy_entries_per_s: (B * D) / dtL, incl_the_fold_entries_per_s: (B * D) / (dtL + dtF), winners_recorded: pending, fold_s: dtF, nodes };
  ab.close();
}
if (process.argv[5] === "apply" || process.argv[6] === "apply") {
  batches.length = 0; batches2.length = 0; colsB.length = 0;
  const { attach } = require("..");
  const MiniBullet = require("./mini-bullet");
  const ab = new MiniBullet("w");
  const h = attach(ab, { capacityRows: 2 * (R + B * D), batchSync: {} });
  // the R resident nodes first, through the seam itself and untimed (first sights: store, meta and device rows of a graph that is already there) — the
  // timed chunks then are what a running peer sees: 90 % updates of nodes it holds, 10 % new ones
  for (let r0 = 0; r0 < R; r0 += 250000) {
    const seed = new Array(Math.min(250000, R - r0));
    for (let i = 0; i < seed.length; i++) seed[i] = { path: "n/k" + (r0 + i), data: { f: (r0 + i) & 1023 }, vectorClock: { w: 5 } };
    h.sync.processSyncEntries(seed);
  }
  const chunks = mkBatches(3, B, true), chunks2 = mkBatches(3, Math.max(1, B >> 2), true);
  h.sync.processSyncEntries(chunks[0].slice(0, 1000));
  const T = {};
  if (process.env.E2E_PHASES) {      // where the time of the seam goes (wrappers around the phases of GpuCRT.mergeEntries; ns per entry)
    const wrap = (obj, name, label) => { const fn = obj[name]; obj[name] = function (...a) { const t = process.hrtime.bigint(); try { return fn.apply(this, a); } finally { T[label] = (T[label] || 0) + Number(process.hrtime.bigint() - t); } }; };
    wrap(h.crt, "_packEntries", "pack"); wrap(h.crt, "mergeBatch", "merge (addon + GPU)"); wrap(h.crt, "_applyWinners", "apply winners (store, meta, log, value rows)");
    wrap(h.crt, "_unaliasLosers", "losers"); wrap(h.crt, "_flushDeviceWrites", "put rows"); wrap(h.crt, "mergeEntries", "mergeEntries total");
  }
  t0 = process.hrtime.bigint();
  for (const c of chunks) h.sync.processSyncEntries(c);
  const dtA = Number(process.hrtime.bigint() - t0) / 1e9;
  if (process.env.E2E_PHASES) console.error("phases (ns per entry): total " + Math.round(dtA * 1e9 / (B * D)) + "; " + Object.keys(T).map((k) => k + " " + Math.round(T[k] / (B * D))).join("; "));
  const tb = new MiniBullet("w"); tb.crt = new GpuCRT(tb);
  for (let i = 0; i < R; i++) tb.setData("n/k" + i, { f: i & 1023, __fromNetwork: true, __vectorClock: { w: 5 } }, false);   // the same resident graph for the per-entry loop
  const few = chunks2;
  t0 = process.hrtime.bigint();
  for (const c of few) for (const e of c) tb.setData(e.path, Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }), false);
  const dtH = Number(process.hrtime.bigint() - t0) / 1e9;
  applied_path = { batchSync_apply_entries_per_s: (B * D) / dtA, per_entry_host_loop_entries_per_s: (few.length * D) / dtH, nodes: Object.keys(ab.store.n || {}).length };
  ab.close();
}
// general vector clocks (N4): the same entries under clocks over ordered subsets of three writers, nodes' clock rows in the vector-clock table
let vector = null;
if (process.argv[5] === "vector" || process.argv[6] === "vector") {
  batches.length = 0; batches2.length = 0; colsB.length = 0;
  const WR = ["a", "b", "w"];
  const vcrt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { writers: WR, capacityRows: 2 * (R + B * D) });
  const vb = mkBatches(2);
  for (const entries of vb) for (const e of entries) { const c = {}; const k = rnd() % 4; for (let x = 0; x < k; x++) c[WR[(x + (rnd() % 3)) % 3]] = rnd() % 5; e.vectorClock = c; }
  vcrt.mergeEntries(vb[0].slice(0, 1000));
  t0 = process.hrtime.bigint();
  let conc = 0;
  for (const entries of vb) conc += vcrt.mergeEntries(entries).nConflicts;
  vector = { mergeEntries_per_s: (B * D) / (Number(process.hrtime.bigint() - t0) / 1e9), concurrent_merges: conc, writers: 3 };
  // ... and through mergeEntriesAsync (vcMergeBatchAsync on a worker thread of the addon), two chunks in flight: chunk b + 1 is packed while chunk b is merged
  const vb2 = mkBatches(5);
  for (const entries of vb2) for (const e of entries) { const c = {}; const k = rnd() % 4; for (let x = 0; x < k; x++) c[WR[(x + (rnd() % 3)) % 3]] = rnd() % 5; e.vectorClock = c; }
  await vcrt.mergeEntriesAsync(vb2[0].slice(0, 1000));
  t0 = process.hrtime.bigint();
  await vcrt.mergeEntriesPipelined(vb2);
  vector.mergeEntriesPipelined_per_s = (B * D) / (Number(process.hrtime.bigint() - t0) / 1e9);
  vector.host_only = vcrt.hostOnlyInfo();
  vcrt.close();
}
console.log(JSON.stringify({ applied_path, lazy_path, vector, mergeEntries_per_s: ONLY ? null : (B * D) / dtEntries, mergeEntriesPipelined_per_s: ONLY ? null : (B * D) / dtPipe, mergeBatch_typed_columns_per_s: ONLY ? null : (B * D) / dtCols, unit: "deltas/s", resident_keys: R,
  entries_per_batch: D, batches: B, applied, node: process.version }));
crt.close();
})().catch((e) => { console.error(e); process.exit(1); });
