"use strict";
/*
 * js_apply_profile.js — DEVELOPMENT TOOL, not product and not a test oracle: where does the HOST time of the store-kept ingestion seam go
 * (attach(bullet, {batchSync}) -> processSyncEntries -> GpuCRT.mergeEntries({apply: true}) -> batch-apply.js)? Runs without a GPU: the device
 * graph is replaced by a stub whose "merge" is a last-writer-wins table over a JS Map, so every number printed here is V8 work only (packing,
 * winners, store, meta, log). Nothing in bullet-js_amd/ or tests/ loads this file. Usage: node [--prof] js_apply_profile.js [R] [D] [B]
 */
const path = require("path");
const JS = path.join(__dirname, "..", "bullet-js_amd", "js");
const { attach, hash } = require(JS);
const MiniBullet = require(path.join(JS, "test", "mini-bullet"));
const R = parseInt(process.argv[2] || "300000", 10), D = parseInt(process.argv[3] || "200000", 10), B = parseInt(process.argv[4] || "5", 10);

class StubGraph {
  constructor() {
    this.keys = new hash.KeyDictionary();
    this.native = { INSERT_REFERENCE: 0, INSERT_DELTA: 1, MERGE_UNIQUE_KEYS: 0x100, MERGE_STRICT_FLAGS: 0x200, MERGE_MARK_CREATED: 0x1000 };
    this.rows = new Map(); this.comm = null; this.preOp = null; this.tMerge = 0;
  }
  takeColumns(n) { return new hash.Columns(Math.max(n, 1)); }
  giveColumns() {}
  mergeBatch(cols, mode) {
    const t0 = process.hrtime.bigint();
    const n = cols.n, win = new Map();
    for (let j = 0; j < n; j++) {
      const k = cols._id32[2 * j + 1] * 4294967296 + cols._id32[2 * j];
      const ts = cols._ts32[2 * j + 1] * 4294967296 + cols._ts32[2 * j];
      const cur = this.rows.get(k);
      if (cur === undefined) { this.rows.set(k, 2); win.set(k, j | 0x80000000); }
      else if (ts >= cur) { this.rows.set(k, ts); win.set(k, j); }
    }
    const applied = Uint32Array.from(win.values()).sort((a, b) => (a & 0xffffff) - (b & 0xffffff));
    this.tMerge += Number(process.hrtime.bigint() - t0);
    return { applied, flags: null, nApplied: applied.length, nConflicts: 0, nRows: this.rows.size };
  }
  putRows() {}
  getRows() { throw new Error("stub"); }
  close() {}
}

let s = 12345;
const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
const mk = (salt) => {
  const out = [];
  for (let b = 0; b < B; b++) {
    const entries = new Array(D);
    for (let j = 0; j < D; j++) {
      const ins = rnd() % 100 < 10;
      entries[j] = { path: "n/k" + (ins ? R + salt * B * D + b * D + j : rnd() % R), data: { f: (rnd() % 2001) - 1000 }, vectorClock: { w: 1000000 + (rnd() % 2000000) } };
    }
    out.push(entries);
  }
  return out;
};
const ab = new MiniBullet("w");
const g = new StubGraph();
const h = attach(ab, { graph: g, batchSync: {} });
// resident nodes through the seam itself (first sights), untimed
{
  const seed = new Array(R);
  for (let i = 0; i < R; i++) seed[i] = { path: "n/k" + i, data: { f: i & 1023 }, vectorClock: { w: 5 } };
  h.sync.processSyncEntries(seed);
}
const chunks = mk(3);
g.tMerge = 0;
const t0 = process.hrtime.bigint();
for (const c of chunks) h.sync.processSyncEntries(c);
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ entries_per_s_host_only: Math.round((B * D) / (dt - g.tMerge / 1e9)), seconds: +dt.toFixed(3), stub_merge_seconds: +(g.tMerge / 1e9).toFixed(3), nodes: Object.keys(ab.store.n).length, stats: h.sync.stats }));
// ---- second pass with per-phase timers (wrappers cost a little; read the shares, not the total)
if (process.env.PHASES) {
  const T = {};
  const wrap = (obj, name, label) => {
    const fn = obj[name];
    obj[name] = function (...a) { const t = process.hrtime.bigint(); try { return fn.apply(this, a); } finally { T[label] = (T[label] || 0) + Number(process.hrtime.bigint() - t); } };
  };
  const crt = h.crt;
  wrap(crt, "_packEntries", "pack"); wrap(crt, "_applyWinners", "applyWinners(incl applyBatch)"); wrap(crt, "_unaliasLosers", "unaliasLosers"); wrap(crt, "_notifyIndexHook", "indexHook");
  wrap(crt, "mergeBatch", "merge(stub)"); wrap(crt, "mergeEntries", "mergeEntries(total)");
  ab._applyBatch = function (updates, fromNet) { const t = process.hrtime.bigint(); const r = require(path.join(JS, "batch-apply")).applyBatch(ab, updates, fromNet, false); T.applyBatch = (T.applyBatch || 0) + Number(process.hrtime.bigint() - t); return r; };
  const chunks2 = mk(4);
  const t1 = process.hrtime.bigint();
  for (const c of chunks2) h.sync.processSyncEntries(c);
  const tot = Number(process.hrtime.bigint() - t1);
  const per = (x) => (x / (B * D)).toFixed(0) + " ns/entry";
  console.log("phases: total " + per(tot) + "; " + Object.keys(T).map((k) => k + " " + per(T[k])).join("; "));
}
if (process.env.DIGEST) {   // differential check of two forms of the apply pass: same store, meta (without timestamps) and log tail?
  const crypto = require("crypto");
  const m = {}; for (const k of Object.keys(ab.meta)) m[k] = [ab.meta[k].source, ab.meta[k].vectorClock];
  const h1 = crypto.createHash("sha1").update(JSON.stringify(ab.store)).digest("hex"), h2 = crypto.createHash("sha1").update(JSON.stringify(m)).digest("hex");
  const h3 = crypto.createHash("sha1").update(JSON.stringify(ab.log.map((r) => [r.op, r.path, r.data, r.vectorClock]))).digest("hex");
  console.log("digest store", h1, "meta", h2, "log", h3, "log length", ab.log.length, "lazy clocks", h.crt._nLazy);
}
