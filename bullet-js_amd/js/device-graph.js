"use strict";
/*
 * device-graph.js — the GPU-resident graph plus the host dictionary that maps the device's hashed keys back to Bullet paths.
 * Shared by GpuCRT (merge) and GpuQuery (index scans). One shard (a bmx_ctx) by default; with `devices: [0, 1, ...]` or
 * `shards: N` the graph is split by node-id hash over N contexts owned by ONE handle (a bmx_comm): a delta is merged by the
 * shard that owns its node instead of by every peer (reference fan-out: src/bullet-network.js:378-418), scans run on every shard
 * and are concatenated (src/bullet-query.js:186-261). The methods below behave the same either way.
 */
const { requireNative } = require("./native");
const { KeyDictionary, Columns } = require("./hash");

class DeviceGraph {
  constructor(opts = {}) {
    this.native = requireNative();                 // throws if the addon is missing: no CPU fallback
    this.device = opts.device || 0;
    this.capacityRows = opts.capacityRows || (1 << 16);   // grows on demand (bmx_reserve / automatic rehash)
    this.devices = Array.isArray(opts.devices) ? opts.devices.slice() : (opts.shards > 1 ? new Array(opts.shards).fill(this.device) : null);
    if (this.devices && this.devices.length > 1) {
      this.nShards = this.devices.length;
      this.comm = this.native.commCreate(this.devices, Math.ceil(this.capacityRows / this.nShards) + 1024);
      this.handle = null;
    } else {
      this.nShards = 1;
      this.comm = null;
      this.handle = this.native.create(this.devices ? this.devices[0] : this.device, this.capacityRows);
    }
    this.keys = new KeyDictionary();
    this.batches = 0;
    this.preOp = null;             // GpuCRT hangs its queue of host-decided rows here: flushed in front of every operation below
    // column sets of >= 4096 rows are page-locked (bmx_host_alloc) and reused: the runtime neither pins fresh pages per upload nor stages them
    this._pinned = typeof this.native.hostColumns === "function" && opts.pinnedColumns !== false && process.env.BMX_JS_PINNED !== "0";
    this._pool = new Map();        // capacity (a power of two) -> idle Columns
  }
  /* a Columns of at least n rows to build a batch in; give it back (giveColumns) once the call that read it has returned */
  takeColumns(n) {
    if (n < 4096) return new Columns(Math.max(n, 1));
    let cap = 4096; while (cap < n) cap *= 2;
    const idle = this._pool.get(cap);
    if (idle && idle.length) return idle.pop();
    let backing;
    if (this._pinned) { try { backing = this.native.hostColumns(cap); } catch (e) { this._pinned = false; } }   // no page-locked memory left: plain arrays
    const c = new Columns(cap, backing);
    c._pooled = true;
    return c;
  }
  giveColumns(cols) {
    const root = cols && (cols._root || cols);
    if (!root || !root._pooled) return;
    let idle = this._pool.get(root.n);
    if (!idle) this._pool.set(root.n, idle = []);
    if (idle.length < 4 && idle.indexOf(root) < 0) idle.push(root);
  }
  mergeBatch(cols, mode) {          // (merges come from GpuCRT, which decides itself what has to be flushed in front of them)
    this.batches++;
    if (this.comm) return this.native.commMergeBatch(this.comm, cols.id, cols.field, cols.ts, cols.val, mode | 0);   // no per-delta flags across shards
    return this.native.mergeBatch(this.handle, cols.id, cols.field, cols.ts, cols.val, mode | 0);
  }
  mergeBatchAsync(cols, mode) {
    this.batches++;
    if (this.comm) return Promise.resolve().then(() => this.native.commMergeBatch(this.comm, cols.id, cols.field, cols.ts, cols.val, mode | 0));
    return this.native.mergeBatchAsync(this.handle, cols.id, cols.field, cols.ts, cols.val, mode | 0);
  }
  reserve(capacityRows) { if (!this.comm) this.native.reserve(this.handle, capacityRows); }   // shards grow on their own
  loadRows(cols) { if (this.preOp) this.preOp(); return this.comm ? this.native.commLoadRows(this.comm, cols.id, cols.field, cols.ts, cols.val) : this.native.loadRows(this.handle, cols.id, cols.field, cols.ts, cols.val); }
  /* rows decided on the host, stored as given (bmx_put_rows); val === VAL_DELETED leaves a tombstone. Keys unique within one call. */
  putRows(cols) { return this.comm ? this.native.commPutRows(this.comm, cols.id, cols.field, cols.ts, cols.val) : this.native.putRows(this.handle, cols.id, cols.field, cols.ts, cols.val); }
  getRows(id, field) { if (this.preOp) this.preOp(); return this.comm ? this.native.commGetRows(this.comm, id, field) : this.native.getRows(this.handle, id, field); }
  rowCount() { if (this.preOp) this.preOp(); return this.comm ? this.native.commRowCount(this.comm) : this.native.rowCount(this.handle); }
  dumpRows() { if (this.preOp) this.preOp(); return this.comm ? this.native.commDumpRows(this.comm) : this.native.dumpRows(this.handle); }
  indexBuild(f) { if (this.preOp) this.preOp(); return this.comm ? this.native.commIndexBuild(this.comm, f) : this.native.indexBuild(this.handle, f); }
  indexDrop(f) { return this.comm ? this.native.commIndexDrop(this.comm, f) : this.native.indexDrop(this.handle, f); }
  /* value-ordered view of the index on field f (bmx_index_set_ordered): equals / range answer in O(log R + matches) while the field is not written;
   * afterQueries: the view is sorted again by that many queries after a change (0 = off). On a sharded graph every shard keeps its own view. */
  indexSetOrdered(f, afterQueries = 2) {
    if (this.preOp) this.preOp();
    if (this.comm) this.native.commIndexSetOrdered(this.comm, f, afterQueries >>> 0); else this.native.indexSetOrdered(this.handle, f, afterQueries >>> 0);
    return true;
  }
  indexOrderedInfo(f) { return this.comm ? null : this.native.indexOrderedInfo(this.handle, f); }
  indexSize(f) { if (this.preOp) this.preOp(); return this.comm ? this.native.commIndexSize(this.comm, f) : this.native.indexSize(this.handle, f); }
  /* {fullBuilds, incremental}: index rebuilds from the table vs updates from the merges' change log (one context only) */
  indexRefreshCounts() { return this.comm ? null : this.native.indexRefreshCounts(this.handle); }
  scanRange(f, lo, hi) { if (this.preOp) this.preOp(); return this.comm ? this.native.commScanRange(this.comm, f, lo, hi) : this.native.scanRange(this.handle, f, lo, hi); }
  /* positions of the matches in the index columns (one context only: shards number their rows independently) and the ids behind positions */
  scanRangePos(f, lo, hi) { if (this.preOp) this.preOp(); return this.native.scanRangePos(this.handle, f, lo, hi); }
  indexIds(f, first, count) { if (this.preOp) this.preOp(); return this.native.indexIds(this.handle, f, first, count); }
  scanCount(f, lo, hi) { if (this.preOp) this.preOp(); return this.comm ? this.native.commScanCount(this.comm, f, lo, hi) : this.native.scanCount(this.handle, f, lo, hi); }
  scanFilter(terms) { if (this.preOp) this.preOp(); return this.comm ? this.native.commScanFilter(this.comm, terms) : this.native.scanFilter(this.handle, terms); }
  info() { return this.comm ? { nShards: this.nShards, devices: this.devices, nRows: this.rowCount() } : this.native.info(this.handle); }
  close() {
    if (this.handle) { this.native.destroy(this.handle); this.handle = null; }
    if (this.comm) { this.native.commDestroy(this.comm); this.comm = null; }
  }
}

/*
 * DeviceVcTable — N4 (SURVEY §8(f)): rows with a K-writer vector clock (a bmx_vc). `writers` fixes the component order of the writers known up
 * front; opts.maxWriters (<= 8) makes the table that wide and lets unknown writers take the free components as clocks name them;
 * `local` is this peer's id (the reference stores clock {local: 2} for a first write). Host typed arrays in and out.
 */
/* writer id -> component of the table. With room left (opts.maxWriters > writers given) a writer nobody has named before takes the next free
 * component the first time a clock names it — a gossip mesh learns its peers as it goes (src/bullet-network.js:404-418), the table does not have to be
 * told them up front; once all components are taken, clocks naming yet another writer stay on the host (GpuCRT.hostOnlyInfo() counts them).
 * Integer-like ids never get one: a JS object moves such keys to the front whatever the insertion order, so the key ORDER of the reference's merged
 * clocks (part of their identity: src/bullet-crt.js:200-203) would not be the order the device tracks. */
const NUMERIC_ID = /^(0|[1-9][0-9]*)$/;
class WriterIndex extends Map {
  constructor(table) { super(); this._t = table; }
  get(w) {
    let k = super.get(w);
    if (k === undefined) {
      const t = this._t;
      if (t.writers.length < t.K && typeof w === "string" && w !== "" && !NUMERIC_ID.test(w)) { k = t.writers.length; t.writers.push(w); super.set(w, k); }
    }
    return k;
  }
}

class DeviceVcTable {
  /** opts: { device, capacityRows | vcCapacityRows, shards: n (logical shards on one GPU) | devices: [..] (one table per GPU) } — with more than one
   *  table, rows are owned by bmx_owner_of(node id): a batch is split on the host in batch order (a key never straddles tables, so the
   *  order-dependent outcome of concurrent clocks is the single-table outcome) and flags / updated indices come back in the caller's index space. */
  constructor(writers, local, opts = {}) {
    this.native = requireNative();
    const maxW = opts.maxWriters === undefined ? 0 : opts.maxWriters | 0;
    if (!Array.isArray(writers) || writers.length < 1 || writers.length > this.native.VC_MAX_WRITERS || writers.indexOf(local) < 0 ||
        (maxW !== 0 && (maxW < writers.length || maxW > this.native.VC_MAX_WRITERS))) {
      const err = new Error("bmx: vector-clock mode needs 1.." + this.native.VC_MAX_WRITERS + " writer ids that include this peer's id");
      err.code = "BMX_BAD_WRITERS";
      throw err;
    }
    if (writers.some((w) => typeof w !== "string" || w === "" || NUMERIC_ID.test(w)) || new Set(writers).size !== writers.length) {
      // an integer-like key is moved to the front of a JS object whatever the insertion order: the key ORDER of the reference's merged clocks
      // (part of their identity: src/bullet-crt.js:200-203) would not be the insertion order the device tracks
      const err = new Error("bmx: vector-clock mode needs distinct, non-numeric string writer ids");
      err.code = "BMX_BAD_WRITERS";
      throw err;
    }
    this.writers = writers.slice();                 // grows up to K when opts.maxWriters leaves room (WriterIndex)
    this.K = maxW || writers.length;
    this.local = writers.indexOf(local);
    this.writerIndex = new WriterIndex(this);
    writers.forEach((w, k) => this.writerIndex.set(w, k));
    const devs = Array.isArray(opts.devices) && opts.devices.length ? opts.devices.slice() : new Array(Math.max(1, opts.shards | 0)).fill(opts.device || 0);
    const cap = opts.vcCapacityRows || opts.capacityRows || (1 << 16);
    this.handles = devs.map((d) => this.native.vcCreate(d, Math.max(1024, Math.ceil(cap / devs.length)), this.K, this.local));
    this.handle = this.handles[0];
    this.N = devs.length;
    this.keys = new KeyDictionary();
  }
  /* rows of shard g (indices into the caller's columns), in batch order */
  _split(id) {
    const owners = this.native.ownersOf(id, this.N);
    const counts = new Uint32Array(this.N);
    for (let i = 0; i < owners.length; i++) counts[owners[i]]++;
    const back = []; for (let g = 0; g < this.N; g++) back.push(new Uint32Array(counts[g]));
    counts.fill(0);
    for (let i = 0; i < owners.length; i++) { const g = owners[i]; back[g][counts[g]++] = i; }
    return back;
  }
  _sub(c, idx) {
    const m = idx.length, K = this.K;
    const id = new BigUint64Array(m), field = new Uint32Array(m), clocks = new Uint32Array(m * K), val = new BigInt64Array(m);
    const keysets = c.keysets ? new Uint32Array(m) : undefined;
    for (let x = 0; x < m; x++) {
      const j = idx[x];
      id[x] = c.id[j]; field[x] = c.field[j]; val[x] = c.val[j];
      if (keysets) keysets[x] = c.keysets[j];
      for (let k = 0; k < K; k++) clocks[x * K + k] = c.clocks[j * K + k];
    }
    return { id, field, clocks, val, keysets };
  }
  /* c.keysets (optional Uint32Array): which writers each clock names and in which order; without it every clock names all K, in order */
  loadRows(c) {
    if (this.N === 1) { this.native.vcLoadRows(this.handle, c.id, c.field, c.clocks, c.val, c.keysets); return; }
    const back = this._split(c.id);
    for (let g = 0; g < this.N; g++) if (back[g].length) { const s = this._sub(c, back[g]); this.native.vcLoadRows(this.handles[g], s.id, s.field, s.clocks, s.val, s.keysets); }
  }
  mergeBatch(c) {
    if (this.N === 1) return this.native.vcMergeBatch(this.handle, c.id, c.field, c.clocks, c.val, c.keysets);
    const n = c.id.length, back = this._split(c.id);
    const flags = new Uint8Array(n), upd = [];
    let nRows = 0;
    for (let g = 0; g < this.N; g++) {
      if (!back[g].length) { nRows += this.native.vcRowCount(this.handles[g]); continue; }
      const s = this._sub(c, back[g]);
      const r = this.native.vcMergeBatch(this.handles[g], s.id, s.field, s.clocks, s.val, s.keysets);
      for (let x = 0; x < r.flags.length; x++) flags[back[g][x]] = r.flags[x];
      for (let x = 0; x < r.updated.length; x++) upd.push(back[g][r.updated[x]]);
      nRows += r.nRows;
    }
    return { updated: Uint32Array.from(upd).sort(), flags, nRows };
  }
  /* mergeBatch off the event loop (one table: a worker thread of the addon, in issue order with every other operation on the table; the clocks of the
   * updated rows come back with the result as `rows`, read right behind the merge). Several tables: the synchronous form behind a resolved promise. */
  mergeBatchAsync(c) {
    if (this.N === 1 && typeof this.native.vcMergeBatchAsync === "function") return this.native.vcMergeBatchAsync(this.handle, c.id, c.field, c.clocks, c.val, c.keysets);
    return Promise.resolve().then(() => this.mergeBatch(c));
  }
  getRows(id, field) {
    if (this.N === 1) return this.native.vcGetRows(this.handle, id, field);
    const n = id.length, K = this.K, back = this._split(id);
    const clocks = new Uint32Array(n * K), val = new BigInt64Array(n), state = new Uint8Array(n), keysets = new Uint32Array(n);
    for (let g = 0; g < this.N; g++) {
      const idx = back[g], m = idx.length;
      if (!m) continue;
      const gi = new BigUint64Array(m), gf = new Uint32Array(m);
      for (let x = 0; x < m; x++) { gi[x] = id[idx[x]]; gf[x] = field[idx[x]]; }
      const r = this.native.vcGetRows(this.handles[g], gi, gf);
      for (let x = 0; x < m; x++) { const j = idx[x]; val[j] = r.val[x]; state[j] = r.state[x]; keysets[j] = r.keysets[x]; for (let k = 0; k < K; k++) clocks[j * K + k] = r.clocks[x * K + k]; }
    }
    return { clocks, val, state, keysets };
  }
  rowCount() { let t = 0; for (const h of this.handles) t += this.native.vcRowCount(h); return t; }
  /* node ids (BigUint64Array) of the rows of `field` with lo <= value <= hi: range()/equals() over the K-writer rows (every table scans its own) */
  scanRange(field, lo, hi) {
    if (this.N === 1) return this.native.vcScanRange(this.handle, field, lo, hi);
    const parts = this.handles.map((h) => this.native.vcScanRange(h, field, lo, hi));
    const out = new BigUint64Array(parts.reduce((a, p) => a + p.length, 0));
    let o = 0; for (const p of parts) { out.set(p, o); o += p.length; }
    return out;
  }
  close() {
    for (const h of this.handles || []) this.native.vcDestroy(h);
    this.handles = []; this.handle = null;
  }
}

module.exports = DeviceGraph;
module.exports.DeviceVcTable = DeviceVcTable;
module.exports.WriterIndex = WriterIndex;
