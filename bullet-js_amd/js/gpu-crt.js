"use strict";
/*
 * gpu-crt.js — GpuCRT: drop-in for the reference's conflict resolver behind `bullet.crt`
 * (plug point: `new Bullet({disableCRT: true}); bullet.crt = new GpuCRT(bullet)`, SURVEY §8(b)).
 *
 * Interface mirrored (same names, arguments, return shapes and error behaviour as the reference's class in
 * src/bullet-crt.js): setCompare :23, createVectorClock :33, getVectorClock :44, incrementVectorClock :56,
 * compareVectorClocks :68, mergeVectorClocks :103, mergeValues :122, resolve :164, createUpdate :287,
 * processUpdate :304, handleUpdate :329 (the only method Bullet.setData calls, src/bullet.js:141-142), formatClock :392.
 *
 * Two paths:
 *   - single operations (one put / one network message, arbitrary JS values, general vector clocks) are decided
 *     here on the host: a PCIe round trip per put would be pointless, and strings/objects cannot live on the GPU;
 *   - batches (sync chunks, bulk loads: src/bullet-network-sync.js:551-569) go to the MI355X through
 *     mergeBatch()/mergeEntries(): typed columns -> bmx_merge_batch. That path has no host implementation.
 */
const { Columns, VcColumns, fieldId, isDeviceInt, scalarClock, denseClock } = require("./hash");

const REASON = {
  fresh: "no current state",
  same: "identical clocks and values",
  byValue: "identical clocks, decided by value comparison",
  newer: "incoming vector clock dominates",
  older: "current vector clock dominates (incoming is historical)",
  forked: "concurrent modifications, merged objects",
};

function threeWay(a, b) {
  if (a === b) return 0;
  return a < b ? -1 : 1;          // anything unordered (objects, NaN, null vs object) falls through to +1
}

function isMergeable(v) { return typeof v === "object" && v !== null && !Array.isArray(v); }

function growColumns(cols, rowEntry, rowField, used, cap) {
  const bigger = new Columns(cap);
  bigger.id.set(cols.id.subarray(0, used)); bigger.field.set(cols.field.subarray(0, used));
  bigger.ts.set(cols.ts.subarray(0, used)); bigger.val.set(cols.val.subarray(0, used));
  const re = new Int32Array(cap), rf = new Int32Array(cap);
  re.set(rowEntry.subarray(0, used)); rf.set(rowField.subarray(0, used));
  return { cols: bigger, rowEntry: re, rowField: rf, cap };
}

function verdict(winner, clock, value, reason, extra) {
  /* decision record; field set of the reference's resolve() results */
  return Object.assign({
    defer: false, historical: false, converge: true,
    incoming: winner === "in", current: winner === "cur", concurrent: false,
    vectorClock: clock, reason, value,
  }, extra || {});
}

class GpuCRT {
  /**
   * @param {object} bullet  the Bullet instance (needs .id, .meta, ._getData)
   * @param {object} [opts]  { graph: DeviceGraph (shared with GpuQuery), device, capacityRows, writer,
   *                           writers: [ids] — N4: batch rows carry a vector clock over exactly these (<= 8) writers,
   *                           this peer's id among them, and mergeEntries() uses the device's vector-clock table }
   */
  constructor(bullet, opts = {}) {
    this.bullet = bullet;
    this.vectorClocks = new Map();
    this.compare = threeWay;
    this._opts = opts;
    this._graph = opts.graph || null;
  }

  /* ---------------------------------------------------------------- clock bookkeeping (host) */
  setCompare(fn) { this.compare = fn; return this; }

  createVectorClock(key) {
    const c = {};
    c[this.bullet.id] = 1;
    this.vectorClocks.set(key, c);
    return c;
  }

  getVectorClock(key) {
    const c = this.vectorClocks.get(key);
    return c === undefined ? this.createVectorClock(key) : c;
  }

  incrementVectorClock(key) {
    const c = this.getVectorClock(key);
    const me = this.bullet.id;
    c[me] = (c[me] || 0) + 1;
    return c;                       // the live object, as the reference hands out (meta aliases it)
  }

  compareVectorClocks(a, b) {
    if (!a) return -1;
    if (!b) return 1;
    let aAhead = false, bAhead = false;
    const seen = new Set();
    for (const list of [Object.keys(a), Object.keys(b)]) {
      for (const node of list) {
        if (seen.has(node)) continue;
        seen.add(node);
        const x = a[node] || 0, y = b[node] || 0;
        if (x > y) aAhead = true; else if (y > x) bAhead = true;
        if (aAhead && bAhead) return 0;
      }
    }
    return aAhead ? 1 : (bAhead ? -1 : 0);
  }

  mergeVectorClocks(a, b) {
    if (!a) return Object.assign({}, b);
    if (!b) return Object.assign({}, a);
    const out = Object.assign({}, a);
    for (const node of Object.keys(b)) out[node] = Math.max(out[node] || 0, b[node]);
    return out;
  }

  mergeValues(incoming, current) {
    if (!isMergeable(incoming) || !isMergeable(current)) {
      return this.compare(incoming, current) >= 0 ? incoming : current;
    }
    const out = Object.assign({}, current);
    for (const k of Object.keys(incoming)) {
      out[k] = (k in out) ? this.mergeValues(incoming[k], out[k]) : incoming[k];
    }
    return out;
  }

  /* ---------------------------------------------------------------- one decision (host) */
  resolve(key, incomingClock, currentClock, incomingValue, currentValue) {
    if (!currentClock) {
      return verdict("in", this.incrementVectorClock(key), incomingValue, REASON.fresh);
    }
    const order = this.compareVectorClocks(incomingClock, currentClock);
    const merged = this.mergeVectorClocks(incomingClock, currentClock);
    this.vectorClocks.set(key, merged);

    if (order > 0) return verdict("in", merged, incomingValue, REASON.newer);
    if (order < 0) return verdict("cur", merged, currentValue, REASON.older, { historical: true });

    if (JSON.stringify(incomingClock) === JSON.stringify(currentClock)) {
      const c = this.compare(incomingValue, currentValue);
      if (c === 0) return verdict("none", merged, currentValue, REASON.same);
      return verdict(c > 0 ? "in" : "cur", merged, c > 0 ? incomingValue : currentValue, REASON.byValue);
    }
    return verdict("none", merged, this.mergeValues(incomingValue, currentValue), REASON.forked, { concurrent: true });
  }

  createUpdate(key, value) {
    return { value, vectorClock: Object.assign({}, this.incrementVectorClock(key)) };
  }

  processUpdate(key, incomingValue, incomingClock, currentValue, currentClock) {
    const decision = this.resolve(key, incomingClock, currentClock, incomingValue, currentValue);
    return { value: decision.value, vectorClock: decision.vectorClock, decision };
  }

  handleUpdate(path, incomingData, isFromNetwork = false) {
    const currentData = this.bullet._getData(path);
    const currentClock = (this.bullet.meta[path] || {}).vectorClock;

    let clock, payload = incomingData;
    const tagged = isFromNetwork && incomingData && typeof incomingData === "object" && incomingData.__vectorClock;
    if (tagged) {
      clock = incomingData.__vectorClock;
      if (Array.isArray(incomingData)) {
        payload = incomingData.slice();
      } else {
        payload = {};
        for (const k of Object.keys(incomingData)) if (k !== "__vectorClock") payload[k] = incomingData[k];
      }
    } else {
      clock = this.incrementVectorClock(path);      // local write (or untagged network primitive)
    }

    const d = this.resolve(path, clock, currentClock, payload, currentData);

    let broadcastData = d.value;
    if (typeof broadcastData === "object" && broadcastData !== null) {
      broadcastData = Array.isArray(broadcastData)
        ? broadcastData.concat([{ __vectorClock: d.vectorClock }])
        : Object.assign({}, broadcastData, { __vectorClock: d.vectorClock });
    }
    const doUpdate = d.incoming || !currentClock || d.concurrent;
    if (doUpdate) this._queueDeviceWrite(path, d.value, d.vectorClock);
    return {
      value: d.value,
      vectorClock: d.vectorClock,
      broadcastData,
      decision: d,
      doUpdate,
    };
  }

  /*
   * Write-through of single writes: once the device holds rows, a leaf that is written through setData (a local put, or a remote write that
   * took the host path) has to reach its row too, or the next batch would be resolved against a state the host has already left behind.
   * Only what the device can hold: the leaf `<node>/<field>` with a safe-integer value and the scalar clock {writer: ts}. Queued here, sent
   * as one LWW load in front of the next device operation (no device call per put).
   */
  _queueDeviceWrite(path, value, clock) {
    if (!this._graph || this._opts.writers) return;           // no device table in use (or the K-writer table: its rows are not scalar-clock rows)
    if (!isDeviceInt(value)) return;
    const ts = scalarClock(clock, this._opts.writer || this.bullet.id);
    if (ts < 0) return;
    const cut = path.lastIndexOf("/");
    if (cut <= 0) return;
    (this._pendingRows || (this._pendingRows = [])).push(path.slice(0, cut), path.slice(cut + 1), ts, value);
  }

  _flushDeviceWrites() {
    const q = this._pendingRows;
    if (!q || q.length === 0) return;
    this._pendingRows = null;
    const g = this.graph, keys = g.keys, n = q.length / 4;
    const cols = new Columns(n);
    for (let i = 0; i < n; i++) {
      const node = q[4 * i], c = node.lastIndexOf("/");
      cols.set(i, keys.idOf(node), keys.fieldOf(c < 0 ? "" : node.slice(0, c), q[4 * i + 1]), q[4 * i + 2], q[4 * i + 3]);
    }
    g.loadRows(cols);
  }

  formatClock(clock) {
    if (!clock) return "null";
    return Object.keys(clock).map((n) => n + ":" + clock[n]).join(", ");
  }

  /* ---------------------------------------------------------------- batch path (MI355X) */
  get graph() {
    if (!this._graph) {
      const DeviceGraph = require("./device-graph");
      this._graph = new DeviceGraph(this._opts);    // throws without the addon / a GPU
      this._ownsGraph = true;
    }
    return this._graph;
  }

  /**
   * Merge typed columns on the GPU.
   * cols: {id: BigUint64Array, field: Uint32Array, ts: BigInt64Array, val: BigInt64Array}
   * opts: {insertMode: 'reference'|'delta', uniqueKeys: bool, strictFlags: bool (exact sequential per-delta flags even with
   *        duplicate keys in the batch; about twice as slow)}
   * -> {applied: Uint32Array (ascending delta indices whose value is now stored), flags, nApplied, nConflicts, nRows}
   */
  mergeBatch(cols, opts = {}) {
    this._flushDeviceWrites();
    const g = this.graph;
    let mode = opts.insertMode === "delta" ? g.native.INSERT_DELTA : g.native.INSERT_REFERENCE;
    if (opts.uniqueKeys) mode |= g.native.MERGE_UNIQUE_KEYS;
    if (opts.strictFlags) mode |= g.native.MERGE_STRICT_FLAGS;
    return g.mergeBatch(cols, mode);
  }

  /** Same as mergeBatch but off the event loop: resolves to the same result object (the reference API is synchronous;
   *  this is the Promise variant SURVEY §8(b) calls for). The columns must not be mutated until it settles. */
  mergeBatchAsync(cols, opts = {}) {
    this._flushDeviceWrites();
    const g = this.graph;
    let mode = opts.insertMode === "delta" ? g.native.INSERT_DELTA : g.native.INSERT_REFERENCE;
    if (opts.uniqueKeys) mode |= g.native.MERGE_UNIQUE_KEYS;
    if (opts.strictFlags) mode |= g.native.MERGE_STRICT_FLAGS;
    return g.mergeBatchAsync(cols, mode);
  }

  /**
   * Batch adapter for sync chunks (reference loop: src/bullet-network-sync.js:551-569).
   * entries: [{path, data, vectorClock}] where data is an integer or an object of integer fields and
   * vectorClock is {<writer>: ts}. Each (node, field) becomes one device row. Entries outside that contract
   * (strings, nested objects, multi-writer clocks) are returned in `host` for the caller to pass to setData().
   * opts.apply: true = update the facade's store/meta/log/listeners once for the batch (N1), "each" = one _applyUpdate per winner.
   * -> {applied: [{entry, field}], nConflicts, host: [entry indices], broadcast: [{path, broadcastData}] when applied}
   */
  mergeEntries(entries, opts = {}) {
    if (this._opts.writers) return this._mergeEntriesVector(entries, opts);
    const writer = opts.writer || this.bullet.id;
    const g = this.graph;
    const keys = g.keys;
    // one pass: eligible entries are written straight into growable typed columns (a second pass over a million JS objects costs
    // more than the GPU merge); rowEntry / rowField map device rows back to (entry, field name)
    let cap = Math.max(16, entries.length * 2);
    let cols = new Columns(cap);
    let rowEntry = new Int32Array(cap), rowField = new Int32Array(cap);
    const fieldNames = [], fieldIndex = new Map();          // names seen in this call, -1 = the node's own value
    const host = [];
    let i = 0;
    let lastParent = "";                                    // consecutive entries usually share their collection: reuse the sliced string
    const parentOf = (path, cut) => {
      if (cut < 0) return "";
      if (cut === lastParent.length && path.startsWith(lastParent)) return lastParent;
      lastParent = path.slice(0, cut);
      return lastParent;
    };
    for (let ei = 0; ei < entries.length; ei++) {
      const e = entries[ei];
      const ts = scalarClock(e.vectorClock, writer);
      const d = e.data;
      if (ts < 0) { host.push(ei); continue; }
      const first = i;
      let ok = true;
      if (isDeviceInt(d)) {
        if (i + 1 > cap) ({ cols, rowEntry, rowField, cap } = growColumns(cols, rowEntry, rowField, i, cap * 2));
        keys.lookup(e.path);
        cols.set2(i, keys.lo, keys.hi, keys.fieldOf(parentOf(e.path, e.path.lastIndexOf("/")), null), ts, d);
        rowEntry[i] = ei; rowField[i] = -1; i++;
      } else if (d && typeof d === "object" && !Array.isArray(d)) {
        const parent = parentOf(e.path, e.path.lastIndexOf("/"));
        let idLo = 0, idHi = 0, haveId = false;
        for (const k in d) {
          if (k === "__vectorClock" || k === "__fromNetwork" || !Object.prototype.hasOwnProperty.call(d, k)) continue;
          const v = d[k];
          if (!isDeviceInt(v)) { ok = false; break; }
          if (!haveId) { keys.lookup(e.path); idLo = keys.lo; idHi = keys.hi; haveId = true; }
          if (i + 1 > cap) ({ cols, rowEntry, rowField, cap } = growColumns(cols, rowEntry, rowField, i, cap * 2));
          let fi = fieldIndex.get(k);
          if (fi === undefined) { fi = fieldNames.length; fieldNames.push(k); fieldIndex.set(k, fi); }
          cols.set2(i, idLo, idHi, keys.fieldOf(parent, k), ts, v);
          rowEntry[i] = ei; rowField[i] = fi; i++;
        }
        if (ok && i === first) ok = false;                   // an object without fields: nothing for the device
      } else ok = false;
      if (!ok) { i = first; host.push(ei); }                 // roll back the rows of an entry that turned out to be off-contract
    }
    const used = cols.slice(i);
    const r = this.mergeBatch(used, opts);
    const applied = new Array(r.applied.length);
    for (let k = 0; k < applied.length; k++) { const j = r.applied[k]; applied[k] = { entry: rowEntry[j], field: rowField[j] < 0 ? null : fieldNames[rowField[j]] }; }
    const broadcast = opts.apply ? this._applyWinners(entries, used, r.applied, applied, opts.apply, opts.broadcast !== false) : undefined;
    if (opts.apply) this._notifyIndexHook(entries, host);
    return { applied, nApplied: r.nApplied, nConflicts: r.nConflicts, nRows: r.nRows, host, broadcast };
  }

  /*
   * The reference's query engine keeps its indices current through a wrapper around setData (src/bullet-query.js:13-21): after EVERY
   * write, accepted or not, it calls _updateIndices(path, data). A batch that is applied here never passes through setData, so the
   * hook is called for it: GpuQuery marks the touched children of its indexed collections (it re-reads them from the store at the next
   * query), a reference BulletQuery gets its own _updateIndices. Entries the batch handed back (`host`) reach the hook through setData.
   */
  _notifyIndexHook(entries, hostIdx) {
    const q = this.bullet.query;
    if (!q) return;
    const skip = hostIdx && hostIdx.length ? new Set(hostIdx) : null;
    if (typeof q._touch === "function") {
      if (!q.indexedPaths || q.indexedPaths.size === 0) return;
      for (let i = 0; i < entries.length; i++) if (!skip || !skip.has(i)) q._touch(entries[i].path);
    } else if (typeof q._updateIndices === "function") {
      if (!q.indexedPaths || q.indexedPaths.size === 0) return;
      for (let i = 0; i < entries.length; i++) if (!skip || !skip.has(i)) q._updateIndices(entries[i].path, entries[i].data);
    }
  }

  /*
   * N1 (SURVEY §8(f)): hand the batch's final winners to the facade in ONE pass. Each winning row is the leaf
   * `<entry.path>[/<field>]`; its stored clock and value are read back from the device (an inserted row's clock is
   * {writer: 2}, not the incoming one). opts.apply === true: store, meta, op log and listeners are updated once for the whole
   * batch (batch-apply.js: ring-style log, ancestor listeners de-duplicated; a facade may supply its own `_applyBatch`);
   * opts.apply === "each": the facade's per-write `_applyUpdate` is called once per winner (src/bullet.js:184-266), the
   * reference's own cost per write. Returns what setData would have broadcast per winner (src/bullet-crt.js:371-376).
   */
  _applyWinners(entries, cols, appliedIdx, applied, mode, wantBroadcast = true) {
    const n = appliedIdx.length;
    if (n === 0) return [];
    const ids = new BigUint64Array(n), id32 = new Uint32Array(ids.buffer), fields = new Uint32Array(n);
    const src32 = cols._id32 || new Uint32Array(cols.id.buffer, cols.id.byteOffset, cols.id.length * 2);
    for (let k = 0; k < n; k++) { const j = appliedIdx[k]; id32[2 * k] = src32[2 * j]; id32[2 * k + 1] = src32[2 * j + 1]; fields[k] = cols.field[j]; }
    const rows = this.graph.getRows(ids, fields);
    const ts32 = new Uint32Array(rows.ts.buffer, rows.ts.byteOffset, n * 2), val32 = new Int32Array(rows.val.buffer, rows.val.byteOffset, n * 2);
    const valLo = new Uint32Array(rows.val.buffer, rows.val.byteOffset, n * 2);
    const writer = this._opts.writer || this.bullet.id;
    const updates = new Array(n);
    for (let k = 0; k < n; k++) {
      const a = applied[k];
      const e = entries[a.entry];
      const leaf = a.field === null ? e.path : e.path + "/" + a.field;
      const clock = {};
      clock[writer] = ts32[2 * k + 1] * 4294967296 + ts32[2 * k];                        // 0 <= ts <= 2^53-1
      updates[k] = { path: leaf, value: val32[2 * k + 1] * 4294967296 + valLo[2 * k], vectorClock: clock };   // signed high half, unsigned low half
      this.vectorClocks.set(leaf, clock);
    }
    const b = this.bullet;
    if (mode === "each") {
      if (typeof b._applyUpdate === "function") for (const u of updates) b._applyUpdate(u.path, u.value, u.vectorClock, true);
      return wantBroadcast ? updates.map((u) => ({ path: u.path, broadcastData: u.value })) : [];
    }
    if (typeof b._applyBatch === "function") return b._applyBatch(updates, true) || [];
    return require("./batch-apply").applyBatch(b, updates, true, wantBroadcast);
  }

  /* ---------------------------------------------------------------- N4: K-writer vector clocks on the device */
  get vcTable() {
    if (!this._vc) {
      const { DeviceVcTable } = require("./device-graph");
      this._vc = new DeviceVcTable(this._opts.writers, this.bullet.id, this._opts);   // throws without the addon / a GPU
    }
    return this._vc;
  }

  /* stored clock of a device row as the reference would hold it: {local: n} after a first write, all K writers otherwise */
  _clockObject(comps, off, state) {
    const t = this.vcTable, c = {};
    if (state === t.native.VC_SPARSE) { c[t.writers[t.local]] = comps[off + t.local]; return c; }
    for (let k = 0; k < t.K; k++) c[t.writers[k]] = comps[off + k];
    return c;
  }

  /**
   * mergeEntries() when the resolver was created with opts.writers (general vector clocks, SURVEY §8(f) N4).
   * An entry goes to the device when its clock has exactly those writers as keys, in that order, with uint32 counters, and
   * its data is an integer or an object of integer fields; everything else is returned in `host`.
   * Device semantics = resolve() applied delta by delta in entry order (src/bullet-crt.js:164-279).
   * -> {applied: [{entry, field}] (last updating delta of every row that changed, in entry order), flags: Uint8Array per
   *     device row (1 incoming, 2 current, 4 historical, 8 concurrent), rows: [{entry, field}] per device row,
   *     nApplied, nConflicts (concurrent merges), nRows, host}
   */
  _mergeEntriesVector(entries, opts = {}) {
    const t = this.vcTable;
    const plan = [];
    let rows = 0;
    for (const e of entries) {
      const comps = denseClock(e.vectorClock, t.writers);
      let fields = null;
      if (comps) {
        if (isDeviceInt(e.data)) fields = [[null, e.data]];
        else if (e.data && typeof e.data === "object" && !Array.isArray(e.data)) {
          fields = [];
          for (const k of Object.keys(e.data)) {
            if (k === "__vectorClock" || k === "__fromNetwork") continue;
            if (!isDeviceInt(e.data[k])) { fields = null; break; }
            fields.push([k, e.data[k]]);
          }
        }
      }
      plan.push(fields && fields.length ? { comps, fields } : null);
      if (fields) rows += fields.length;
    }
    const cols = new VcColumns(rows, t.K);
    const back = new Array(rows);
    const host = [];
    let i = 0;
    entries.forEach((e, ei) => {
      const p = plan[ei];
      if (!p) { host.push(ei); return; }
      const cut = e.path.lastIndexOf("/");
      const parent = cut < 0 ? "" : e.path.slice(0, cut);
      const id = t.keys.idOf(e.path);
      for (const [fname, v] of p.fields) {
        cols.set(i, id, t.keys.fieldOf(parent, fname), p.comps, v);
        back[i] = { entry: ei, field: fname };
        i++;
      }
    });
    const r = t.mergeBatch(cols);
    const applied = Array.from(r.updated, (j) => back[j]);
    let nConflicts = 0;
    for (let j = 0; j < r.flags.length; j++) if (r.flags[j] & t.native.FLAG_CONCURRENT) nConflicts++;
    if (opts.apply && r.updated.length) {
      const n = r.updated.length;
      const ids = new BigUint64Array(n), fields = new Uint32Array(n);
      for (let k = 0; k < n; k++) { ids[k] = cols.id[r.updated[k]]; fields[k] = cols.field[r.updated[k]]; }
      const got = t.getRows(ids, fields);
      for (let k = 0; k < n; k++) {
        const a = applied[k], e = entries[a.entry];
        const leaf = a.field === null ? e.path : e.path + "/" + a.field;
        const clock = this._clockObject(got.clocks, k * t.K, got.state[k]);
        if (typeof this.bullet._applyUpdate === "function") this.bullet._applyUpdate(leaf, Number(got.val[k]), clock, true);
        this.vectorClocks.set(leaf, clock);
      }
    }
    return { applied, flags: r.flags, rows: back, nApplied: applied.length, nConflicts, nRows: r.nRows, host };
  }

  /** Stored (value, clock) of device rows in vector mode: [{path, field}] -> [{value, vectorClock} | null]. */
  vcLookup(keys) {
    const t = this.vcTable, n = keys.length;
    const ids = new BigUint64Array(n), id32 = new Uint32Array(ids.buffer), fields = new Uint32Array(n);
    keys.forEach((k, i) => {
      const cut = k.path.lastIndexOf("/");
      const id = t.keys.idOf(k.path);
      id32[2 * i] = id[0]; id32[2 * i + 1] = id[1];
      fields[i] = t.keys.fieldOf(cut < 0 ? "" : k.path.slice(0, cut), k.field === undefined ? null : k.field);
    });
    const got = t.getRows(ids, fields);
    return keys.map((_, i) => (got.state[i] === t.native.VC_ABSENT ? null
      : { value: Number(got.val[i]), vectorClock: this._clockObject(got.clocks, i * t.K, got.state[i]) }));
  }

  /*
   * N3 (SURVEY §8(f)): checkpoint of the device rows in host terms — [{path, collection, field, ts, val}] — and its
   * inverse. The shape matches what the reference persists per path (value + vectorClock: src/bullet-file-storage.js:170-210).
   */
  checkpoint() {
    this._flushDeviceWrites();
    const g = this.graph;
    const d = g.dumpRows();
    const id32 = new Uint32Array(d.id.buffer, d.id.byteOffset, d.id.length * 2);
    const out = [];
    for (let i = 0; i < d.id.length; i++) {
      const path = g.keys.pathOf(id32[2 * i], id32[2 * i + 1]);
      const f = g.keys.fields.get(d.field[i]);
      out.push({ path: path === undefined ? null : path, id: d.id[i].toString(16), fieldHash: d.field[i],
        collection: f ? f[0] : null, field: f ? f[1] : null, ts: Number(d.ts[i]), val: Number(d.val[i]) });
    }
    return out;
  }

  restore(rows) {
    const g = this.graph;
    const cols = new Columns(rows.length);
    rows.forEach((r, i) => {
      const id = r.path !== null && r.path !== undefined ? g.keys.idOf(r.path) : [Number(BigInt("0x" + r.id) & 0xffffffffn), Number(BigInt("0x" + r.id) >> 32n)];
      const f = r.collection !== null && r.collection !== undefined ? g.keys.fieldOf(r.collection, r.field) : r.fieldHash;
      cols.set(i, id, f, r.ts, r.val);
    });
    g.loadRows(cols);
    return rows.length;
  }

  close() {
    if (this._ownsGraph && this._graph) this._graph.close();
    this._graph = null;
    if (this._vc) { this._vc.close(); this._vc = null; }
  }
}

module.exports = GpuCRT;
