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
const { Columns, VcColumns, fieldId, isDeviceInt, scalarClock, clockKeyset, keysetWriters, NODE_CLOCK, VAL_DELETED } = require("./hash");

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

/* value at `path` without the facade's autovivification (src/bullet.js:115-129 creates {} on the way): undefined when absent */
function peek(root, path) {
  let cur = root;
  if (!path) return cur;
  let from = 0;
  for (;;) {
    const cut = path.indexOf("/", from);
    const seg = cut < 0 ? path.slice(from) : path.slice(from, cut);
    if (seg) {
      if (cur === null || typeof cur !== "object" || !Object.prototype.hasOwnProperty.call(cur, seg)) return undefined;
      cur = cur[seg];
    }
    if (cut < 0) return cur;
    from = cut + 1;
  }
}

/* Rows for bmx_put_rows, kept as SEGMENTS that are sent in order, one call each (a call wants unique keys). Single host writes go to a segment that
 * keeps the LAST write per key (a Map keyed by the row key: fine for a put at a time); the winners of a batch are unique by construction — one
 * winner per node, one row per field — and are appended to a segment of their own with no look-up at all. */
class PutQueue {
  /* graph: where segments take their columns from and give them back to (DeviceGraph.takeColumns: page-locked, reused); plain Columns without one */
  constructor(graph) { this.segs = []; this.n = 0; this.hostRows = 0; this._sealed = false; this._graph = graph || null; }
  _alloc(n) { return this._graph ? this._graph.takeColumns(n) : new Columns(n); }
  _seg(dedupe, room) {
    let sg = this.segs.length ? this.segs[this.segs.length - 1] : null;
    if (!sg || sg.dedupe !== dedupe || (this._sealed && !dedupe)) {
      const cols = this._alloc(Math.max(64, room));
      sg = { dedupe, n: 0, cap: cols.n, cols, at: dedupe ? new Map() : null };
      this.segs.push(sg);
      this._sealed = false;
    }
    if (sg.n + room > sg.cap) {
      let cap = sg.cap; while (sg.n + room > cap) cap *= 2;
      const bigger = this._alloc(cap), m = sg.n;
      bigger.id.set(sg.cols.id.subarray(0, m)); bigger.field.set(sg.cols.field.subarray(0, m)); bigger.ts.set(sg.cols.ts.subarray(0, m)); bigger.val.set(sg.cols.val.subarray(0, m));
      if (this._graph) this._graph.giveColumns(sg.cols);
      sg.cols = bigger; sg.cap = bigger.n;
    }
    return sg;
  }
  push(lo, hi, field, ts, val) {                      // a single host write: the last one per key stays
    const sg = this._seg(true, 1);
    const key = lo + ":" + hi + ":" + field;
    let i = sg.at.get(key);
    if (i === undefined) { i = sg.n++; sg.at.set(key, i); this.n++; this.hostRows++; }
    sg.cols.set2(i, lo, hi, field, ts, val);
  }
  pushUnique(lo, hi, field, ts, val) {                // a batch winner's row: unique within its batch by construction
    const sg = this._seg(false, 1);
    sg.cols.set2(sg.n++, lo, hi, field, ts, val);
    this.n++;
  }
  /* the segment a batch's winners append to, with room for `rows` more rows (one allocation per batch instead of doubling from 64) */
  uniqueSegment(rows) { return this._seg(false, rows); }
  closeBatch() { this._sealed = true; }               // the next batch's rows may name the same keys: they start a segment of their own
  /* rows of single host writes are waiting (they may be clock rows: a merge must see them); batch winners' value rows alone can wait for a reader */
  hasHostRows() { return this.hostRows > 0; }
  take() {
    const out = [];
    for (const sg of this.segs) { if (sg.n > 0) out.push(sg.cols.slice(sg.n)); else if (this._graph) this._graph.giveColumns(sg.cols); }
    this.segs = []; this.n = 0; this.hostRows = 0; this._sealed = false;
    return out;
  }
}

function verdict(winner, clock, value, reason, extra) {
  /* decision record; field set of the reference's resolve() results */
  return Object.assign({
    defer: false, historical: false, converge: true,
    incoming: winner === "in", current: winner === "cur", concurrent: false,
    vectorClock: clock, reason, value,
  }, extra || {});
}

const KIND_NODE = 1, KIND_INT = 2, KIND_MASK = 3;
// more bits of GpuCRT._kind[path number] (the store-kept batch path, see ClockMap / _unaliasLosers):
const P_FALSY = 4;        // the path holds a falsy value (null of a deletion, 0, "", false): _getData replaces it by {} when it is read (src/bullet.js:115-129)
const P_CLOCK_COPY = 8;   // crt.vectorClocks[path] is a fresh COPY of meta[path].vectorClock (the path's last batch entry LOST), not materialised yet
const P_CLOCK_META = 16;  // crt.vectorClocks[path] IS meta[path].vectorClock (a batch winner), not materialised yet
const P_CLOCK_LAZY = P_CLOCK_COPY | P_CLOCK_META;

/* crt.vectorClocks (src/bullet-crt.js:8): a Map path -> clock object. The store-kept batch path would pay one insertion into this million-entry,
 * string-keyed table per winner and per loser (0.7 us each, cache-cold) for values nobody reads until a LOCAL write touches that very path — so a
 * batch only marks, in a typed array indexed by the path's dictionary number, what the entry WOULD be (P_CLOCK_META: the object meta[path] holds;
 * P_CLOCK_COPY: a fresh copy of it), and get()/has() materialise that before they answer. set() and delete() drop the mark: what is stored
 * explicitly is newer. Observable state through get / has / set is the reference's; size and iteration see only what was materialised. */
class ClockMap extends Map {
  constructor(crt) { super(); this._crt = crt; }
  get(key) { const c = this._crt; if (c !== undefined && c._nLazy > 0) c._materializeClock(key, this); return super.get(key); }
  has(key) { const c = this._crt; if (c !== undefined && c._nLazy > 0) c._materializeClock(key, this); return super.has(key); }
  set(key, v) { const c = this._crt; if (c !== undefined && c._nLazy > 0) c._dropLazyClock(key); return super.set(key, v); }
  delete(key) { const c = this._crt; if (c !== undefined && c._nLazy > 0) c._dropLazyClock(key); return super.delete(key); }
  _put(key, v) { return super.set(key, v); }
}

const PACK_BLOCK = 1024;   // entries whose dictionary probes are issued back to back (GpuCRT._packEntries); measured 16..4096: flat from 1024 on

class GpuCRT {
  /**
   * @param {object} bullet  the Bullet instance (needs .id, .meta, ._getData)
   * @param {object} [opts]  { graph: DeviceGraph (shared with GpuQuery), device, capacityRows, writer,
   *                           writers: [ids] — N4: clocks may name any of these (<= 8) writers, this peer's id among them, in any order
   *                           (general vector clocks): the nodes' clocks live in the device's vector-clock table;
   *                           writers: "auto" (= [bullet.id] + maxWriters: 8), or maxWriters: n next to a list — the table is n writers wide and a
   *                           writer nobody named before takes a free component when a clock first names it (a mesh learns its peers as it
   *                           goes); from the (n + 1)-th distinct writer on, clocks naming it keep their paths on the host: hostOnlyInfo() }
   */
  constructor(bullet, opts = {}) {
    this.bullet = bullet;
    this.vectorClocks = new ClockMap(this);
    this._nLazy = 0;                // paths whose vectorClocks entry is only marked (P_CLOCK_*)
    this._nFalsy = 0;               // paths known to hold a falsy value (P_FALSY)
    this.compare = threeWay;
    if (opts.writers === "auto") opts = Object.assign({}, opts, { writers: [bullet.id], maxWriters: opts.maxWriters || 8 });
    this._opts = opts;
    this._graph = opts.graph || null;
    this._apiClocks = new Set();    // paths whose clock the public helpers touched before any write gave them a meta entry (entryEligible)
    this._hostOnly = new Set();     // paths whose clock the device cannot hold (more than one writer, foreign key sets): their entries stay on the host
    this._nodeSeq = 1;              // arrival number of node writes (val of the clock rows): a tie on the clock goes to the later write
    // What a path HOLDS decides what its clock row's val means (see _kindAllows): KIND_NODE — an object, or anything compare() treats like one
    // against an integer (null, booleans, arrays, floats, the 0 that _getData turns into {}): val = arrival number; KIND_INT — a non-zero safe
    // integer: val = the integer itself, so that identical clocks are decided BY VALUE exactly as resolve() does (src/bullet-crt.js:200-233).
    this._kind = new Uint8Array(1024);   // by the path's number in the graph's key dictionary; 0 = nothing known (no entry and no host write seen)
    // opts.integerEntries === false: integer entries are never the device's (they come back in `host`) and every path is a node path. This is what the
    // sync adapter sets (attach(..., {batchSync})): a primitive of a sync chunk is a LOCAL write there (src/bullet-network-sync.js:560-563), so no integer
    // ever needs the device — and an object entry never has to leave it because its path happens to hold an integer.
    this._intEntries = opts.integerEntries !== false;
    this._nIntPaths = 0;
    this.hostOnlyPaths = 0;         // how often a path has left the device path for the host's (monotone; hostOnlyInfo() has the current number)
    this._puts = null;              // PutQueue: rows the host decided, not yet on the device
    this._vcPuts = [];              // writers mode: clock rows of host writes for the vector-clock table [path, parent, keyset, counters...], not yet on the device
  }

  /* Kinds of paths (KIND_*). An INTEGER entry under an identical clock is decided by value against a stored integer — larger wins, equal is a no-op
   * (src/bullet-crt.js:200-233) — but against a stored OBJECT compare() answers +1 whatever the values (:11-15), and an object entry beats a
   * stored integer the same way. One row cannot order both ways, so a path is one kind at a time: integer entries are device deltas on integer
   * paths (val = the integer), object entries on node paths (val = arrival number); an entry of the other kind goes to `host`, whose write flips
   * the kind (_mirrorWrite). A path nobody has written yet takes the kind of its first entry. */
  _kindAt(idx) { return idx < this._kind.length ? this._kind[idx] & KIND_MASK : 0; }
  _kindRoom(idx) { if (idx >= this._kind.length) { const g = new Uint8Array(Math.max(2 * this._kind.length, idx + 1024)); g.set(this._kind); this._kind = g; } }
  _setKind(idx, k) {
    this._kindRoom(idx);
    const b = this._kind[idx], was = b & KIND_MASK;
    if (was === k) return;
    if (k === KIND_INT) this._nIntPaths++; else if (was === KIND_INT) this._nIntPaths--;
    this._kind[idx] = (b & ~KIND_MASK) | k;
  }
  _setFalsy(idx, falsy) {
    this._kindRoom(idx);
    const b = this._kind[idx];
    if (falsy) { if (!(b & P_FALSY)) { this._kind[idx] = b | P_FALSY; this._nFalsy++; } }
    else if (b & P_FALSY) { this._kind[idx] = b & ~P_FALSY; this._nFalsy--; }
  }
  /* ClockMap: what a batch left marked for `path` becomes a real entry (meta[path]'s clock object, or a copy of it) */
  _materializeClock(path, map) {
    const g = this._graph;
    if (!g) return;
    const idx = g.keys.find(path);
    if (idx < 0 || idx >= this._kind.length) return;
    const b = this._kind[idx];
    if (!(b & P_CLOCK_LAZY)) return;
    this._kind[idx] = b & ~P_CLOCK_LAZY; this._nLazy--;
    const m = (this.bullet.meta || {})[path];
    if (m && m.vectorClock) map._put(path, (b & P_CLOCK_META) ? m.vectorClock : Object.assign({}, m.vectorClock));
  }
  _dropLazyClock(path) {
    const g = this._graph;
    if (!g) return;
    const idx = g.keys.find(path);
    if (idx >= 0 && idx < this._kind.length && (this._kind[idx] & P_CLOCK_LAZY)) { this._kind[idx] &= ~P_CLOCK_LAZY; this._nLazy--; }
  }
  _markLazyClock(idx, bit) {
    this._kindRoom(idx);
    const b = this._kind[idx];
    if (!(b & P_CLOCK_LAZY)) this._nLazy++;
    this._kind[idx] = (b & ~P_CLOCK_LAZY) | bit;
  }
  _markHostOnly(path) { if (!this._hostOnly.has(path)) { this._hostOnly.add(path); this.hostOnlyPaths++; } }
  _clearHostOnly(path) { this._hostOnly.delete(path); }
  /** How much of the graph the device cannot resolve at the moment: paths whose clock names a writer outside the device table (more than 8 peers
   *  in a gossip mesh, src/bullet-network.js:404-418), or that hold a string; entries on them are resolved by the host path, one by one. */
  hostOnlyInfo() {
    const o = { hostOnlyPaths: this._hostOnly.size, marked: this.hostOnlyPaths, integerPaths: this._nIntPaths };
    if (this._opts.writers) { o.writers = this._vc ? this._vc.writers.slice() : this._opts.writers.slice(); o.writerSlots = this._vc ? this._vc.K : (this._opts.maxWriters || this._opts.writers.length); }
    return o;
  }

  /* ---------------------------------------------------------------- clock bookkeeping (host) */
  setCompare(fn) { this.compare = fn; return this; }

  createVectorClock(key) {
    if (!this._inUpdate) this._apiClocks.add(key);
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
    this._inUpdate = true;
    try { return this._handleUpdate(path, incomingData, isFromNetwork); } finally { this._inUpdate = false; }
  }

  _handleUpdate(path, incomingData, isFromNetwork) {
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
    if (doUpdate) this._mirrorWrite(path, currentData, d.value, d.vectorClock);
    // a local write that is REFUSED has still incremented the stored clock: the clock it incremented was the very object meta[path] holds (that is why it met
    // "identical clocks" and was decided by value, SURVEY §5) — the device's clock row follows
    else if (!tagged && currentClock && clock === currentClock) this._mirrorWrite(path, currentData, currentData, currentClock);
    return {
      value: d.value,
      vectorClock: d.vectorClock,
      broadcastData,
      decision: d,
      doUpdate,
    };
  }

  /*
   * Write-through of writes the HOST resolved (single puts, values or clocks outside the device contract, deletions): once the device holds
   * rows, its table has to follow, or the next batch would be resolved against a state the host has already left behind, and device-side
   * indexes would keep rows of nodes that are gone (src/bullet-query.js:83-85 skips null). What is mirrored for a write of `value` at `path`
   * (parent = its collection, key = its last segment) that ended with clock `clock`:
   *   clock row   (path, NODE_CLOCK)            ts = the clock if it is the single component {writer: ts}; otherwise the path is marked host-only
   *   value rows  (path, f) for every safe-integer field f of an object value; (path, null) for a safe-integer primitive;
   *               (parent node, key) as well — the same leaf seen as a FIELD of its parent node, which is what index(collection, field) scans;
   *   tombstones  for every such row the old value had and the new one has not: a dominating object REPLACES the node
   *               (src/bullet-crt.js:236-248), a deleted entry becomes null (src/bullet-network-sync.js:553-555).
   * Queued here, sent with ONE bmx_put_rows in front of the next device operation (no device call per put).
   */
  _mirrorWrite(path, oldValue, value, clock) {
    if (this._opts.writers) return this._mirrorWriteVector(path, oldValue, value, clock);
    if (!this._graph) return;                                  // no device table in use
    const ts = scalarClock(clock, this._opts.writer || this.bullet.id);
    // host-only paths: a clock the device cannot hold, or a STRING value — an object that meets it under an identical clock is compared with it as text
    // ("[object Object]" < "zebra": src/bullet-crt.js:11-15), the one case where an object entry's fate depends on the stored value
    if (ts < 0 || typeof value === "string") this._markHostOnly(path); else this._clearHostOnly(path);
    const cut = path.lastIndexOf("/");
    const parent = cut < 0 ? "" : path.slice(0, cut);
    const keys = this._graph.keys;
    const q = this._putQueue();
    keys.lookup(path);
    const lo = keys.lo, hi = keys.hi;
    const intLeaf = this._intEntries && isDeviceInt(value) && value !== 0;   // (a stored 0 is read back as {}: src/bullet.js:115-129 replaces falsy values on the way)
    this._setKind(keys.idx, intLeaf ? KIND_INT : KIND_NODE);
    if (!value || this._nFalsy) this._setFalsy(keys.idx, !value);
    if (ts >= 0) q.push(lo, hi, keys.fieldOf(parent, NODE_CLOCK), ts, intLeaf ? value : this._nodeSeq++);
    this._queueValueRows(q, path, parent, lo, hi, oldValue, value, ts < 0 ? 0 : ts, false);
    if (cut > 0) {                                             // the leaf as a field of its parent node
      const key = path.slice(cut + 1), c2 = parent.lastIndexOf("/");
      const was = isDeviceInt(oldValue), is = isDeviceInt(value);
      if (was || is) {
        keys.lookup(parent);
        q.push(keys.lo, keys.hi, keys.fieldOf(c2 < 0 ? "" : parent.slice(0, c2), key), ts < 0 ? 0 : ts, is ? value : VAL_DELETED);
      }
    }
  }

  /* value rows of node `path` (id lo:hi) for `value`, tombstones for what `oldValue` had on the device and `value` has not */
  _queueValueRows(q, path, parent, lo, hi, oldValue, value, ts, unique) {
    const keys = this._graph.keys;
    const newObj = isMergeable(value), oldObj = isMergeable(oldValue);
    if (newObj) {
      for (const f in value) {
        if (f === "__vectorClock" || f === "__fromNetwork" || !Object.prototype.hasOwnProperty.call(value, f)) continue;
        const v = value[f];
        let row;
        if (isDeviceInt(v)) row = v;
        else if (oldObj && isDeviceInt(oldValue[f])) row = VAL_DELETED;
        else continue;
        if (unique) q.pushUnique(lo, hi, keys.fieldOf(parent, f), ts, row); else q.push(lo, hi, keys.fieldOf(parent, f), ts, row);
      }
    }
    if (oldObj) {
      for (const f in oldValue) {
        if (!Object.prototype.hasOwnProperty.call(oldValue, f) || !isDeviceInt(oldValue[f])) continue;
        if (newObj && Object.prototype.hasOwnProperty.call(value, f)) continue;
        if (unique) q.pushUnique(lo, hi, keys.fieldOf(parent, f), ts, VAL_DELETED); else q.push(lo, hi, keys.fieldOf(parent, f), ts, VAL_DELETED);
      }
    }
    const prim = isDeviceInt(value) ? value : (isDeviceInt(oldValue) ? VAL_DELETED : undefined);
    if (prim !== undefined) { if (unique) q.pushUnique(lo, hi, keys.fieldOf(parent, null), ts, prim); else q.push(lo, hi, keys.fieldOf(parent, null), ts, prim); }
  }

  /* the queue of rows for bmx_put_rows; the graph flushes it in front of every device operation it is asked for (DeviceGraph.preOp), so
   * direct readers of the graph (getRows, scans, dumps) never see a table that lags the host's own writes */
  _putQueue() {
    if (!this._puts) {
      const g = this._graph;
      this._puts = new PutQueue(g);
      if (g && !g.preOp) g.preOp = () => { if (this._lazy && this._lazy.n) this._lazy.foldAll(); this._flushDeviceWrites(); };   // (recorded winners first: their integer fields become rows in the fold)
    }
    return this._puts;
  }

  /* in front of a merge: rows of host writes must be there (clock rows decide the merge); the value rows of earlier batches' winners only feed
   * scans, so they stay queued — a synchronous put would make the event loop wait for the merge in flight — until a reader comes or a million wait */
  _flushBeforeMerge() {
    const q = this._puts;
    if (q && q.n && (q.hasHostRows() || q.n > (1 << 20))) this._flushDeviceWrites();
  }

  _flushDeviceWrites() {
    const q = this._puts;
    if (!q || q.n === 0) return;
    const g = this.graph;
    for (const cols of q.take()) { g.putRows(cols); g.giveColumns(cols); }   // bmx_put_rows has returned: the columns are free again
  }

  /*
   * Bring a NEW device table in step with what the facade already holds (writes that happened before the graph existed, a store loaded from
   * disk: gpu-storage.js): every path with a clock gets its clock row (or the host-only mark) and its value rows. Called once when the graph
   * is created (index.js attach(), GpuStorage.restoreDevice()). -> rows queued
   */
  seedDevice() {
    if (!this._graph && !(this._opts.writers && this._vc)) return 0;
    const b = this.bullet, meta = b.meta || {};
    let n = 0;
    for (const p of Object.keys(meta)) {
      const m = meta[p];
      if (!m || !m.vectorClock) continue;
      this._mirrorWrite(p, undefined, peek(b.store, p), m.vectorClock);
      n++;
    }
    if (this._graph) this._flushDeviceWrites();
    return n;
  }

  /* May entry e = {path, data, vectorClock} be resolved by the device?  (SURVEY §8(a) contract, node level)
   *   data: a plain object with at least one field — of ANY JSON type: the device resolves the node's CLOCK, the object itself stays on the host and
   *   replaces the node when its entry wins (two objects under identical clocks: compare() answers +1, src/bullet-crt.js:11-15, so no value is ever
   *   looked at); only its safe-integer fields additionally become device value rows for the index scans — or a safe integer;
   *   clock: the single component {writer: ts}; the path's stored clock is one the device holds and its stored value is not a string (not host-only),
   *   and a first sight of it would store {writer: 2} in the reference too (src/bullet-crt.js:172-185 increments whatever crt.vectorClocks already
   *   holds for a path that has no meta clock yet). */
  entryEligible(e, writer, objectsOnly) {
    if (!e || e.deleted) return false;
    if (this._opts.writers) { if (clockKeyset(e.vectorClock, this.vcTable.writerIndex, this._vcComps()) < 0) return false; }   // any clock over the table's writers
    else if (scalarClock(e.vectorClock, writer) < 0) return false;
    const d = e.data;
    const isInt = isDeviceInt(d);
    if (isInt) { if (objectsOnly || !this._intEntries || d === 0 || this._opts.writers) return false; }
    else {
      if (!isMergeable(d)) return false;
      let k = 0;
      for (const f in d) {
        if (f === "__vectorClock" || f === "__fromNetwork" || !Object.prototype.hasOwnProperty.call(d, f)) continue;
        k++;
        break;
      }
      if (k === 0) return false;
    }
    if (!this._pathEligible(e.path, writer)) return false;
    if (this._graph && !this._opts.writers && (isInt || this._nIntPaths > 0)) {     // the path's kind must be the entry's (or still open)
      const keys = this._graph.keys;
      keys.lookup(e.path);
      const kd = this._kindAt(keys.idx);
      if (kd !== 0 && kd !== (isInt ? KIND_INT : KIND_NODE)) return false;
    }
    return true;
  }

  /* the part of entryEligible that depends on what this peer holds for the path */
  _pathEligible(path, writer) {
    if (this._hostOnly.size && this._hostOnly.has(path)) return false;
    if (this._apiClocks.size && this._apiClocks.has(path)) {     // createVectorClock / getVectorClock / createUpdate were used on this path
      const m = this.bullet.meta[path];
      if (m && m.vectorClock) this._apiClocks.delete(path);
      else if (scalarClock(this.vectorClocks.get(path), writer) !== 1) return false;   // the reference would store that clock + 1, not {writer: 2}
    }
    return true;
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
   *        duplicate keys in the batch; about twice as slow), markCreated: bool (bit 31 of an applied index: that delta created its row)}
   * -> {applied: Uint32Array (ascending delta indices whose value is now stored), flags, nApplied, nConflicts, nRows}
   */
  mergeBatch(cols, opts = {}) {
    this._flushBeforeMerge();
    const g = this.graph;
    let mode = opts.insertMode === "delta" ? g.native.INSERT_DELTA : g.native.INSERT_REFERENCE;
    if (opts.uniqueKeys) mode |= g.native.MERGE_UNIQUE_KEYS;
    if (opts.strictFlags) mode |= g.native.MERGE_STRICT_FLAGS;
    if (opts.markCreated) mode |= g.native.MERGE_MARK_CREATED;
    return g.mergeBatch(cols, mode);
  }

  /** Same as mergeBatch but off the event loop: resolves to the same result object (the reference API is synchronous;
   *  this is the Promise variant SURVEY §8(b) calls for). The columns must not be mutated until it settles. */
  mergeBatchAsync(cols, opts = {}) {
    this._flushBeforeMerge();
    const g = this.graph;
    let mode = opts.insertMode === "delta" ? g.native.INSERT_DELTA : g.native.INSERT_REFERENCE;
    if (opts.uniqueKeys) mode |= g.native.MERGE_UNIQUE_KEYS;
    if (opts.strictFlags) mode |= g.native.MERGE_STRICT_FLAGS;
    if (opts.markCreated) mode |= g.native.MERGE_MARK_CREATED;
    return g.mergeBatchAsync(cols, mode);
  }

  /**
   * Batch adapter for sync chunks (reference loop: src/bullet-network-sync.js:551-569), NODE-level like the reference: an entry
   * {path, data, vectorClock} is ONE conflict-resolution unit — the object (or integer) at `path` under the clock in meta[path]
   * (src/bullet-crt.js:329-385). Every eligible entry (entryEligible) becomes one device delta on the node's CLOCK row
   * (path, NODE_CLOCK): ts = its clock, val = its arrival number. The scalar-clock merge then IS resolve() for the node: first sight stores
   * {writer: 2} (:172-185), a dominating clock wins (:236-248), an older one is historical (:251-263), and on equal clocks the later
   * arrival wins — what compare() answers for two objects (:11-15, :200-233). Winners are the entries whose object is the node's final
   * value; entries outside the contract are returned in `host` for setData().
   * opts.apply: true = the winners REPLACE their nodes in the facade's store, meta[path] gets the stored clock, log and listeners run once
   * for the batch (N1, batch-apply.js); "each" = one _applyUpdate per winner. The winners' integer fields are queued as device value rows
   * (and tombstones for the fields the replaced node had: known only when the store is kept, i.e. with opts.apply) for the device-side
   * indexes; they reach the device in front of the next device read (opts.valueRows: false leaves them out: clock rows only).
   * -> {appliedEntries: Int32Array (winner k is entry appliedEntries[k]), applied: [{entry, field: null}] (the same, built on first read), nApplied, nConflicts,
   *     nRows, host: [entry indices], broadcast: [{path, broadcastData}] when applied}
   */
  mergeEntries(entries, opts = {}) {
    if (this._opts.writers) return this._mergeEntriesVector(entries, opts);
    const p = this._packEntries(entries, opts);
    // (nothing packed — the walk of batch-sync.js stopped at its first entry: no device call)
    const r = p.cols.n === 0 && opts.stopAtHost ? { applied: new Uint32Array(0), flags: null, nApplied: 0, nConflicts: 0, nRows: undefined } : this.mergeBatch(p.cols, p.mergeOpts);
    return this._finishEntries(entries, p, r, opts);
  }

  /**
   * mergeEntries off the event loop: the merge runs on a worker thread of the addon, the Promise resolves to the same result. Calls may be issued
   * back to back without awaiting: the addon keeps the issue order, so while chunk b is on the GPU the event loop packs chunk b + 1 (path hashing,
   * typed columns), and the winners of chunk b are applied when its Promise settles — one chunk late, in order (src/bullet-network-sync.js:551-569
   * is the loop this pipelines). Host-path writes issued in between are ordered with the merges through the same queue.
   */
  mergeEntriesAsync(entries, opts = {}) {
    if (this._opts.writers) {          // the same split for the vector-clock table: packed now, merged on a worker thread of the addon, applied when it settles
      const p = this._packVector(entries, opts);
      if (!p.used.n) return Promise.resolve(this._finishVector(entries, p, { updated: new Uint32Array(0), flags: new Uint8Array(0), nRows: this.vcTable.rowCount() }, opts));
      return this.vcTable.mergeBatchAsync(p.used).then((r) => this._finishVector(entries, p, r, opts));
    }
    const p = this._packEntries(entries, opts);
    return this.mergeBatchAsync(p.cols, p.mergeOpts).then((r) => this._finishEntries(entries, p, r, opts));
  }

  /** Run chunks of entries through mergeEntriesAsync with `depth` (default 2) in flight. -> Promise<{nApplied, nConflicts, nRows, host: [[chunk, entry index]]}> */
  async mergeEntriesPipelined(chunks, opts = {}) {
    const depth = Math.max(1, opts.depth || 2);
    const inflight = [];
    const total = { nApplied: 0, nConflicts: 0, nRows: 0, host: [] };
    let ci = 0;
    const settle = async () => {
      const f = inflight.shift();
      const r = await f.promise;
      total.nApplied += r.nApplied; total.nConflicts += r.nConflicts; total.nRows = r.nRows;
      for (const k of r.host) total.host.push([f.chunk, k]);
    };
    for (const chunk of chunks) {
      inflight.push({ chunk: ci++, promise: this.mergeEntriesAsync(chunk, opts) });
      if (inflight.length >= depth) await settle();
    }
    while (inflight.length) await settle();
    return total;
  }

  /*
   * entries -> one clock-row delta per eligible entry (entryEligible's rules, checked here in one pass over the entry).
   * When the store is not kept (no opts.apply) the value rows of EVERY eligible entry are prepared in the same pass — the entry is in cache
   * now and is not when its chunk comes back from the GPU — as typed rows [rowStart[i], rowStart[i + 1]) of `vcols`; _finishEntries copies the
   * winners' rows into the put queue without touching an entry again.
   */
  _packEntries(entries, opts) {
    const writer = opts.writer || this.bullet.id;
    const g = this.graph;
    const keys = g.keys;
    // opts.from / opts.stopAtHost (batch-sync.js): start at entry `from` and STOP at the first entry that is not the device's — the caller resolves that
    // one on the host and comes back for the rest, so that a chunk is walked once (no separate eligibility pass) and still applied in entry order.
    // opts.objectsOnly: integer entries are not the device's either (a primitive of a sync chunk is a LOCAL write: src/bullet-network-sync.js:560-563)
    const from = opts.from | 0, stopAtHost = opts.stopAtHost === true, objectsOnly = opts.objectsOnly === true || !this._intEntries;
    let n = entries.length, consumed = n;
    const cols = g.takeColumns(Math.max(n - from, 1));        // given back by _finishEntries
    const rowEntry = new Int32Array(Math.max(n - from, 1));
    const emit = !opts.apply && opts.valueRows !== false;
    let vcols = emit ? g.takeColumns(Math.max(2 * (n - from), 64)) : null, vn = 0;
    const rowStart = emit ? new Int32Array(n - from + 1) : null;
    const rowNode = opts.apply ? new Int32Array(Math.max(n - from, 1)) : null;   // the node (dictionary number of its path) of every delta: _unaliasLosers
    const host = [];
    const guarded = this._hostOnly.size > 0 || this._apiClocks.size > 0;
    let i = 0, lateHost = false;
    // consecutive entries usually share their collection: the parent is recognised by the position of the last "/" and the hash of the prefix
    // (by-products of the id hash), its string is sliced and its field hashes looked up only when it changes
    let parent = null, pCut = -2, pH1 = 0, pH2 = 0, clockField = 0, per = null;
    // Blocks of PACK_BLOCK entries, three passes each: (1) eligibility + the id hash of the path (touches the entry, no table), (2) the dictionary probes of
    // the whole block back to back — independent cache misses the CPU overlaps —, (3) the rows. One entry at a time the probe's miss stood in the critical path.
    keys._block(PACK_BLOCK);
    const bPath = this._bPath || (this._bPath = new Array(PACK_BLOCK)), bEnt = this._bEnt || (this._bEnt = new Int32Array(PACK_BLOCK));
    const bTs = this._bTs || (this._bTs = new Float64Array(PACK_BLOCK));
    for (let e0 = from; e0 < n; e0 += PACK_BLOCK) {
      const e1 = Math.min(n, e0 + PACK_BLOCK);
      let m = 0;
      for (let ei = e0; ei < e1; ei++) {
        const e = entries[ei];
        let mine = !(!e || e.deleted);
        let ts = 0, d;
        if (mine) {
          ts = scalarClock(e.vectorClock, writer);
          d = e.data;
          if (ts < 0 || (!isDeviceInt(d) && !isMergeable(d)) || d === 0 || (guarded && !this._pathEligible(e.path, writer))) mine = false;
          else if (typeof d !== "number") {                     // {} (nothing but transport tags): left to setData
            mine = false;
            for (const f in d) { if (f !== "__vectorClock" && f !== "__fromNetwork" && Object.prototype.hasOwnProperty.call(d, f)) { mine = true; break; } }
          } else if (objectsOnly) mine = false;
        }
        if (!mine) {
          if (stopAtHost) { consumed = ei; n = ei; break; }     // this block's collected entries are still packed below; nothing behind `ei` is looked at
          host.push(ei); continue;
        }
        keys.hashInto(e.path, m);
        bPath[m] = e.path; bEnt[m] = ei; bTs[m] = ts;
        m++;
      }
      keys.probeBlock(bPath, m);
      for (let x = 0; x < m; x++) {
        const ei = bEnt[x], e = entries[ei], d = e.data, ts = bTs[x];
        const lo = keys.bLo[x], hi = keys.bHi[x];
        const isInt = typeof d === "number";
        if (isInt || this._nIntPaths > 0) {                       // the path's kind must be the entry's (_kindAt); a path seen for the first time takes it
          const want = isInt ? KIND_INT : KIND_NODE, kd = this._kindAt(keys.bIdx[x]);
          if (kd === 0) this._setKind(keys.bIdx[x], want);
          else if (kd !== want) {
            if (stopAtHost) { consumed = ei; n = ei; break; }   // (its successors in this block were only hashed: nothing of them is packed)
            host.push(ei); lateHost = true; continue;
          }
        }
        if (keys.bCut[x] !== pCut || keys.bP1[x] !== pH1 || keys.bP2[x] !== pH2) {
          pCut = keys.bCut[x]; pH1 = keys.bP1[x]; pH2 = keys.bP2[x];
          parent = pCut < 0 ? "" : bPath[x].slice(0, pCut);
          clockField = keys.fieldOf(parent, NODE_CLOCK);
          per = keys._fieldCache.get(parent);
        }
        if (typeof d === "number") {
          if (emit) {
            if (vn === vcols.n) vcols = this._growColumns(vcols, vn);
            vcols.set2(vn++, lo, hi, keys.fieldOf(parent, null), ts, d);
          }
        } else if (emit) {
          for (const f in d) {
            if (f === "__vectorClock" || f === "__fromNetwork" || !Object.prototype.hasOwnProperty.call(d, f)) continue;
            const v = d[f];
            if (!isDeviceInt(v)) continue;                      // strings, nested objects, ...: part of the node on the host, no device row
            let h = per.get(f);
            if (h === undefined) h = keys.fieldOf(parent, f);
            if (vn === vcols.n) vcols = this._growColumns(vcols, vn);
            vcols.set2(vn++, lo, hi, h, ts, v);
          }
        }
        if (emit) rowStart[i + 1] = vn;
        if (rowNode) rowNode[i] = keys.bIdx[x];
        cols.set2(i, lo, hi, clockField, ts, isInt ? d : this._nodeSeq++);     // an integer path's ties are decided by value, a node's by arrival
        rowEntry[i++] = ei;
      }
      for (let x = 0; x < m; x++) bPath[x] = undefined;          // no strings kept alive by the scratch array
    }
    if (lateHost) host.sort((a, b) => a - b);
    // one context: the winners that created their node are marked (stored clock = the insert rule's, no read-back); shards: read back
    const mergeOpts = g.comm ? opts : Object.assign({}, opts, { markCreated: true });
    return { cols: cols.slice(i), rowEntry, host, writer, mergeOpts, vcols, rowStart, rowNode, consumed };
  }

  _growColumns(cols, used) {
    const g = this.graph, bigger = g.takeColumns(2 * cols.n);
    bigger.id.set(cols.id.subarray(0, used)); bigger.field.set(cols.field.subarray(0, used)); bigger.ts.set(cols.ts.subarray(0, used)); bigger.val.set(cols.val.subarray(0, used));
    g.giveColumns(cols);
    return bigger;
  }

  _finishEntries(entries, p, r, opts) {
    const nw = r.applied.length;
    const appliedEntries = new Int32Array(nw);                 // winner k is entry appliedEntries[k] (ascending)
    for (let k = 0; k < nw; k++) appliedEntries[k] = p.rowEntry[r.applied[k] & 0xffffff];
    let broadcast = [];
    if (p.vcols) this._queueWinnerRows(p, r.applied, !this.graph.comm, opts.insertMode === "delta");
    else broadcast = this._applyWinners(entries, p.cols, r.applied, appliedEntries, opts.apply, opts.broadcast !== false, !this.graph.comm, opts.insertMode === "delta", p.writer, opts.valueRows !== false, p.rowNode);
    if (opts.apply) this._unaliasLosers(entries, p.rowEntry, p.rowNode, p.cols.n, r.applied, this.graph.keys.size);
    this.graph.giveColumns(p.cols);                           // the merge has returned and the winners' ids are copied
    if (p.vcols) this.graph.giveColumns(p.vcols);
    if (opts.apply) this._notifyIndexHook(entries, p.host, opts.from | 0, p.consumed);
    const out = { appliedEntries, nApplied: r.nApplied, nConflicts: r.nConflicts, nRows: r.nRows, host: p.host, broadcast: opts.apply ? broadcast : undefined, consumed: p.consumed, packed: p.cols.n };
    let list = null;                                           // `applied` in the older shape, built only if somebody reads it (an object per winner is what the ingestion rate can do without)
    Object.defineProperty(out, "applied", { enumerable: true, get() { if (!list) { list = new Array(nw); for (let k = 0; k < nw; k++) list[k] = { entry: appliedEntries[k], field: null }; } return list; } });
    return out;
  }

  /*
   * What crt.vectorClocks holds after a batch. The reference's resolve() stores the MERGED clock of every entry it resolves in crt.vectorClocks
   * (src/bullet-crt.js:193-198), applied or not; an applied entry's merged clock is also the object meta[path] gets (one object in both places: a later
   * local write increments it in place and then meets "identical clocks", SURVEY §5), a losing entry's is a fresh object that only crt.vectorClocks
   * holds (a later local write increments THAT and dominates). The winners are handled where they are applied; this gives every node whose LAST entry of
   * the batch lost what the reference would hold: the merge of that entry's clock with the node's stored clock, as a new object — and the store the side
   * effect of the read every resolution starts with.
   * rowNode[j]: the node of delta j; winners: delta indices (bits 24..31 may carry marks).
   */
  _unaliasLosers(entries, rowEntry, rowNode, nrows, winners, nNodes) {
    if (!rowNode || nrows === 0) return;
    if (!this._seen || this._seen.length < nNodes) { this._seen = new Uint32Array(Math.max(1024, 2 * nNodes)); this._seenStamp = 0; }
    if (++this._seenStamp === 0xffffffff) { this._seen.fill(0); this._seenStamp = 1; }
    const seen = this._seen, stamp = this._seenStamp;
    let meta = null;                                       // (read only where it is needed: with a lazy store the property is an accessor that folds)
    const won = new Uint8Array(nrows);
    for (let k = 0; k < winners.length; k++) won[winners[k] & 0xffffff] = 1;
    const lazy = !this._opts.writers && this._graph;       // single-writer clocks: the loser's merged clock is a copy of the stored one ({w: max(its ts, stored ts)} = stored)
    const readBack = typeof this.bullet._getData === "function";
    for (let j = nrows - 1; j >= 0; j--) {
      const node = rowNode[j];
      if (seen[node] === stamp) continue;
      seen[node] = stamp;
      if (won[j]) continue;
      if (lazy) {
        this._markLazyClock(node, P_CLOCK_COPY);             // what the eager form below stores, when somebody asks for it (ClockMap)
        // the read below matters only where it replaces a falsy value on the way; none is known anywhere -> nothing to replace
        if (this._nFalsy > 0 && readBack) this.bullet._getData(entries[rowEntry[j]].path);
        continue;
      }
      if (meta === null) meta = this.bullet.meta || {};
      const e = entries[rowEntry[j]], m = meta[e.path];
      if (m && m.vectorClock) this.vectorClocks.set(e.path, this.mergeVectorClocks(e.vectorClock, m.vectorClock));
      // ... and the read the reference's handleUpdate starts with: _getData(path) REPLACES a falsy value on the way (null of a deleted node, 0, "") by {}
      // (src/bullet.js:115-129), whether the entry then wins or not. A winner overwrites it; behind a loser it stays.
      if (typeof this.bullet._getData === "function") this.bullet._getData(e.path);
    }
  }

  /* the prepared value rows of the winners -> one segment of the put queue, under the clock each node now stores: the entry's own, or — a winner that
   * CREATED its node, reference insert rule — {writer: 2} (bit 31 of its index on one context; read back from the clock rows on a sharded graph).
   * No tombstones: what a replaced node held is only known with the store (opts.apply keeps it). Typed copies only. */
  _queueWinnerRows(p, appliedIdx, marked, deltaMode) {
    const n = appliedIdx.length;
    if (n === 0) return;
    const rs = p.rowStart, v = p.vcols;
    let total = 0;
    for (let k = 0; k < n; k++) { const j = appliedIdx[k] & 0xffffff; total += rs[j + 1] - rs[j]; }
    let ts32 = null;
    if (!marked) {
      const ids = new BigUint64Array(n), id32 = new Uint32Array(ids.buffer), fields = new Uint32Array(n), c = p.cols;
      for (let k = 0; k < n; k++) { const j = appliedIdx[k] & 0xffffff; id32[2 * k] = c._id32[2 * j]; id32[2 * k + 1] = c._id32[2 * j + 1]; fields[k] = c.field[j]; }
      const rows = this.graph.getRows(ids, fields);
      ts32 = new Uint32Array(rows.ts.buffer, rows.ts.byteOffset, n * 2);
    }
    const q = this._putQueue();
    const sg = q.uniqueSegment(total), o = sg.cols;
    const oi = o._id32, ot = o._ts32, ov = o._val32, of = o.field, vi = v._id32, vt = v._ts32, vv = v._val32, vf = v.field;
    let w = sg.n;
    for (let k = 0; k < n; k++) {
      const a = appliedIdx[k], j = a & 0xffffff, created = (a >>> 31) !== 0 && !deltaMode;
      for (let x = rs[j], end = rs[j + 1]; x < end; x++, w++) {
        oi[2 * w] = vi[2 * x]; oi[2 * w + 1] = vi[2 * x + 1]; of[w] = vf[x]; ov[2 * w] = vv[2 * x]; ov[2 * w + 1] = vv[2 * x + 1];
        if (ts32) { ot[2 * w] = ts32[2 * k]; ot[2 * w + 1] = ts32[2 * k + 1]; }
        else if (created) { ot[2 * w] = 2; ot[2 * w + 1] = 0; }
        else { ot[2 * w] = vt[2 * x]; ot[2 * w + 1] = vt[2 * x + 1]; }
      }
    }
    q.n += w - sg.n; sg.n = w;
    q.closeBatch();
  }

  /*
   * The reference's query engine keeps its indices current through a wrapper around setData (src/bullet-query.js:13-21): after EVERY
   * write, accepted or not, it calls _updateIndices(path, data). A batch that is applied here never passes through setData, so the
   * hook is called for it: GpuQuery marks the touched children of its indexed collections (it re-reads them from the store at the next
   * query), a reference BulletQuery gets its own _updateIndices. Entries the batch handed back (`host`) reach the hook through setData.
   */
  _notifyIndexHook(entries, hostIdx, from = 0, to = entries.length) {
    const q = this.bullet.query;
    if (!q) return;
    const skip = hostIdx && hostIdx.length ? new Set(hostIdx) : null;
    if (typeof q._touch === "function") {
      if (!q.indexedPaths || q.indexedPaths.size === 0) return;
      for (let i = from; i < to; i++) if (!skip || !skip.has(i)) q._touch(entries[i].path);
    } else if (typeof q._updateIndices === "function") {
      if (!q.indexedPaths || q.indexedPaths.size === 0) return;
      for (let i = from; i < to; i++) if (!skip || !skip.has(i)) q._updateIndices(entries[i].path, entries[i].data);
    }
  }

  /*
   * N1 (SURVEY §8(f)): hand the batch's final winners to the facade in ONE pass. A winner is a whole node: its stored clock is read back
   * from its clock row (a first sight stored {writer: 2}, not the incoming clock), its value replaces the node. mode === true: store, meta,
   * op log and listeners are updated once for the whole batch (batch-apply.js: ring-style log, ancestor listeners de-duplicated; a facade
   * may supply its own `_applyBatch`); mode === "each": the facade's per-write `_applyUpdate` is called once per winner
   * (src/bullet.js:184-266), the reference's own cost per write; falsy: the store is left alone (the caller applies). In every mode the
   * winners' value rows are queued for the device. Returns what setData would have broadcast per winner (src/bullet-crt.js:371-376).
   */
  _applyWinners(entries, cols, appliedIdx, applied, mode, wantBroadcast = true, marked = false, deltaMode = false, writerOpt, valueRows = true, rowNode = null) {
    const n = appliedIdx.length;
    if (n === 0) return [];
    const id32 = new Uint32Array(2 * n);
    const src32 = cols._id32 || new Uint32Array(cols.id.buffer, cols.id.byteOffset, cols.id.length * 2);
    for (let k = 0; k < n; k++) { const j = appliedIdx[k] & 0xffffff; id32[2 * k] = src32[2 * j]; id32[2 * k + 1] = src32[2 * j + 1]; }
    let ts32 = null;
    if (!marked) {        // sharded graph: the stored clocks are read back from the clock rows
      const ids = new BigUint64Array(id32.buffer), fields = new Uint32Array(n);
      for (let k = 0; k < n; k++) fields[k] = cols.field[appliedIdx[k] & 0xffffff];
      const rows = this.graph.getRows(ids, fields);
      ts32 = new Uint32Array(rows.ts.buffer, rows.ts.byteOffset, n * 2);
    }
    const writer = writerOpt || this._opts.writer || this.bullet.id;
    const b = this.bullet;
    const q = this._putQueue();
    const lz = this._lazy;
    if (lz && mode === true && !wantBroadcast && rowNode && lz.usable()) {
      // lazy-store.js: RECORD the winners (entry, path number, node id, stored clock); store, meta, op log and the winners' value rows follow when somebody
      // looks (_foldRecords). What the next chunk's resolution needs is already in place: the clock rows on the device, the path dictionary, the marks below.
      lz.beginBatch(writer, valueRows);
      for (let k = 0; k < n; k++) {
        const a = appliedIdx[k], e = entries[applied[k]], idx = rowNode[a & 0xffffff];
        const ts = ts32 ? ts32[2 * k + 1] * 4294967296 + ts32[2 * k] : ((a >>> 31) && !deltaMode ? 2 : e.vectorClock[writer]);
        lz.push(e, idx, id32[2 * k], id32[2 * k + 1], ts);
        this._markLazyClock(idx, P_CLOCK_META);
        if (this._nFalsy) this._setFalsy(idx, !e.data);
      }
      lz.endBatch();
      return [];
    }
    if (lz && lz.n) lz.foldAll();               // anything applied at once comes BEHIND what was recorded earlier
    const updates = mode ? new Array(n) : null;
    const leaf = this._leafKeys || (this._leafKeys = []);
    // consecutive winners usually share their collection: its path string and its object in the store are looked up when it changes
    let parent = null, pLen = -2, pNode;
    for (let k = 0; k < n; k++) {
      const e = entries[applied[k]];
      const path = e.path, cut = path.lastIndexOf("/");
      if (cut !== pLen || !path.startsWith(parent)) { parent = cut < 0 ? "" : path.slice(0, cut); pLen = cut; pNode = mode ? peek(b.store, parent) : undefined; }
      // the clock the node now stores: read back, or — one context — the entry's own unless this winner CREATED the node (bit 31: the insert rule's {writer: 2})
      const ts = ts32 ? ts32[2 * k + 1] * 4294967296 + ts32[2 * k] : ((appliedIdx[k] >>> 31) && !deltaMode ? 2 : e.vectorClock[writer]);
      let value = e.data, old;
      if (mode) {
        const clock = {};
        clock[writer] = ts;
        if (isMergeable(value)) {                                                       // the node's new value: a copy of the entry's object without the transport tags
          if (value.__vectorClock === undefined && value.__fromNetwork === undefined) value = Object.assign({}, value);
          else { const clean = {}; for (const f of Object.keys(value)) if (f !== "__vectorClock" && f !== "__fromNetwork") clean[f] = value[f]; value = clean; }
        }
        const plain = cut !== path.length - 1 && path.indexOf("//") < 0;
        // the node's key in its collection object, ONE string per path for good (by the path's dictionary number): a freshly sliced string has to be
        // interned by the engine before it can address a property of a million-key object — 0.5 us per winner, twice (old value, store write)
        let key;
        if (plain) {
          if (rowNode) { const idx = rowNode[appliedIdx[k] & 0xffffff]; key = leaf[idx]; if (key === undefined) key = leaf[idx] = (cut < 0 ? path : path.slice(cut + 1)); }
          else key = cut < 0 ? path : path.slice(cut + 1);
        }
        updates[k] = { path, value, vectorClock: clock, parentHint: parent, cutHint: cut, keyHint: key };   // (the hints spare applyBatch the same string work)
        // crt.vectorClocks[path] = the same object meta will hold: local writes increment it in place, like the reference's (SURVEY §5 aliasing) — as a
        // MARK where the path's number is at hand (ClockMap), as an entry otherwise
        if (rowNode) { const idx = rowNode[appliedIdx[k] & 0xffffff]; this._markLazyClock(idx, P_CLOCK_META); if (this._nFalsy) this._setFalsy(idx, !value); }
        else this.vectorClocks.set(path, clock);
        if (!plain) old = peek(b.store, path);   // empty segments: the walk that skips them
        // (one look-up: an inherited property — a segment called "constructor" — is a function, which has neither integer fields nor is one: no rows either way)
        else if (pNode !== null && typeof pNode === "object") old = pNode[key];
      }
      if (valueRows) this._queueValueRows(q, path, parent, id32[2 * k], id32[2 * k + 1], old, value, ts, true);
    }
    q.closeBatch();
    if (!mode) return [];
    if (mode === "each") {
      if (typeof b._applyUpdate === "function") for (const u of updates) b._applyUpdate(u.path, u.value, u.vectorClock, true);
      return wantBroadcast ? updates.map((u) => ({ path: u.path, broadcastData: isMergeable(u.value) ? Object.assign({}, u.value, { __vectorClock: u.vectorClock }) : u.value })) : [];
    }
    if (typeof b._applyBatch === "function") return b._applyBatch(updates, true) || [];
    return require("./batch-apply").applyBatch(b, updates, true, wantBroadcast);
  }

  /* lazy-store.js: the recorded winners [from, to) of ONE batch -> store, meta, op log (batch-apply.js, with the batch's arrival time) and the device's value rows;
   * the same per-winner work, in the same order, as the eager loop of _applyWinners above. */
  _foldRecords(ent, idxs, los, his, tss, from, to, at, writer, valueRows) {
    const n = to - from;
    if (n <= 0) return;
    const b = this.bullet, q = this._putQueue();
    const updates = new Array(n);
    const leaf = this._leafKeys || (this._leafKeys = []);
    let parent = null, pLen = -2, pNode;
    for (let k = 0; k < n; k++) {
      const e = ent[from + k], idx = idxs[from + k], ts = tss[from + k];
      const path = e.path, cut = path.lastIndexOf("/");
      if (cut !== pLen || !path.startsWith(parent)) { parent = cut < 0 ? "" : path.slice(0, cut); pLen = cut; pNode = peek(b.store, parent); }
      let value = e.data, old;
      const clock = {};
      clock[writer] = ts;
      if (isMergeable(value)) {
        if (value.__vectorClock === undefined && value.__fromNetwork === undefined) value = Object.assign({}, value);
        else { const clean = {}; for (const f of Object.keys(value)) if (f !== "__vectorClock" && f !== "__fromNetwork") clean[f] = value[f]; value = clean; }
      }
      const plain = cut !== path.length - 1 && path.indexOf("//") < 0;
      let key;
      if (plain) { key = leaf[idx]; if (key === undefined) key = leaf[idx] = (cut < 0 ? path : path.slice(cut + 1)); }
      updates[k] = { path, value, vectorClock: clock, parentHint: parent, cutHint: cut, keyHint: key };
      if (!plain) old = peek(b.store, path);
      else if (pNode !== null && typeof pNode === "object") old = pNode[key];
      if (valueRows) this._queueValueRows(q, path, parent, los[from + k], his[from + k], old, value, ts, true);
    }
    q.closeBatch();
    require("./batch-apply").applyBatch(b, updates, true, false, at);
  }

  /* ---------------------------------------------------------------- N4: K-writer vector clocks on the device */
  get vcTable() {
    if (!this._vc) {
      const { DeviceVcTable } = require("./device-graph");
      this._vc = new DeviceVcTable(this._opts.writers, this.bullet.id, this._opts);   // throws without the addon / a GPU
      this.seedDevice();                                                              // clocks the facade already holds -> clock rows
    }
    return this._vc;
  }
  _vcComps() { return this._vcScratch || (this._vcScratch = new Uint32Array(this._vc ? this._vc.K : (this._opts.maxWriters || this._opts.writers.length))); }

  /* stored clock of a device row as the object the reference would hold: the row's counters under the keys its key set names, in that order */
  _clockObject(comps, off, keyset) {
    const t = this.vcTable, c = {};
    for (const k of keysetWriters(keyset)) c[t.writers[k]] = comps[off + k];
    return c;
  }

  /* _mirrorWrite in writers mode: the node's clock row goes to the vector-clock table (counters + key set; a clock that names a foreign writer or
   * holds something else than uint32 counters marks the path host-only, and so does a string value: see _mirrorWrite), the value rows to the scalar
   * table the index scans read, exactly as in scalar mode. Both are queued and sent in front of the next device operation. */
  _mirrorWriteVector(path, oldValue, value, clock) {
    if (!this._vc && !this._graph) return;                      // nothing on the device yet: seeded when the tables are created
    const comps = this._vcComps();
    const ks = this._vc ? clockKeyset(clock, this._vc.writerIndex, comps) : 0;
    if (ks < 0 || typeof value === "string") this._markHostOnly(path); else this._clearHostOnly(path);
    const cut = path.lastIndexOf("/");
    const parent = cut < 0 ? "" : path.slice(0, cut);
    if (this._vc && ks >= 0) this._vcPuts.push([path, parent, ks, Array.from(comps), this._nodeSeq++]);
    if (!this._graph) return;
    const keys = this._graph.keys;
    const q = this._putQueue();
    keys.lookup(path);
    this._queueValueRows(q, path, parent, keys.lo, keys.hi, oldValue, value, 0, false);
    if (cut > 0) {                                              // the leaf as a field of its parent node
      const key = path.slice(cut + 1), c2 = parent.lastIndexOf("/");
      const was = isDeviceInt(oldValue), is = isDeviceInt(value);
      if (was || is) {
        keys.lookup(parent);
        q.push(keys.lo, keys.hi, keys.fieldOf(c2 < 0 ? "" : parent.slice(0, c2), key), 0, is ? value : VAL_DELETED);
      }
    }
  }

  /* queued clock rows of host writes -> the vector-clock table (a preload: the last row of a key stays) */
  _flushVcPuts() {
    const puts = this._vcPuts;
    if (!puts.length) return;
    this._vcPuts = [];
    const t = this.vcTable, cols = new VcColumns(puts.length, t.K);
    puts.forEach((p, i) => { t.keys.lookup(p[0]); cols.set2(i, t.keys.lo, t.keys.hi, t.keys.fieldOf(p[1], NODE_CLOCK), p[3], p[2], p[4]); });
    t.loadRows(cols);
  }

  /**
   * mergeEntries() when the resolver was created with opts.writers (general vector clocks, SURVEY §8(f) N4), NODE level like the scalar path:
   * an entry {path, data, vectorClock} whose clock names only the table's writers (any subset, any key order, uint32 counters) is ONE delta on the
   * node's clock row of the vector-clock table — counters, key set, and its arrival number as the value — and the device runs resolve() over each
   * node's deltas in entry order (src/bullet-crt.js:164-279): first sight stores {local: 2}; a dominating clock replaces the node; a dominated one is
   * historical; identical clocks (same keys, same order, same counters) take the later object; everything else is CONCURRENT and the stored clock
   * becomes the merge, in the reference's key order. What the device cannot hold is the objects: the per-delta flags tell the host what each entry did,
   * and with opts.apply the host replays just that over the store — INCOMING replaces the node, CONCURRENT is mergeValues(entry, node) (:122-153) —
   * and applies each changed node once (final value, clock read back with its key order), queueing its integer fields as device value rows.
   * -> {appliedEntries: Int32Array (the last updating entry of every node that changed, ascending), applied (the same as [{entry, field: null}]),
   *     flags: Uint8Array per device delta (1 incoming, 2 current, 4 historical, 8 concurrent), rowEntry: Int32Array (delta -> entry index),
   *     nApplied, nConflicts (concurrent merges), nRows, host: [entry indices], broadcast}
   */
  _mergeEntriesVector(entries, opts = {}) {
    const p = this._packVector(entries, opts);
    const r = p.used.n ? this.vcTable.mergeBatch(p.used) : { updated: new Uint32Array(0), flags: new Uint8Array(0), nRows: this.vcTable.rowCount() };
    return this._finishVector(entries, p, r, opts);
  }

  /* entries -> one delta per eligible entry on its node's clock row in the vector-clock table (counters, key set, arrival number) */
  _packVector(entries, opts = {}) {
    const t = this.vcTable, keys = t.keys;
    const writer = opts.writer || this.bullet.id;
    const n = entries.length;
    this._flushVcPuts();
    const cols = new VcColumns(Math.max(n, 1), t.K);
    const rowEntry = new Int32Array(Math.max(n, 1));
    const rowNode = opts.apply ? new Int32Array(Math.max(n, 1)) : null;
    const comps = this._vcComps();
    const guarded = this._hostOnly.size > 0 || this._apiClocks.size > 0;
    const host = [];
    let i = 0, parent = null, pCut = -2, pH1 = 0, pH2 = 0, clockField = 0;
    // blocks of PACK_BLOCK entries: eligibility + counters + path hash, then the dictionary probes of the block back to back, then the rows (see _packEntries)
    keys._block(PACK_BLOCK);
    const bPath = this._bPath || (this._bPath = new Array(PACK_BLOCK));
    const K = t.K;
    for (let e0 = 0; e0 < n; e0 += PACK_BLOCK) {
      const e1 = Math.min(n, e0 + PACK_BLOCK);
      const first = i;
      let m = 0;
      for (let ei = e0; ei < e1; ei++) {
        const e = entries[ei];
        if (!e || e.deleted) { host.push(ei); continue; }
        const d = e.data;
        {   // objects only: under identical clocks two integers are decided by VALUE (src/bullet-crt.js:200-233), which the arrival number of a clock row cannot say
          let any = false;
          if (isMergeable(d)) for (const f in d) { if (f !== "__vectorClock" && f !== "__fromNetwork" && Object.prototype.hasOwnProperty.call(d, f)) { any = true; break; } }
          if (!any) { host.push(ei); continue; }
        }
        const ks = clockKeyset(e.vectorClock, t.writerIndex, comps);
        if (ks < 0 || (guarded && !this._pathEligible(e.path, writer))) { host.push(ei); continue; }
        keys.hashInto(e.path, m);
        bPath[m++] = e.path;
        const o = i * K;                                          // counters, key set and arrival number now; id and field once the path is resolved
        for (let k = 0; k < K; k++) cols.clocks[o + k] = comps[k];
        cols.keysets[i] = ks;
        cols.setVal(i, this._nodeSeq++);
        rowEntry[i++] = ei;
      }
      keys.probeBlock(bPath, m);
      for (let x = 0; x < m; x++) {
        const j = first + x;
        if (keys.bCut[x] !== pCut || keys.bP1[x] !== pH1 || keys.bP2[x] !== pH2) {
          pCut = keys.bCut[x]; pH1 = keys.bP1[x]; pH2 = keys.bP2[x];
          parent = pCut < 0 ? "" : bPath[x].slice(0, pCut);
          clockField = keys.fieldOf(parent, NODE_CLOCK);
        }
        if (rowNode) rowNode[j] = keys.bIdx[x];
        cols.setKey(j, keys.bLo[x], keys.bHi[x], clockField);
        bPath[x] = undefined;
      }
    }
    return { used: cols.slice(i), rowEntry, rowNode, host, n: i };
  }

  /* what the device decided -> winners, flags, the store (opts.apply) */
  _finishVector(entries, p, r, opts = {}) {
    const t = this.vcTable, keys = t.keys, nat = t.native;
    const { used, rowEntry, rowNode, host } = p, i = p.n;
    const nw = r.updated.length;
    const appliedEntries = new Int32Array(nw);
    for (let k = 0; k < nw; k++) appliedEntries[k] = rowEntry[r.updated[k]];
    let nConflicts = 0;
    for (let j = 0; j < i; j++) if (r.flags[j] & nat.FLAG_CONCURRENT) nConflicts++;
    let broadcast;
    if (opts.apply && nw) broadcast = this._applyVectorWinners(entries, used, rowEntry, r, opts);
    if (opts.apply) this._unaliasLosers(entries, rowEntry, rowNode, i, r.updated, keys.size);
    if (opts.apply) this._notifyIndexHook(entries, host);
    const out = { appliedEntries, flags: r.flags, rowEntry: rowEntry.subarray(0, i), nApplied: nw, nConflicts, nRows: r.nRows, host, broadcast: opts.apply ? broadcast || [] : undefined };
    let list = null;
    Object.defineProperty(out, "applied", { enumerable: true, get() { if (!list) list = Array.from(appliedEntries, (ei) => ({ entry: ei, field: null })); return list; } });
    return out;
  }

  /* replay of what the device decided, node by node (see _mergeEntriesVector), and ONE application per changed node */
  _applyVectorWinners(entries, cols, rowEntry, r, opts) {
    const t = this.vcTable, nat = t.native, b = this.bullet;
    const working = new Map(), before = new Map();
    const clean = (d) => {
      if (!isMergeable(d)) return d;
      const c = {};
      for (const f of Object.keys(d)) if (f !== "__vectorClock" && f !== "__fromNetwork") c[f] = d[f];
      return c;
    };
    for (let j = 0; j < r.flags.length; j++) {
      const fl = r.flags[j];
      if (!(fl & (nat.FLAG_INCOMING | nat.FLAG_CONCURRENT))) continue;
      const e = entries[rowEntry[j]], path = e.path;
      let cur;
      if (working.has(path)) cur = working.get(path); else { cur = peek(b.store, path); before.set(path, cur); }
      if (!cur) cur = {};                                       // what handleUpdate's _getData(path) makes of a falsy value (src/bullet.js:115-129)
      const inc = clean(e.data);
      working.set(path, (fl & nat.FLAG_CONCURRENT) ? this.mergeValues(inc, cur) : inc);
    }
    const nw = r.updated.length;
    const ids = new BigUint64Array(nw), fields = new Uint32Array(nw);
    for (let k = 0; k < nw; k++) { ids[k] = cols.id[r.updated[k]]; fields[k] = cols.field[r.updated[k]]; }
    const got = r.rows || t.getRows(ids, fields);       // (the asynchronous merge brings the updated rows' clocks along: read right behind ITS merge, not behind a later one)
    const valueRows = opts.valueRows !== false;
    const q = valueRows ? this._putQueue() : null;
    const gk = valueRows ? this.graph.keys : null;
    const updates = new Array(nw);
    for (let k = 0; k < nw; k++) {
      const path = entries[rowEntry[r.updated[k]]].path;
      const clock = this._clockObject(got.clocks, k * t.K, got.keysets[k]);
      const value = working.get(path);
      updates[k] = { path, value, vectorClock: clock };
      this.vectorClocks.set(path, clock);
      if (valueRows) {
        const cut = path.lastIndexOf("/");
        gk.lookup(path);
        this._queueValueRows(q, path, cut < 0 ? "" : path.slice(0, cut), gk.lo, gk.hi, before.get(path), value, 0, true);
      }
    }
    if (q) q.closeBatch();
    if (opts.apply === "each") {
      if (typeof b._applyUpdate === "function") for (const u of updates) b._applyUpdate(u.path, u.value, u.vectorClock, true);
      return opts.broadcast !== false ? updates.map((u) => ({ path: u.path, broadcastData: isMergeable(u.value) ? Object.assign({}, u.value, { __vectorClock: u.vectorClock }) : u.value })) : [];
    }
    if (typeof b._applyBatch === "function") return b._applyBatch(updates, true) || [];
    return require("./batch-apply").applyBatch(b, updates, true, opts.broadcast !== false);
  }

  /** The clocks the vector-clock table holds for these node paths: [path | {path}] -> [vectorClock object (reference key order) | null]. */
  vcLookup(paths) {
    const t = this.vcTable, n = paths.length;
    this._flushVcPuts();
    const ids = new BigUint64Array(n), id32 = new Uint32Array(ids.buffer), fields = new Uint32Array(n);
    paths.forEach((k, i) => {
      const path = typeof k === "string" ? k : k.path;
      const cut = path.lastIndexOf("/");
      t.keys.lookup(path);
      id32[2 * i] = t.keys.lo; id32[2 * i + 1] = t.keys.hi;
      fields[i] = t.keys.fieldOf(cut < 0 ? "" : path.slice(0, cut), NODE_CLOCK);
    });
    const got = t.getRows(ids, fields);
    return paths.map((_, i) => (got.state[i] === t.native.VC_ABSENT ? null : this._clockObject(got.clocks, i * t.K, got.keysets[i])));
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
    // a path that holds a (non-zero) integer keeps it as the value of its clock row (ties on an integer path are decided by value, _kindAt)
    const prim = new Map();
    for (const r of rows) if (r.field === null && r.path !== null && r.path !== undefined && isDeviceInt(r.val) && r.val !== 0) prim.set(r.path, r.val);
    rows.forEach((r, i) => {
      const named = r.path !== null && r.path !== undefined;
      const id = named ? g.keys.idOf(r.path) : [Number(BigInt("0x" + r.id) & 0xffffffffn), Number(BigInt("0x" + r.id) >> 32n)];
      const pathNo = g.keys.idx;
      const f = r.collection !== null && r.collection !== undefined ? g.keys.fieldOf(r.collection, r.field) : r.fieldHash;
      let val = r.val;
      if (r.field === NODE_CLOCK) {
        const isInt = named && prim.has(r.path);
        val = isInt ? prim.get(r.path) : 0;       // nodes: arrival numbers start over — every write after the restore is later than the restored ones
        if (named) this._setKind(pathNo, isInt ? KIND_INT : KIND_NODE);
      }
      cols.set(i, id, f, r.ts, val);
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
