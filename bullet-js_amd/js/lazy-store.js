"use strict";
/*
 * lazy-store.js — opt-in (attach(bullet, {batchSync: {lazyStore: true}})): the winners of a device batch are RECORDED, and the facade's nested store,
 * meta and op log (src/bullet.js:28-33, written per winner by _applyUpdate :184-220) are brought up to date when somebody looks.
 *
 * Why: with the store kept, 1.29 of a sync entry's 1.62 us are the reference-shaped writes themselves — a property set on a collection object of a
 * million keys, an insert or three property writes on meta[path], the clock object, the update record (profiles/r04_e2e_apply.log) — and none of it is
 * needed to resolve the NEXT chunk: that needs the clock rows on the device and the path dictionary. A burst of chunks is then absorbed at the packer's
 * rate; the store follows when it is read or when the event loop is idle.
 *
 * How the state stays the reference's whenever it is OBSERVED through the facade: bullet.store / bullet.meta / bullet.log become accessors of this instance;
 * reading (or replacing) any of them first folds every recorded winner in, batch by batch and in arrival order, through the very code the eager path runs
 * (GpuCRT._foldRecords -> batch-apply.js) with the arrival time of each batch as its lastModified / log timestamp. Everything of the reference that reads
 * state goes through those three properties (Bullet._getData :115-129, _applyUpdate, the storage providers, the sync collector), as do GpuCRT's and GpuQuery's own
 * reads; device scans fold first as well (the winners' integer fields become device rows in the fold). Listeners are per-write callbacks: while any is
 * registered nothing is deferred. An idle fold runs when the event loop next turns (setImmediate), in slices, so the store never lags for long.
 *
 * The ONE observable difference, and why this is opt-in: an object handed out BEFORE a batch (bullet.get("users").value() returns the live collection
 * object, src/bullet.js:691-693) does not show that batch's nodes until the store has been read through the facade again or the idle fold has run.
 */
class LazyStore {
  constructor(bullet, crt, opts = {}) {
    this.bullet = bullet; this.crt = crt;
    this.n = 0;                       // winners recorded, not folded yet
    this.busy = false;                // folding (the accessors hand out the real objects)
    this.ent = []; this.idx = []; this.lo = []; this.hi = []; this.ts = [];
    this.batches = [];                // {end, now, writer, valueRows} per recorded batch, in arrival order
    this.folds = 0; this.folded = 0; this.recorded = 0;
    this.idleSlice = opts.idleSlice === undefined ? 50000 : opts.idleSlice;   // winners per idle slice (0: no idle folding)
    this._armed = false;
    this.real = { store: bullet.store, meta: bullet.meta, log: bullet.log };
    const self = this;
    for (const name of ["store", "meta", "log"]) {
      Object.defineProperty(bullet, name, {
        configurable: true, enumerable: true,
        get() { if (self.n && !self.busy) self.foldAll(); return self.real[name]; },
        set(v) { if (self.n && !self.busy) self.foldAll(); self.real[name] = v; },
      });
    }
  }
  /* may this batch be deferred? (per-write listeners want their callbacks at the write) */
  usable() {
    const ls = this.bullet.listeners;
    if (ls) for (const k in ls) { if (ls[k] && ls[k].length) return false; }
    return typeof this.bullet._applyBatch !== "function";
  }
  beginBatch(writer, valueRows) { this._cur = { end: this.ent.length, now: Date.now(), writer, valueRows }; }
  push(entry, idx, lo, hi, ts) { this.ent.push(entry); this.idx.push(idx); this.lo.push(lo); this.hi.push(hi); this.ts.push(ts); }
  endBatch() {
    const c = this._cur; this._cur = null;
    const added = this.ent.length - c.end;
    if (added === 0) return;
    c.end = this.ent.length;
    this.batches.push(c);
    this.n += added; this.recorded += added;
    if (this.idleSlice && !this._armed) { this._armed = true; setImmediate(() => this._idle()); }
  }
  _idle() {
    this._armed = false;
    if (!this.n || this.busy) return;
    this.fold(this.idleSlice);
    if (this.n && !this._armed) { this._armed = true; setImmediate(() => this._idle()); }
  }
  foldAll() { this.fold(Infinity); }
  /* fold whole batches, oldest first, until at least `atLeast` winners are in (a batch is never split: its winners' paths are distinct, which is what lets the
   * fold read every old value before it writes any) */
  fold(atLeast) {
    if (!this.n || this.busy) return;
    this.busy = true;
    try {
      let done = 0, from = 0, nb = 0;
      for (const bt of this.batches) {
        this.crt._foldRecords(this.ent, this.idx, this.lo, this.hi, this.ts, from, bt.end, bt.now, bt.writer, bt.valueRows);
        done += bt.end - from; from = bt.end; nb++;
        if (done >= atLeast) break;
      }
      this.folds++; this.folded += done; this.n -= done;
      if (nb === this.batches.length) { this.ent = []; this.idx = []; this.lo = []; this.hi = []; this.ts = []; this.batches = []; }
      else {
        this.ent = this.ent.slice(from); this.idx = this.idx.slice(from); this.lo = this.lo.slice(from); this.hi = this.hi.slice(from); this.ts = this.ts.slice(from);
        this.batches = this.batches.slice(nb);
        for (const bt of this.batches) bt.end -= from;
      }
    } finally { this.busy = false; }
  }
  uninstall() {
    this.foldAll();
    for (const name of ["store", "meta", "log"]) Object.defineProperty(this.bullet, name, { configurable: true, enumerable: true, writable: true, value: this.real[name] });
    this.idleSlice = 0;
  }
}

module.exports = LazyStore;
