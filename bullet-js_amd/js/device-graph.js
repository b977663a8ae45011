"use strict";
/*
 * device-graph.js — one GPU-resident graph shard (a bmx_ctx) plus the host dictionary that maps the
 * device's hashed keys back to Bullet paths. Shared by GpuCRT (merge) and GpuQuery (index scans).
 */
const { requireNative } = require("./native");
const { KeyDictionary } = require("./hash");

class DeviceGraph {
  constructor(opts = {}) {
    this.native = requireNative();                 // throws if the addon is missing: no CPU fallback
    this.device = opts.device || 0;
    this.capacityRows = opts.capacityRows || (1 << 16);   // grows on demand (bmx_reserve / automatic rehash)
    this.handle = this.native.create(this.device, this.capacityRows);
    this.keys = new KeyDictionary();
    this.batches = 0;
  }
  mergeBatch(cols, mode) {
    this.batches++;
    return this.native.mergeBatch(this.handle, cols.id, cols.field, cols.ts, cols.val, mode | 0);
  }
  mergeBatchAsync(cols, mode) {
    this.batches++;
    return this.native.mergeBatchAsync(this.handle, cols.id, cols.field, cols.ts, cols.val, mode | 0);
  }
  reserve(capacityRows) { this.native.reserve(this.handle, capacityRows); }
  loadRows(cols) { this.native.loadRows(this.handle, cols.id, cols.field, cols.ts, cols.val); }
  getRows(id, field) { return this.native.getRows(this.handle, id, field); }
  rowCount() { return this.native.rowCount(this.handle); }
  dumpRows() { return this.native.dumpRows(this.handle); }
  indexBuild(f) { this.native.indexBuild(this.handle, f); }
  indexDrop(f) { this.native.indexDrop(this.handle, f); }
  indexSize(f) { return this.native.indexSize(this.handle, f); }
  scanRange(f, lo, hi) { return this.native.scanRange(this.handle, f, lo, hi); }
  scanCount(f, lo, hi) { return this.native.scanCount(this.handle, f, lo, hi); }
  scanFilter(terms) { return this.native.scanFilter(this.handle, terms); }
  info() { return this.native.info(this.handle); }
  close() {
    if (this.handle) { this.native.destroy(this.handle); this.handle = null; }
  }
}

/*
 * DeviceVcTable — N4 (SURVEY §8(f)): rows with a K-writer vector clock (a bmx_vc). `writers` fixes the component order;
 * `local` is this peer's id (the reference stores clock {local: 2} for a first write). Host typed arrays in and out.
 */
class DeviceVcTable {
  constructor(writers, local, opts = {}) {
    this.native = requireNative();
    if (!Array.isArray(writers) || writers.length < 1 || writers.length > this.native.VC_MAX_WRITERS || writers.indexOf(local) < 0) {
      const err = new Error("bmx: vector-clock mode needs 1.." + this.native.VC_MAX_WRITERS + " writer ids that include this peer's id");
      err.code = "BMX_BAD_WRITERS";
      throw err;
    }
    this.writers = writers.slice();
    this.K = writers.length;
    this.local = writers.indexOf(local);
    this.handle = this.native.vcCreate(opts.device || 0, opts.vcCapacityRows || opts.capacityRows || (1 << 16), this.K, this.local);
    this.keys = new KeyDictionary();
  }
  loadRows(c) { this.native.vcLoadRows(this.handle, c.id, c.field, c.clocks, c.val); }
  mergeBatch(c) { return this.native.vcMergeBatch(this.handle, c.id, c.field, c.clocks, c.val); }
  getRows(id, field) { return this.native.vcGetRows(this.handle, id, field); }
  rowCount() { return this.native.vcRowCount(this.handle); }
  close() {
    if (this.handle) { this.native.vcDestroy(this.handle); this.handle = null; }
  }
}

module.exports = DeviceGraph;
module.exports.DeviceVcTable = DeviceVcTable;
