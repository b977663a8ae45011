"use strict";
/*
 * device-graph.js — the GPU-resident graph plus the host dictionary that maps the device's hashed keys back to Bullet paths.
 * Shared by GpuCRT (merge) and GpuQuery (index scans). One shard (a bmx_ctx) by default; with `devices: [0, 1, ...]` or
 * `shards: N` the graph is split by node-id hash over N contexts owned by ONE handle (a bmx_comm): a delta is merged by the
 * shard that owns its node instead of by every peer (reference fan-out: src/bullet-network.js:378-418), scans run on every shard
 * and are concatenated (src/bullet-query.js:186-261). The methods below behave the same either way.
 */
const { requireNative } = require("./native");
const { KeyDictionary } = require("./hash");

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
  }
  mergeBatch(cols, mode) {
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
  loadRows(cols) { return this.comm ? this.native.commLoadRows(this.comm, cols.id, cols.field, cols.ts, cols.val) : this.native.loadRows(this.handle, cols.id, cols.field, cols.ts, cols.val); }
  getRows(id, field) { return this.comm ? this.native.commGetRows(this.comm, id, field) : this.native.getRows(this.handle, id, field); }
  rowCount() { return this.comm ? this.native.commRowCount(this.comm) : this.native.rowCount(this.handle); }
  dumpRows() { return this.comm ? this.native.commDumpRows(this.comm) : this.native.dumpRows(this.handle); }
  indexBuild(f) { return this.comm ? this.native.commIndexBuild(this.comm, f) : this.native.indexBuild(this.handle, f); }
  indexDrop(f) { return this.comm ? this.native.commIndexDrop(this.comm, f) : this.native.indexDrop(this.handle, f); }
  indexSize(f) { return this.comm ? this.native.commIndexSize(this.comm, f) : this.native.indexSize(this.handle, f); }
  /* {fullBuilds, incremental}: index rebuilds from the table vs updates from the merges' change log (one context only) */
  indexRefreshCounts() { return this.comm ? null : this.native.indexRefreshCounts(this.handle); }
  scanRange(f, lo, hi) { return this.comm ? this.native.commScanRange(this.comm, f, lo, hi) : this.native.scanRange(this.handle, f, lo, hi); }
  scanCount(f, lo, hi) { return this.comm ? this.native.commScanCount(this.comm, f, lo, hi) : this.native.scanCount(this.handle, f, lo, hi); }
  scanFilter(terms) { return this.comm ? this.native.commScanFilter(this.comm, terms) : this.native.scanFilter(this.handle, terms); }
  info() { return this.comm ? { nShards: this.nShards, devices: this.devices, nRows: this.rowCount() } : this.native.info(this.handle); }
  close() {
    if (this.handle) { this.native.destroy(this.handle); this.handle = null; }
    if (this.comm) { this.native.commDestroy(this.comm); this.comm = null; }
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
  /* node ids (BigUint64Array) of the rows of `field` with lo <= value <= hi: range()/equals() over the K-writer rows */
  scanRange(field, lo, hi) { return this.native.vcScanRange(this.handle, field, lo, hi); }
  close() {
    if (this.handle) { this.native.vcDestroy(this.handle); this.handle = null; }
  }
}

module.exports = DeviceGraph;
module.exports.DeviceVcTable = DeviceVcTable;
