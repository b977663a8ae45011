"use strict";
/*
 * bullet-js_amd/js — host side of the MI355X engine in the reference's own language.
 *
 *   const Bullet = require("@korandi/bullet-js");
 *   const { attach } = require("bullet-js_amd/js");
 *   const bullet = new Bullet({ disableCRT: true, enableIndexing: false, ...opts });   // the two seams: src/bullet.js:46-48, 62-64
 *   attach(bullet, { device: 0, capacityRows: 16e6 });                                   // bullet.crt = GpuCRT, bullet.query = GpuQuery
 */
const GpuCRT = require("./gpu-crt");
const GpuQuery = require("./gpu-query");
const GpuStorage = require("./gpu-storage");
const { installBatchSync } = require("./batch-sync");
const { applyBatch } = require("./batch-apply");
const hash = require("./hash");
const native = require("./native");

/** Install GpuCRT and GpuQuery on a Bullet instance; one DeviceGraph is shared and created on first device use.
 *  opts.batchSync: also route the facade's remote-write ingestion (sync chunks, network puts: src/bullet-network-sync.js:551-569,
 *  src/bullet-network.js:332-346) through the batch path (batch-sync.js). If the instance was created with
 *  `storageType: GpuStorage`, the rows that storage loaded from disk are preloaded into the device table when the graph is created (as is
 *  everything else the facade holds by then: GpuCRT.seedDevice). */
function attach(bullet, opts = {}) {
  const shared = Object.assign({}, opts);
  let graph = opts.graph || null;
  const lazy = {
    get graph() {
      if (!graph) {
        const DeviceGraph = require("./device-graph");
        graph = new DeviceGraph(shared);
        crt.seedDevice();        // what the facade already holds (earlier writes, a store loaded from disk: GpuStorage): clocks and integer values -> device rows
      }
      return graph;
    },
  };
  // with the sync adapter primitives are local writes (batch-sync.js): no integer entry is ever the device's, every path is a node path
  const crt = new GpuCRT(bullet, opts.batchSync && opts.integerEntries === undefined ? Object.assign({}, opts, { integerEntries: false }) : opts);
  Object.defineProperty(crt, "graph", { get: () => lazy.graph });
  bullet.crt = crt;
  const query = new GpuQuery(bullet, opts);
  Object.defineProperty(query, "graph", { get: () => lazy.graph });
  bullet.query = query;
  Object.defineProperty(crt, "_graph", { get: () => graph, set: (g) => { graph = g; }, configurable: true });
  const sync = opts.batchSync ? installBatchSync(bullet, crt, typeof opts.batchSync === "object" ? opts.batchSync : {}) : null;
  // opt-in: store, meta and op log follow the batches lazily (lazy-store.js; one documented difference: live objects handed out before a batch)
  const lazyStore = opts.batchSync && typeof opts.batchSync === "object" && opts.batchSync.lazyStore ? new (require("./lazy-store"))(bullet, crt, typeof opts.batchSync.lazyStore === "object" ? opts.batchSync.lazyStore : {}) : null;
  if (lazyStore) crt._lazy = lazyStore;
  const close = bullet.close ? bullet.close.bind(bullet) : null;
  bullet.close = async function () {
    if (sync) sync.uninstall();
    if (lazyStore) { lazyStore.uninstall(); crt._lazy = null; }
    let r;
    if (close) r = await close();          // storage.close() saves first: it may still need the device rows
    if (graph) { graph.close(); graph = null; }
    return r;
  };
  return { crt, query, sync, lazyStore };
}

module.exports = { GpuCRT, GpuQuery, GpuStorage, attach, installBatchSync, applyBatch, hash, nativeAvailable: native.available };
