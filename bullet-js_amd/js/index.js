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
const hash = require("./hash");
const native = require("./native");

/** Install GpuCRT and GpuQuery on a Bullet instance; one DeviceGraph is shared and created on first device use. */
function attach(bullet, opts = {}) {
  const shared = Object.assign({}, opts);
  let graph = opts.graph || null;
  const lazy = {
    get graph() {
      if (!graph) { const DeviceGraph = require("./device-graph"); graph = new DeviceGraph(shared); }
      return graph;
    },
  };
  const crt = new GpuCRT(bullet, opts);
  Object.defineProperty(crt, "graph", { get: () => lazy.graph });
  bullet.crt = crt;
  const query = new GpuQuery(bullet, opts);
  Object.defineProperty(query, "graph", { get: () => lazy.graph });
  bullet.query = query;
  const close = bullet.close ? bullet.close.bind(bullet) : null;
  bullet.close = async function () {
    if (graph) { graph.close(); graph = null; }
    if (close) return close();
  };
  return { crt, query };
}

module.exports = { GpuCRT, GpuQuery, attach, hash, nativeAvailable: native.available };
