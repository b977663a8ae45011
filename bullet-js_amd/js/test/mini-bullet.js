"use strict";
/*
 * mini-bullet.js — TEST HARNESS ONLY: the smallest host object GpuCRT/GpuQuery can be plugged into on a box
 * that has no reference checkout (the GPU box). It provides the members the two classes touch
 * (id, store, meta, log, _getData, setData, get) with the reference facade's observable behaviour for them
 * (src/bullet.js:106-129, 139-220): path autovivification on read, whole-value replace on apply, op log capped
 * at 1000, meta {source, vectorClock, lastModified}. It is not shipped and not a Bullet replacement.
 */
class MiniNode {
  constructor(db, path) { this.bullet = db; this.path = path; }
  value() { return this.bullet._getData(this.path); }
  put(data) { this.bullet.setData(this.path, data); return this; }
}

class MiniBullet {
  constructor(id) {
    this.id = id || "w";
    this.store = {};
    this.meta = {};
    this.log = [];
    this.crt = null;
    this.query = null;
  }
  get(path) { return new MiniNode(this, path); }
  _getData(path) {
    if (!path) return this.store;
    let cur = this.store;
    for (const seg of path.split("/")) {
      if (!seg) continue;
      if (!cur[seg]) cur[seg] = {};
      cur = cur[seg];
    }
    return cur;
  }
  setData(path, raw) {
    let data = raw, fromNetwork = false;
    if (raw && typeof raw === "object" && raw.__fromNetwork) {
      fromNetwork = true;
      if (Array.isArray(raw)) data = raw.slice();
      else { data = {}; for (const k of Object.keys(raw)) if (k !== "__fromNetwork") data[k] = raw[k]; }
    }
    const r = this.crt.handleUpdate(path, data, fromNetwork);
    if (!r.doUpdate) return r.value;
    const segs = path.split("/").filter(Boolean);
    let node = this.store;
    for (const seg of segs.slice(0, -1)) { if (!node[seg]) node[seg] = {}; node = node[seg]; }
    const leaf = segs[segs.length - 1];
    if (leaf) {
      node[leaf] = r.value;
      this.meta[path] = Object.assign({}, this.meta[path] || {}, { source: fromNetwork ? "network" : "local", vectorClock: r.vectorClock, lastModified: Date.now() });
      this.log.push({ op: "set", path, data: r.value, vectorClock: r.vectorClock, timestamp: Date.now() });
      if (this.log.length > 1000) this.log.splice(0, this.log.length - 1000);
    }
    return r.value;
  }
}

module.exports = MiniBullet;
