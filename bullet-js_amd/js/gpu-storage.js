"use strict";
/*
 * gpu-storage.js — N3 (SURVEY §8(f)): a storage provider for the reference's one formal plug point
 *     new Bullet({ storage: true, storageType: GpuStorage, storagePath: dir, ... })       src/bullet.js:88-91
 * that persists and restores the graph in the reference's own on-disk shape (src/bullet-file-storage.js:96-210):
 *     <dir>/store.json   JSON.stringify(bullet.store)      the nested value tree
 *     <dir>/meta.json    JSON.stringify(bullet.meta)       {path: {source, vectorClock, lastModified}}
 *     <dir>/log.json     JSON.stringify(bullet.log)
 * A directory written by the reference's BulletFileStorage loads here and vice versa (encryption is not offered).
 *
 * What it adds is the DEVICE side of both directions:
 *   load  — after store/meta are in place, the device table is seeded from them when the graph is created (GpuCRT.seedDevice): every path with
 *           a clock gets its clock row (single-component clocks {<peer id>: ts}; any other clock marks the path host-only) and the value rows of
 *           its safe-integer fields, so a restart does not lose the resident graph nor the clocks later sync entries are resolved against;
 *   save  — rows that reached the device through typed columns only (no facade write) are folded back into store/meta first:
 *           the device table is dumped (bmx_dump_rows), value rows whose path is known to the key dictionary and whose value the store does
 *           not hold are written as leaves `<node path>/<field>` with clock {<peer id>: ts}, then the three files are written.
 * Interface of the reference's BulletStorage that the core calls: save(), close() (src/bullet.js:257-265, 288-304).
 */
const fs = require("fs");
const path = require("path");
const { Columns, isDeviceInt, scalarClock, NODE_CLOCK } = require("./hash");

class GpuStorage {
  constructor(bullet, options = {}) {
    this.bullet = bullet;
    this.options = Object.assign({ path: "./.bullet", saveInterval: 5000, encrypt: false, enableStorageLog: false }, options);
    if (this.options.encrypt) throw new Error("GpuStorage: encryption is not supported (use the reference's file storage for encrypted stores)");
    if (!fs.existsSync(this.options.path)) fs.mkdirSync(this.options.path, { recursive: true });
    this.loaded = this._loadData();
    this.saveInterval = null;
    if (this.options.saveInterval > 0) {
      this.saveInterval = setInterval(() => { this._saveData(); }, this.options.saveInterval);
      if (this.saveInterval.unref) this.saveInterval.unref();
    }
  }

  _emit(ev, arg) { if (this.bullet.middleware && this.bullet.middleware.emitEvent) this.bullet.middleware.emitEvent(ev, arg); }

  /* store.json / meta.json / log.json -> bullet.store / meta / log, as src/bullet-file-storage.js:96-163 does */
  _loadData() {
    const t0 = Date.now();
    let items = 0;
    try {
      const sp = path.join(this.options.path, "store.json");
      if (fs.existsSync(sp)) { const parsed = JSON.parse(fs.readFileSync(sp, "utf8")); deepMerge(this.bullet.store, parsed); items += Object.keys(parsed).length; }
      const mp = path.join(this.options.path, "meta.json");
      if (fs.existsSync(mp)) { const parsed = JSON.parse(fs.readFileSync(mp, "utf8")); Object.assign(this.bullet.meta, parsed); items += Object.keys(parsed).length; }
      const lp = path.join(this.options.path, "log.json");
      if (fs.existsSync(lp)) {
        const parsed = JSON.parse(fs.readFileSync(lp, "utf8"));
        this.bullet.log = (this.bullet.log || []).concat(parsed);
        if (this.bullet.log.length > 1000) this.bullet.log = this.bullet.log.slice(-1000);
        items += parsed.length;
      }
      this._emit("storage:load:complete", { store: this.bullet.store, duration: Date.now() - t0, items });
    } catch (err) {
      console.error("Error loading persisted data:", err);
      this._emit("storage:error", err);
    }
    return items;
  }

  /* rows of the device contract found in store/meta, as typed columns for bmx_load_rows. -> {cols, n} */
  deviceRows(keys) {
    const b = this.bullet, writer = (b.crt && b.crt._opts && b.crt._opts.writer) || b.id;
    const rows = [];
    for (const p of Object.keys(b.meta)) {
      const ts = scalarClock(b.meta[p].vectorClock, writer);
      if (ts < 0) continue;
      const v = readPath(b.store, p);
      const cut = p.lastIndexOf("/");
      if (isDeviceInt(v)) {                 // leaf entry: <node path>/<field>
        if (cut < 0) continue;
        const nodePath = p.slice(0, cut), c2 = nodePath.lastIndexOf("/");
        rows.push([nodePath, c2 < 0 ? "" : nodePath.slice(0, c2), p.slice(cut + 1), ts, v]);
      } else if (v && typeof v === "object" && !Array.isArray(v)) {   // node entry: every integer field shares the node's clock
        const fields = Object.keys(v);
        if (!fields.length || !fields.every((f) => isDeviceInt(v[f]))) continue;
        for (const f of fields) if (!(b.meta[p + "/" + f])) rows.push([p, cut < 0 ? "" : p.slice(0, cut), f, ts, v[f]]);
      }
    }
    const cols = new Columns(rows.length);
    rows.forEach((r, i) => cols.set(i, keys.idOf(r[0]), keys.fieldOf(r[1], r[2]), r[3], r[4]));
    return { cols, n: rows.length };
  }

  /* preload the device table from what was loaded (attach() does it once the graph exists; by hand for a graph of one's own) */
  restoreDevice(graph) {
    const crt = this.bullet.crt;
    if (crt && typeof crt.seedDevice === "function" && crt._graph === graph) return crt.seedDevice();
    const { cols, n } = this.deviceRows(graph.keys);
    if (n) graph.loadRows(cols);
    return n;
  }

  /* device rows that the JS store does not reflect yet -> store/meta leaves */
  foldDevice(graph) {
    const b = this.bullet, writer = (b.crt && b.crt._opts && b.crt._opts.writer) || b.id;
    if (b.crt && typeof b.crt._flushDeviceWrites === "function") b.crt._flushDeviceWrites();   // single writes still queued for the device
    const d = graph.dumpRows();
    const id32 = new Uint32Array(d.id.buffer, d.id.byteOffset, d.id.length * 2);
    let folded = 0;
    for (let i = 0; i < d.id.length; i++) {
      const nodePath = graph.keys.pathOf(id32[2 * i], id32[2 * i + 1]);
      const f = graph.keys.fields.get(d.field[i]);
      if (nodePath === undefined || !f) continue;          // raw hashed keys: nothing to call them in the store
      if (f[1] === NODE_CLOCK) continue;                   // a node's clock row: meta[path] is where the facade keeps it
      const leaf = f[1] === null ? nodePath : nodePath + "/" + f[1];
      const ts = Number(d.ts[i]), val = Number(d.val[i]);
      if (readPath(b.store, leaf) === val) continue;       // the store holds it: a row that mirrors a facade write (its clock lives in the node's meta entry)
      const m = b.meta[leaf];
      writePath(b.store, leaf, val);
      const clock = {}; clock[writer] = ts;
      b.meta[leaf] = Object.assign({}, m || {}, { source: (m && m.source) || "network", vectorClock: clock, lastModified: (m && m.lastModified) || Date.now() });
      folded++;
    }
    return folded;
  }

  _saveData() {
    try {
      this._emit("storage:save:start");
      const g = this.bullet.crt && this.bullet.crt._graph;       // only if the device was ever used
      if (g) this.foldDevice(g);
      fs.writeFileSync(path.join(this.options.path, "store.json"), JSON.stringify(this.bullet.store));
      fs.writeFileSync(path.join(this.options.path, "meta.json"), JSON.stringify(this.bullet.meta));
      fs.writeFileSync(path.join(this.options.path, "log.json"), JSON.stringify(this.bullet.log || []));
      this._emit("storage:save:complete");
    } catch (err) {
      console.error("Error saving data:", err);
      this._emit("storage:error", err);
    }
    return Promise.resolve();
  }

  save() { return this._saveData(); }
  close() {
    if (this.saveInterval) { clearInterval(this.saveInterval); this.saveInterval = null; }
    return this._saveData();
  }
}

function readPath(root, p) {
  let cur = root;
  for (const seg of p.split("/")) { if (!seg) continue; if (cur === null || typeof cur !== "object" || !(seg in cur)) return undefined; cur = cur[seg]; }
  return cur;
}
function writePath(root, p, v) {
  const segs = p.split("/").filter(Boolean);
  let cur = root;
  for (const seg of segs.slice(0, -1)) { if (!cur[seg] || typeof cur[seg] !== "object") cur[seg] = {}; cur = cur[seg]; }
  cur[segs[segs.length - 1]] = v;
}
function deepMerge(target, src) {   // src/bullet-storage.js _deepMerge: objects merge, everything else replaces
  for (const k of Object.keys(src)) {
    const v = src[k];
    if (v && typeof v === "object" && !Array.isArray(v)) {
      if (!target[k] || typeof target[k] !== "object" || Array.isArray(target[k])) target[k] = {};
      deepMerge(target[k], v);
    } else target[k] = v;
  }
  return target;
}

module.exports = GpuStorage;
