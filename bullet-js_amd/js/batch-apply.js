"use strict";
/*
 * batch-apply.js — N1 (SURVEY §8(f)): the facade's per-write `_applyUpdate` (src/bullet.js:184-220) + `_notify` (:227-266),
 * done ONCE for a whole batch of winners instead of once per winner.
 *
 * The reference's per-write cost is dominated by this tail (39 % of its setData profile): an object spread for the meta entry,
 * `log.push` + `log.splice` on every write once the log holds 1000 entries, a path split per listener lookup and one parent-listener
 * call per ancestor per write. For a batch the same observable state needs:
 *   store  — each winner's leaf written (parents created on the way), in winner order;
 *   meta   — {...old, source, vectorClock, lastModified} per winner path (one clock read for the batch);
 *   log    — the last 1000 operations: when the batch alone has >= 1000 winners the log IS the batch's tail (no push/splice churn);
 *   notify — exact-path listeners per winner, in order; every ANCESTOR path's listeners once per batch, with the data the
 *            reference's last call would have passed (the state after the batch). Call COUNT for ancestors is the documented
 *            difference from the per-write loop; the data a listener last sees is the same;
 *   save   — the deferred storage save is (re)armed once (src/bullet.js:257-265).
 * Returns what `setData` would have handed to `network.broadcast` per winner (src/bullet-crt.js:371-376: objects get their clock
 * attached, primitives travel bare).
 */
const LOG_CAP = 1000;   // src/bullet.js:213-215

function applyBatch(bullet, updates, fromNetwork, wantBroadcast = true, at) {
  const n = updates.length;
  if (n === 0) return [];
  const now = at === undefined ? Date.now() : at;      // `at`: when the batch ARRIVED (a deferred fold, lazy-store.js, applies it later with its own time)
  const source = fromNetwork ? "network" : "local";
  const listeners = bullet.listeners || {};
  const hasListeners = Object.keys(listeners).length > 0;
  const parentsToNotify = hasListeners ? new Set() : null;
  const broadcast = wantBroadcast ? new Array(n) : null;   // remote writes are not re-broadcast (setData(path, data, false)): their caller skips it
  // parent object cache: consecutive winners of one node (its fields) share the walk
  let lastParentPath = null, lastParentNode = null;
  const entries = bullet.log && n < LOG_CAP ? null : [];
  for (let i = 0; i < n; i++) {
    const u = updates[i];
    const path = u.path;
    // u.parentHint / u.cutHint (optional, GpuCRT): the collection's path string shared by consecutive winners and where the path's last "/" is
    let cut = u.cutHint !== undefined ? u.cutHint : path.lastIndexOf("/");
    let parentPath = u.parentHint !== undefined ? u.parentHint : (cut < 0 ? "" : path.slice(0, cut));
    let key = u.keyHint !== undefined ? u.keyHint : (cut < 0 ? path : path.slice(cut + 1));
    if (u.keyHint === undefined && (!key || path.indexOf("//") >= 0)) {       // trailing or doubled slashes: the key is the last NON-EMPTY segment (src/bullet.js:186 path.split('/').filter(Boolean))
      const segs = path.split("/").filter(Boolean);
      key = segs.length ? segs.pop() : "";
      parentPath = segs.join("/");
    }
    let node;
    if (parentPath === lastParentPath) node = lastParentNode;
    else {
      node = bullet.store;
      if (parentPath) for (const seg of parentPath.split("/")) { if (!seg) continue; if (!node[seg]) node[seg] = {}; node = node[seg]; }
      lastParentPath = parentPath; lastParentNode = node;
    }
    if (!key) { if (broadcast) broadcast[i] = null; continue; }     // no segment at all: nothing to write (the reference's loop falls through the same way)
    node[key] = u.value;
    const old = bullet.meta[path];
    // {...old, source, vectorClock, lastModified} (src/bullet.js:196-201): the entry keeps whatever else somebody hung on it and gets these
    // three keys. An existing entry is updated in place: same content as the spread's copy, without re-inserting a key into a meta object of
    // millions of entries (2 us per write in V8's dictionary mode against 0.5 us; the entry OBJECT is then the same one as before the write)
    if (old === undefined) bullet.meta[path] = { source, vectorClock: u.vectorClock, lastModified: now };
    else { old.source = source; old.vectorClock = u.vectorClock; old.lastModified = now; }
    if (bullet.log && (!entries || i >= n - LOG_CAP)) {   // only the records that stay in the log are built
      const rec = { op: "set", path, data: u.value, vectorClock: u.vectorClock, timestamp: now };
      if (entries) entries.push(rec); else bullet.log.push(rec);
    }
    if (broadcast) {
      let b = u.value;
      if (typeof b === "object" && b !== null) b = Array.isArray(b) ? b.concat([{ __vectorClock: u.vectorClock }]) : Object.assign({}, b, { __vectorClock: u.vectorClock });
      broadcast[i] = { path, broadcastData: b };
    }
    if (hasListeners) {
      const ls = listeners[path];
      if (ls) for (const cb of ls) { try { cb(u.value); } catch (err) { console.error(`Error in listener callback for ${path}:`, err); } }
      let p = parentPath;
      for (;;) {
        if (parentsToNotify.has(p)) break;          // its ancestors are in the set already
        parentsToNotify.add(p);
        if (!p) break;
        const c = p.lastIndexOf("/");
        p = c < 0 ? "" : p.slice(0, c);
      }
    }
  }
  if (bullet.log) {
    if (entries) { bullet.log.length = 0; for (const r of entries) bullet.log.push(r); }      // the batch's tail IS the last 1000 operations
    else if (bullet.log.length > LOG_CAP) bullet.log.splice(0, bullet.log.length - LOG_CAP);  // one trim per batch
  }
  if (hasListeners) {
    // deepest first, like the reference's walk from the leaf upwards
    const ps = Array.from(parentsToNotify).sort((a, b) => b.length - a.length);
    for (const p of ps) {
      const ls = listeners[p];
      if (!ls) continue;
      const data = bullet._getData(p);
      for (const cb of ls) { try { cb(data); } catch (err) { console.error(`Error in parent listener callback for ${p}:`, err); } }
    }
  }
  if (bullet.storage && bullet.options && bullet.options.storageType !== "file") {
    clearTimeout(bullet._saveTimeout);
    bullet._saveTimeout = setTimeout(() => { bullet.storage.save(); }, 1000);
  }
  return broadcast ? broadcast.filter((x) => x !== null) : [];
}

module.exports = { applyBatch, LOG_CAP };
