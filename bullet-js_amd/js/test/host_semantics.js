"use strict";
/*
 * host_semantics.js — CPU-only test of GpuCRT's single-operation path and GpuQuery's host-side behaviour against
 * golden vectors produced by the real reference (tests/golden/g1, g3, g4, g5_query_example).
 * If a reference checkout is available ($BULLET_REF or /root/reference) the same scripts are also replayed
 * through the REAL Bullet facade with GpuCRT plugged in at the disableCRT seam.
 * Usage: node host_semantics.js <golden dir>
 */
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const { GpuCRT, GpuQuery } = require("..");
const MiniBullet = require("./mini-bullet");

const GOLD = process.argv[2] || path.join(__dirname, "..", "..", "..", "tests", "golden");
const load = (n) => JSON.parse(fs.readFileSync(path.join(GOLD, n), "utf8"));
const flagsOf = (d) => (d.incoming ? 1 : 0) | (d.current ? 2 : 0) | (d.historical ? 4 : 0) | (d.concurrent ? 8 : 0);
let checks = 0;

function newCrt() { return new GpuCRT({ id: "w", meta: {}, _getData() { return undefined; } }); }

/* G1: decision table through processUpdate */
for (const c of load("g1_decision_table.json").cases) {
  const crt = newCrt();
  const r = crt.processUpdate("k", c.inc[1], { w: c.inc[0] }, c.cur ? c.cur[1] : undefined, c.cur ? { w: c.cur[0] } : undefined);
  assert.strictEqual(flagsOf(r.decision), c.flags, JSON.stringify(c));
  assert.deepStrictEqual([r.vectorClock.w, r.value], c.out, JSON.stringify(c));
  assert.strictEqual(r.decision.reason, c.reason);
  checks++;
}

/* G3: sequences on one key, caller applies the store rule */
for (const s of load("g3_sequences.json").seqs) {
  const crt = newCrt();
  let cur = s.start ? { value: s.start[1], clock: { w: s.start[0] } } : null;
  s.deltas.forEach((d, j) => {
    const r = crt.processUpdate("k", d[1], { w: d[0] }, cur ? cur.value : undefined, cur ? cur.clock : undefined);
    assert.strictEqual(flagsOf(r.decision), s.flags[j], JSON.stringify(s));
    if (r.decision.incoming || !cur || r.decision.concurrent) cur = { value: r.value, clock: r.vectorClock };
  });
  assert.deepStrictEqual([cur.clock.w, cur.value], s.final);
  checks++;
}

/* G4: node-level operations through a facade (mini harness, and the real Bullet when present) */
function replayL1(makeBullet, label) {
  const b = makeBullet();
  for (const st of load("g4_l1_ops.json").steps) {
    const ret = b.setData(st.path, JSON.parse(JSON.stringify(st.data)), false);
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), st.store, label + " store after " + st.path);
    assert.deepStrictEqual(ret === undefined ? null : JSON.parse(JSON.stringify(ret)), st.ret, label + " return of " + st.path);
    assert.deepStrictEqual(b.meta[st.path] ? JSON.parse(JSON.stringify(b.meta[st.path].vectorClock)) : null, st.clock, label + " clock of " + st.path);
    assert.strictEqual(b.meta[st.path] ? b.meta[st.path].source : null, st.source);
    assert.strictEqual(b.log.length, st.log_len);
    checks++;
  }
}
replayL1(() => { const b = new MiniBullet("w"); b.crt = new GpuCRT(b); return b; }, "mini");

/* helpers of the public surface */
{
  const crt = newCrt();
  assert.strictEqual(crt.compareVectorClocks({ a: 1, b: 2 }, { a: 2, b: 1 }), 0);
  assert.strictEqual(crt.compareVectorClocks({ a: 2 }, { a: 1, b: 0 }), 1);
  assert.strictEqual(crt.compareVectorClocks(undefined, { a: 1 }), -1);
  assert.strictEqual(crt.compareVectorClocks({ a: 1 }, null), 1);
  assert.deepStrictEqual(crt.mergeVectorClocks({ a: 1, b: 5 }, { a: 3, c: 2 }), { a: 3, b: 5, c: 2 });
  assert.deepStrictEqual(crt.mergeVectorClocks(null, { a: 3 }), { a: 3 });
  assert.deepStrictEqual(crt.mergeValues({ x: 1, n: { p: 1 } }, { x: 2, y: 3, n: { q: 2 } }), { x: 2, y: 3, n: { q: 2, p: 1 } });
  assert.deepStrictEqual(crt.mergeValues([1], [2]), [2]);       // arrays are not merged: compared as values ('1' < '2' -> current)
  assert.deepStrictEqual(crt.mergeValues([3], [2]), [3]);
  assert.strictEqual(crt.formatClock({ a: 1, b: 2 }), "a:1, b:2");
  assert.strictEqual(crt.formatClock(null), "null");
  const u = crt.createUpdate("z", 5);
  assert.deepStrictEqual(u, { value: 5, vectorClock: { w: 2 } });
  assert.strictEqual(crt.setCompare(() => 0), crt);
  checks += 10;
}

/* G5 (host side): string/boolean indices and reference ordering without a GPU */
function queryExample(makeBullet, label) {
  const g = load("g5_query_example.json");
  const b = makeBullet();
  for (const [k, v] of Object.entries(g.users)) b.get("users/" + k).put(v);
  for (const [k, v] of Object.entries(g.products)) b.get("products/" + k).put(v);
  const keys = (nodes) => nodes.map((n) => n.path.split("/").pop());
  assert.deepStrictEqual(keys(b.query.equals("users", "role", "admin")), ["user1", "user6", "user10"], label);
  assert.strictEqual(b.query.lastPath, "host");
  assert.deepStrictEqual(keys(b.query.equals("users", "active", false)), ["user3", "user6", "user9"], label);
  assert.strictEqual(b.query.count("products", "category", "electronics"), 5);
  assert.deepStrictEqual(keys(b.query.filter("products", (p) => p.price >= 200 && p.stock <= 12)), ["prod7", "prod8", "prod9", "prod10"]);
  assert.strictEqual(b.query.find("users", (u) => u.age > 40).path, "users/user3");
  assert.deepStrictEqual(b.query.map("users", (u) => u.age).slice(0, 3), [28, 35, 42]);
  assert.ok("users:role" in b.query.indices && "users:active" in b.query.indices);
  checks += 7;
  return b;
}
queryExample(() => { const b = new MiniBullet("w"); b.crt = new GpuCRT(b); b.query = new GpuQuery(b); return b; }, "mini");

/* N1: one batched apply pass == the facade's per-write _applyUpdate loop (store, meta, log tail, what listeners see) */
const { applyBatch, GpuStorage } = require("..");
function perWriteApply(b, path, value, vectorClock, fromNetwork) {      // the facade's tail as MiniBullet/the reference run it per write
  if (typeof b._applyUpdate === "function") return b._applyUpdate(path, value, vectorClock, fromNetwork);
  const segs = path.split("/").filter(Boolean); let node = b.store;
  for (const sgm of segs.slice(0, -1)) { if (!node[sgm]) node[sgm] = {}; node = node[sgm]; }
  node[segs[segs.length - 1]] = value;
  b.meta[path] = Object.assign({}, b.meta[path] || {}, { source: fromNetwork ? "network" : "local", vectorClock, lastModified: Date.now() });
  b.log.push({ op: "set", path, data: value, vectorClock, timestamp: Date.now() });
  if (b.log.length > 1000) b.log.splice(0, b.log.length - 1000);
}
function batchApplyCase(makeBullet, label, n) {
  let s = 777; const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
  const updates = [];
  for (let i = 0; i < n; i++) updates.push({ path: "n/k" + (rnd() % 300) + "/" + ["age", "score", "hits"][rnd() % 3], value: (rnd() % 2001) - 1000, vectorClock: { w: 2 + (rnd() % 500) } });
  for (let i = 50; i < n; i += 97) updates[i].path = i % 2 ? updates[i].path + "/" : updates[i].path.replace("n/", "n//");   // trailing / doubled slashes: the key is the last non-empty segment (src/bullet.js:186)
  const a = makeBullet(), b = makeBullet();
  const seenA = { leaf: [], node: [], root: [] }, seenB = { leaf: [], node: [], root: [] };
  for (const [bb, seen] of [[a, seenA], [b, seenB]]) {
    bb.listeners = bb.listeners || {};
    bb.listeners[updates[0].path] = [(v) => seen.leaf.push(v)];
    bb.listeners["n/" + updates[0].path.split("/")[1]] = [(v) => seen.node.push(JSON.stringify(v))];
    bb.listeners["n"] = [(v) => seen.root.push(Object.keys(v).length)];
  }
  for (const u of updates) perWriteApply(a, u.path, u.value, u.vectorClock, true);
  const out = applyBatch(b, updates, true);
  assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), JSON.parse(JSON.stringify(a.store)), label + " store");
  const strip = (m) => { const o = {}; for (const k of Object.keys(m)) o[k] = { source: m[k].source, vectorClock: m[k].vectorClock }; return o; };
  assert.deepStrictEqual(strip(b.meta), strip(a.meta), label + " meta");
  const ops = (l) => l.map((e) => [e.op, e.path, e.data, e.vectorClock.w]);
  assert.deepStrictEqual(ops(b.log), ops(a.log), label + " log tail");
  assert.ok(b.log.length <= 1000);
  assert.strictEqual(out.length, n);
  assert.deepStrictEqual(out[n - 1], { path: updates[n - 1].path, broadcastData: updates[n - 1].value });
  if (typeof a._notify === "function" || a.listeners) {   // a facade that notifies: exact-path listeners see every write, ancestors end on the same data
    if (seenA.leaf.length) assert.deepStrictEqual(seenB.leaf, seenA.leaf, label + " leaf listener");
    if (seenA.node.length) { assert.strictEqual(seenB.node[seenB.node.length - 1], seenA.node[seenA.node.length - 1]); assert.strictEqual(seenB.node.length, 1); }
    if (seenA.root.length) { assert.strictEqual(seenB.root[seenB.root.length - 1], seenA.root[seenA.root.length - 1]); assert.strictEqual(seenB.root.length, 1); }
  }
  checks += 6;
}
batchApplyCase(() => new MiniBullet("w"), "mini 400", 400);
batchApplyCase(() => new MiniBullet("w"), "mini 2500", 2500);     // more winners than the log keeps

/* N3 (file side, no GPU): a directory written by the reference's BulletFileStorage loads through GpuStorage, yields the device rows
 * of the contract, and is written back in the same shape */
{
  const os = require("os");
  const src = path.join(GOLD, "g7_storage_dir");
  const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "bmx-n3-"));
  for (const f of ["store.json", "meta.json"]) fs.copyFileSync(path.join(src, f), path.join(tmp, f));
  const b = new MiniBullet("w");
  b.crt = new GpuCRT(b);
  const st = new GpuStorage(b, { path: tmp, saveInterval: 0 });
  const wantStore = JSON.parse(fs.readFileSync(path.join(src, "store.json"), "utf8")), wantMeta = JSON.parse(fs.readFileSync(path.join(src, "meta.json"), "utf8"));
  assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), wantStore);
  assert.deepStrictEqual(JSON.parse(JSON.stringify(b.meta)), wantMeta);
  const { KeyDictionary } = require("../hash");
  const { cols, n } = st.deviceRows(new KeyDictionary());
  let expect = 0;
  for (const p of Object.keys(wantMeta)) {
    const v = p.split("/").reduce((o, k) => (o === undefined ? o : o[k]), wantStore);
    const c = wantMeta[p].vectorClock;
    if (Object.keys(c).length !== 1 || c.w === undefined) continue;
    if (Number.isInteger(v)) expect += 1; else if (v && typeof v === "object" && Object.values(v).every(Number.isInteger)) expect += Object.keys(v).length;
  }
  assert.strictEqual(n, expect);
  assert.ok(n >= 80);
  assert.strictEqual(Number(cols.ts[0]), wantMeta["n/k0"].vectorClock.w);
  st.save();
  assert.deepStrictEqual(JSON.parse(fs.readFileSync(path.join(tmp, "store.json"), "utf8")), wantStore, "round trip store.json");
  assert.deepStrictEqual(JSON.parse(fs.readFileSync(path.join(tmp, "meta.json"), "utf8")), wantMeta, "round trip meta.json");
  st.close();
  checks += 7;
}

/* With the real reference present: plug into the real Bullet at its two seams */
const REF = process.env.BULLET_REF || "/root/reference";
if (fs.existsSync(path.join(REF, "src", "bullet.js"))) {
  const Bullet = require(path.join(REF, "src", "bullet.js"));
  const quiet = (fn) => { const l = console.log; console.log = () => {}; try { return fn(); } finally { console.log = l; } };
  const mk = (withQuery) => quiet(() => {
    const b = new Bullet({ disableNetwork: true, storage: false, server: false, enableMiddleware: false, enableValidation: false,
      enableSerializer: false, enableIndexing: false, disableCRT: true });
    b.id = "w";
    b.crt = new GpuCRT(b);
    if (withQuery) b.query = new GpuQuery(b);
    return b;
  });
  replayL1(() => mk(false), "real-bullet");
  quiet(() => queryExample(() => mk(true), "real-bullet"));
  quiet(() => batchApplyCase(() => mk(false), "real-bullet 2500", 2500));     // against the reference's own _applyUpdate/_notify
  quiet(() => g9OnTheHost(() => mk(false), "real-bullet"));
  quiet(() => g9OnTheHost(() => mk(false), "real-bullet", "g10_sync_mixed_values.json"));
  {
    // the reference's file storage reads what GpuStorage wrote, and the real facade accepts GpuStorage at its provider hook
    const os = require("os");
    const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "bmx-n3r-"));
    const b1 = quiet(() => new Bullet({ disableNetwork: true, storage: true, storageType: GpuStorage, storagePath: tmp, saveInterval: 0, server: false, enableIndexing: false, disableCRT: true }));
    b1.id = "w"; b1.crt = new GpuCRT(b1);
    assert.ok(b1.storage instanceof GpuStorage);
    quiet(() => { b1.setData("n/a", { x: 1, y: 2, __fromNetwork: true, __vectorClock: { w: 9 } }, false); b1.setData("cfg/t", "s", false); });
    b1.storage.save();
    const b2 = quiet(() => new Bullet({ disableNetwork: true, storage: true, storageType: "file", storagePath: tmp, saveInterval: 0, server: false, enableIndexing: false }));
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b2.store)), JSON.parse(JSON.stringify(b1.store)));
    assert.deepStrictEqual(b2.meta["n/a"].vectorClock, b1.meta["n/a"].vectorClock);
    if (b2.storage.saveInterval) clearInterval(b2.storage.saveInterval);
    checks += 3;
  }
  console.log("reference facade present: replayed through the real Bullet with GpuCRT/GpuQuery plugged in");
}

/* g9 (host side, no GPU): the reference's sync loop entry by entry through GpuCRT.handleUpdate — the host resolver that takes every entry the device
 * contract leaves out — reproduces the reference's node-level outcomes: store and every clock + source after each chunk, then the queries */
function g9OnTheHost(makeBullet, label, fixture = "g9_sync_node_semantics.json") {
  const g = load(fixture);
  label += " " + fixture.slice(0, fixture.indexOf("_"));
  const b = makeBullet();
  g.chunks.forEach((chunk, ci) => {
    for (const e of JSON.parse(JSON.stringify(chunk))) {              // the loop body of src/bullet-network-sync.js:552-568
      if (e.deleted) b.setData(e.path, null, false);
      else b.setData(e.path, typeof e.data === "object" && e.data !== null ? Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }) : e.data, false);
    }
    const want = g.after[ci];
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), want.store, label + ": store after chunk " + (ci + 1));
    assert.deepStrictEqual(Object.keys(b.meta).sort(), Object.keys(want.meta).sort(), label);
    for (const p of Object.keys(want.meta)) {
      assert.deepStrictEqual(JSON.parse(JSON.stringify(b.meta[p].vectorClock)), want.meta[p].vectorClock, label + ": clock of " + p + " after chunk " + (ci + 1));
      assert.strictEqual(b.meta[p].source, want.meta[p].source, label + ": source of " + p);
      checks++;
    }
  });
  return b;
}
g9OnTheHost(() => { const b = new MiniBullet("w"); b.crt = new GpuCRT(b); return b; }, "mini");
/* g12 (host side): multi-writer clocks, node level — the host resolver alone reproduces the reference's stores and clocks, key ORDER included */
{
  const g = load("g12_vc_node_semantics.json");
  const b = new MiniBullet("w"); b.crt = new GpuCRT(b);
  g.chunks.forEach((chunk, ci) => {
    for (const e of JSON.parse(JSON.stringify(chunk))) {
      if (e.deleted) b.setData(e.path, null, false);
      else b.setData(e.path, Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }), false);
    }
    const want = g.after[ci];
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), want.store, "g12 (host) store after chunk " + (ci + 1));
    for (const p of Object.keys(want.meta)) {
      assert.deepStrictEqual(Object.keys(b.meta[p].vectorClock).map((w) => [w, b.meta[p].vectorClock[w]]), want.meta[p].clock, "g12 (host) clock of " + p + " after chunk " + (ci + 1));
      checks++;
    }
  });
}
g9OnTheHost(() => { const b = new MiniBullet("w"); b.crt = new GpuCRT(b); return b; }, "mini", "g10_sync_mixed_values.json");   // fields of every JSON type, ties against stored strings / numbers / null

/* N2 and the wrappers around setData: with put middleware registered nothing is batched (every entry takes setData, one by one);
 * without it the device-eligible run goes to mergeEntries, and the query engine's index hook is told about every batched entry */
{
  const { installBatchSync } = require("../batch-sync");
  const GpuCRTc = require("../gpu-crt");
  const seen = [];
  const fake = { id: "w", meta: {}, store: {}, setData(p, d) { seen.push(p); }, _getData() { return {}; }, middleware: { middleware: { put: [], afterPut: [] }, eventListeners: {} } };
  let merged = 0;
  const crtStub = new GpuCRTc(fake);                     // the real eligibility rules; the device call is stubbed
  crtStub.mergeEntries = function (run) { merged += run.length; return { host: [], nApplied: 0 }; };
  const sync = installBatchSync(fake, crtStub, {});
  const entries = [0, 1, 2].map((i) => ({ path: "m/k" + i, data: { v: i }, vectorClock: { w: 10 + i } }));
  sync.processSyncEntries(entries);
  assert.strictEqual(merged, 3); assert.strictEqual(seen.length, 0);
  fake.middleware.middleware.put.push((p, d) => d);
  sync.processSyncEntries(entries);
  assert.strictEqual(merged, 3); assert.deepStrictEqual(seen, ["m/k0", "m/k1", "m/k2"]);
  fake.middleware.middleware.put.length = 0; fake.middleware.eventListeners.write = [() => {}];
  sync.processSyncEntries(entries);
  assert.strictEqual(merged, 3); assert.strictEqual(seen.length, 6);
  // the index hook of a batch that never passes through setData
  const touched = [], updated = [];
  const c1 = new GpuCRTc({ id: "w", meta: {}, query: { indexedPaths: new Set(["m"]), _touch(p) { touched.push(p); } } });
  c1._notifyIndexHook(entries, [1]);
  assert.deepStrictEqual(touched, ["m/k0", "m/k2"]);
  const c2 = new GpuCRTc({ id: "w", meta: {}, query: { indexedPaths: new Set(["m"]), _updateIndices(p, d) { updated.push([p, d.v]); } } });
  c2._notifyIndexHook(entries, []);
  assert.deepStrictEqual(updated, [["m/k0", 0], ["m/k1", 1], ["m/k2", 2]]);
  checks += 8;
}

/* g13 (host side): integer ENTRIES resolved against the sender's clock through the host twin's processUpdate — the reference's public resolver on the
 * same entry list (identical clocks are decided by value; tests/golden/g13_entries_integer_ties.json). The device path of the same fixture: device_parity.js */
{
  const g = load("g13_entries_integer_ties.json");
  const crt = newCrt();
  const state = new Map();
  g.chunks.forEach((chunk, ci) => {
    chunk.forEach((e, j) => {
      const cur = state.get(e.path);
      const r = crt.processUpdate(e.path, JSON.parse(JSON.stringify(e.data)), e.vectorClock, cur ? cur.value : undefined, cur ? cur.clock : undefined);
      assert.strictEqual(flagsOf(r.decision), g.decisions[ci][j], "g13 flags of entry " + j + " of chunk " + (ci + 1));
      if (r.decision.incoming || !cur || r.decision.concurrent) state.set(e.path, { value: r.value, clock: r.vectorClock });
      checks++;
    });
    const snap = {};
    for (const [p, v] of state) snap[p] = { value: v.value, clock: v.clock };
    assert.deepStrictEqual(JSON.parse(JSON.stringify(snap)), g.after[ci], "g13 (host) state after chunk " + (ci + 1));
  });
}

console.log("host_semantics ok:", checks, "checks");
