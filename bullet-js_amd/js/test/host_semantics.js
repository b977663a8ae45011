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
  console.log("reference facade present: replayed through the real Bullet with GpuCRT/GpuQuery plugged in");
}

console.log("host_semantics ok:", checks, "checks");
