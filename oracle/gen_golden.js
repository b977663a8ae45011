#!/usr/bin/env node
/*
 * gen_golden.js — TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Runs the REAL reference modules (KORandi/bullet-js, mounted read-only at
 * $BULLET_REF or /root/reference) under Node on seeded synthetic inputs and
 * writes input-spec / expected-output pairs into tests/golden/*.json.
 * Only *data* is written: no reference source text is copied anywhere.
 *
 * What is exercised (reference file:line):
 *   - BulletCRT.processUpdate / resolve            src/bullet-crt.js:164-318   (oracle level L0)
 *   - Bullet.setData -> crt.handleUpdate -> apply  src/bullet.js:139-220, src/bullet-crt.js:329-385 (L1)
 *   - BulletQuery.index/equals/range/count/filter  src/bullet-query.js:30-313
 *
 * The stream generator below is mirrored bit-for-bit by oracle/streams.py; a
 * mismatch between the two shows up as a digest mismatch in tests/test_oracle_golden.py.
 *
 * Usage: node oracle/gen_golden.js [outdir]      (default: tests/golden)
 */
"use strict";
const fs = require("fs");
const path = require("path");

const REF = process.env.BULLET_REF || "/root/reference";
const OUT = process.argv[2] || path.join(__dirname, "..", "tests", "golden");
let BulletCRT = null;   /* loaded lazily: importing this file for genStream() must not need the reference */

/* ------------------------------------------------------------------ PRNG / hashes */
const M64 = (1n << 64n) - 1n;
function xorshift32(seed) {
  let s = seed >>> 0;
  if (s === 0) s = 0x9e3779b9;
  return function next() {
    s ^= s << 13; s >>>= 0;
    s ^= s >>> 17;
    s ^= s << 5; s >>>= 0;
    return s;
  };
}
function splitmix64(x) {
  let z = (BigInt.asUintN(64, x) + 0x9e3779b97f4a7c15n) & M64;
  z = ((z ^ (z >> 30n)) * 0xbf58476d1ce4e5b9n) & M64;
  z = ((z ^ (z >> 27n)) * 0x94d049bb133111ebn) & M64;
  return z ^ (z >> 31n);
}
function fnv1a32(str) {
  let h = 0x811c9dc5;
  for (let i = 0; i < str.length; i++) {
    h ^= str.charCodeAt(i) & 0xff;
    h = Math.imul(h, 0x01000193) >>> 0;
  }
  return h >>> 0;
}
function fieldHash(fi) { return fnv1a32("f" + fi); }
function rowId(row, F) { return splitmix64(BigInt(Math.floor(row / F)) + 1n); }
function rowField(row, F) { return fieldHash(row % F); }

/* order-independent digest of a (id, field, ts, val) row set */
function rowDigest(id, field, ts, val) {
  let h = splitmix64(BigInt.asUintN(64, BigInt(val)));
  h = splitmix64(h ^ BigInt.asUintN(64, BigInt(ts)));
  h = splitmix64(h ^ BigInt(field));
  h = splitmix64(h ^ id);
  return h;
}

/* ------------------------------------------------------------------ stream spec -> rows */
const PERM_PRIME = 1000003;
function genStream(spec) {
  const rng = xorshift32(spec.seed);
  const F = spec.F || 1;
  const resident = [];
  for (let r = 0; r < spec.R; r++) {
    const ts = spec.T0 + (rng() % spec.DT);
    const val = (rng() % spec.VR) - spec.VOFF;
    resident.push({ row: r, ts, val });
  }
  const deltas = [];
  const insSpace = spec.ins_space || Math.max(1, Math.floor(spec.R / 10));
  for (let j = 0; j < spec.D; j++) {
    const u = rng() % 100;
    let row;
    if (u < spec.insert_pct) {
      row = spec.unique ? spec.R + j : spec.R + (rng() % insSpace);
    } else if (u < spec.insert_pct + spec.hot_pct) {
      row = rng() % spec.H;
    } else if (spec.unique) {
      row = (j * PERM_PRIME + 7) % spec.R;
    } else {
      row = rng() % spec.R;
    }
    const ts = spec.T0 + (rng() % (2 * spec.DT));
    const val = (rng() % spec.VR) - spec.VOFF;
    deltas.push({ row, ts, val });
  }
  return { resident, deltas, F };
}

/* ------------------------------------------------------------------ L0 harness around the real BulletCRT */
const WID = "w";
function newCrt() { if (!BulletCRT) BulletCRT = require(path.join(REF, "src", "bullet-crt.js")); return new BulletCRT({ id: WID, meta: {}, _getData() { return undefined; } }); }
function flagsOf(d) {
  return (d.incoming ? 1 : 0) | (d.current ? 2 : 0) | (d.historical ? 4 : 0) | (d.concurrent ? 8 : 0);
}
function clockTs(clock) {
  const ks = Object.keys(clock);
  if (ks.length !== 1 || ks[0] !== WID) throw new Error("off-contract clock " + JSON.stringify(clock));
  return clock[WID];
}
/* state: Map key -> {value, clock}; returns decision */
function applyDelta(crt, state, key, ts, val) {
  const cur = state.get(key);
  const r = crt.processUpdate(key, val, { [WID]: ts }, cur ? cur.value : undefined, cur ? cur.clock : undefined);
  const d = r.decision;
  /* store rule of the caller: src/bullet-crt.js:383 + src/bullet.js:144-148 */
  if (d.incoming || !cur || d.concurrent) state.set(key, { value: r.value, clock: r.vectorClock });
  return d;
}

function runStream(spec) {
  const { resident, deltas, F } = genStream(spec);
  const crt = newCrt();
  const state = new Map();
  const keyOf = (row) => rowId(row, F).toString(16) + ":" + rowField(row, F);
  for (const r of resident) state.set(keyOf(r.row), { value: r.val, clock: { [WID]: r.ts }, row: r.row });
  const flags = Buffer.alloc(deltas.length);
  const lastApplied = new Map();
  let nApplied = 0, nHist = 0;
  deltas.forEach((d, j) => {
    const key = keyOf(d.row);
    const dec = applyDelta(crt, state, key, d.ts, d.val);
    flags[j] = flagsOf(dec);
    if (dec.concurrent) throw new Error("concurrent branch hit: off-contract");
    if (dec.incoming) { lastApplied.set(key, j); nApplied++; state.get(key).row = d.row; }
    if (dec.historical) nHist++;
  });
  const winners = Array.from(lastApplied.values()).sort((a, b) => a - b);
  let digest = 0n;
  const finalRows = [];
  for (const [, st] of state) {
    const ts = clockTs(st.clock);
    digest = (digest + rowDigest(rowId(st.row, F), rowField(st.row, F), ts, st.value)) & M64;
    finalRows.push([st.row, ts, st.value]);
  }
  finalRows.sort((a, b) => a[0] - b[0]);
  const out = {
    kind: "stream",
    source: "reference BulletCRT.processUpdate (src/bullet-crt.js:304-318), single-component clocks {w:ts}",
    spec,
    n_rows_final: state.size,
    n_incoming: nApplied,
    n_historical: nHist,
    flags_b64: flags.toString("base64"),
    winners,
    digest: digest.toString(16),
  };
  if (spec.store_final) out.final_rows = finalRows; /* [row ordinal, ts, val] */
  else out.final_sample = finalRows.filter((_, i) => i % Math.ceil(finalRows.length / 200) === 0);
  return out;
}

/* ------------------------------------------------------------------ G1/G3: decision table + ts edge cases */
function genDecisionTable() {
  const TS = [0, 1, 2, 3, 2147483647, 2147483648, 2147483649, Number.MAX_SAFE_INTEGER];
  const VALS = [-5, 0, 7];
  const VEDGE = [-(2 ** 53 - 1), 2 ** 53 - 1, -1, 1];
  const cases = [];
  const one = (cur, a, v) => {
    const crt = newCrt();
    const r = crt.processUpdate("k", v, { [WID]: a }, cur ? cur.val : undefined, cur ? { [WID]: cur.ts } : undefined);
    const d = r.decision;
    cases.push({
      cur: cur ? [cur.ts, cur.val] : null, inc: [a, v], flags: flagsOf(d),
      out: [clockTs(r.vectorClock), r.value], reason: d.reason,
    });
  };
  for (const a of TS) for (const v of VALS) one(null, a, v);
  for (const b of TS) for (const c of VALS) for (const a of TS) for (const v of VALS) one({ ts: b, val: c }, a, v);
  for (const c of VEDGE) for (const v of VEDGE) { one({ ts: 5, val: c }, 5, v); one({ ts: 5, val: c }, 6, v); one({ ts: 6, val: c }, 5, v); }
  return { kind: "decision_table", source: "reference BulletCRT.processUpdate, fresh resolver per case", cases };
}

/* G3: short scripted sequences on one key over tiny domains: exercises the insert quirk ({w:2}) */
function genSequences() {
  const rng = xorshift32(424242);
  const seqs = [];
  for (let s = 0; s < 300; s++) {
    const crt = newCrt();
    const state = new Map();
    const startResident = (rng() % 3) === 0;
    let start = null;
    if (startResident) { start = [rng() % 5, (rng() % 3) - 1]; state.set("k", { value: start[1], clock: { [WID]: start[0] } }); }
    const n = 2 + (rng() % 7);
    const deltas = [], flags = [];
    let winner = -1;
    for (let j = 0; j < n; j++) {
      const ts = rng() % 5, val = (rng() % 3) - 1;
      const d = applyDelta(crt, state, "k", ts, val);
      deltas.push([ts, val]); flags.push(flagsOf(d));
      if (d.incoming) winner = j;
    }
    const st = state.get("k");
    seqs.push({ start, deltas, flags, final: [clockTs(st.clock), st.value], winner });
  }
  return { kind: "sequences", source: "reference BulletCRT.processUpdate, one key, ts in 0..4, val in -1..1", seqs };
}

/* ------------------------------------------------------------------ N4: multi-writer ("dense") vector clocks, integer values
 * Contract of the device path: every incoming clock lists the same K writers in the same key order (WRITERS below, the local
 * id last), components are small non-negative integers. Then JSON-equality of two clocks is component equality, except
 * against the one-key clock {local: 2} that a first write stores (src/bullet-crt.js:172-185) — fixtures cover that too. */
const WRITERS = ["a", "b", "w"];
function denseClock(c) { const o = {}; WRITERS.forEach((k, i) => { o[k] = c[i]; }); return o; }
function compsOf(clock) { return WRITERS.map((k) => clock[k] || 0); }
function runVcStream(spec) {
  const rng = xorshift32(spec.seed);
  const crt = newCrt();
  const state = new Map();
  const K = WRITERS.length;
  const rc = () => { const c = []; for (let k = 0; k < K; k++) c.push(rng() % spec.CMAX); return c; };
  const resident = [];
  for (let r = 0; r < spec.R; r++) {
    const c = rc(); if (c.every((x) => x === 0)) c[K - 1] = 1;
    const val = (rng() % spec.VR) - spec.VOFF;
    resident.push({ row: r, clock: c, val });
    state.set("k" + r, { value: val, clock: denseClock(c), row: r });
  }
  const deltas = [];
  const flags = Buffer.alloc(spec.D);
  const last = new Map();
  for (let j = 0; j < spec.D; j++) {
    const u = rng() % 100;
    let row;
    if (u < spec.insert_pct) row = spec.R + (rng() % spec.ins_space);
    else if (u < spec.insert_pct + spec.hot_pct) row = rng() % spec.H;
    else row = rng() % spec.R;
    const c = rc();
    const val = (rng() % spec.VR) - spec.VOFF;
    deltas.push({ row, clock: c, val });
    const key = "k" + row;
    const cur = state.get(key);
    const r = crt.processUpdate(key, val, denseClock(c), cur ? cur.value : undefined, cur ? cur.clock : undefined);
    const d = r.decision;
    flags[j] = flagsOf(d);
    if (d.incoming || !cur || d.concurrent) { state.set(key, { value: r.value, clock: r.vectorClock, row }); last.set(key, j); }
  }
  const finalRows = [];
  for (const [, st] of state) finalRows.push([st.row, compsOf(st.clock), st.value, Object.keys(st.clock).length]);
  finalRows.sort((x, y) => x[0] - y[0]);
  return { kind: "vc_stream", source: "reference BulletCRT.processUpdate with dense 3-writer clocks {a,b,w}, bullet.id = 'w'", writers: WRITERS, spec,
    resident: resident.map((r) => [r.row, r.clock, r.val]), deltas: deltas.map((d) => [d.row, d.clock, d.val]),
    flags_b64: flags.toString("base64"), updated: Array.from(last.values()).sort((x, y) => x - y), final_rows: finalRows };
}

/* N4 with arbitrary KEY SETS: every clock is an object over a random ordered subset of the writers (any order, any number of them, {} included).
 * What the fixture pins beyond the dense streams: a missing key counts as 0 in the dominance test (src/bullet-crt.js:76-79), clocks with equal
 * counters but different key sets / key orders are NOT identical (JSON.stringify, :200-203) and fall through to the concurrent branch, and the
 * stored clock's key order after a merge is the incoming clock's keys followed by the stored clock's other keys (:103-114).
 * Rows: [row, keys (writer indices in the object's key order), counters (K, 0 for a writer the clock does not name), value]. */
function runVcKeysetStream(spec) {
  const rng = xorshift32(spec.seed);
  const crt = newCrt();
  const state = new Map();
  const K = WRITERS.length;
  const randClock = () => {
    const order = WRITERS.map((_, i) => i);
    for (let i = K - 1; i > 0; i--) { const j = rng() % (i + 1); const t = order[i]; order[i] = order[j]; order[j] = t; }
    const cnt = rng() % 100 < spec.full_pct ? K : rng() % (K + 1);
    const keys = order.slice(0, cnt), comps = new Array(K).fill(0), obj = {};
    for (const k of keys) { comps[k] = rng() % spec.CMAX; obj[WRITERS[k]] = comps[k]; }
    return { keys, comps, obj };
  };
  const keysOf = (clock) => Object.keys(clock).map((w) => WRITERS.indexOf(w));
  const resident = [];
  for (let r = 0; r < spec.R; r++) {
    const c = randClock();
    const val = (rng() % spec.VR) - spec.VOFF;
    resident.push([r, c.keys, c.comps, val]);
    state.set("k" + r, { value: val, clock: c.obj, row: r });
  }
  const deltas = [];
  const flags = Buffer.alloc(spec.D);
  const last = new Map();
  for (let j = 0; j < spec.D; j++) {
    const u = rng() % 100;
    let row;
    if (u < spec.insert_pct) row = spec.R + (rng() % spec.ins_space);
    else if (u < spec.insert_pct + spec.hot_pct) row = rng() % spec.H;
    else row = rng() % Math.max(spec.R, 1);
    const c = randClock();
    const val = (rng() % spec.VR) - spec.VOFF;
    deltas.push([row, c.keys, c.comps, val]);
    const key = "k" + row;
    const cur = state.get(key);
    const r = crt.processUpdate(key, val, c.obj, cur ? cur.value : undefined, cur ? cur.clock : undefined);
    const d = r.decision;
    flags[j] = flagsOf(d);
    if (d.incoming || !cur || d.concurrent) { state.set(key, { value: r.value, clock: r.vectorClock, row }); last.set(key, j); }
  }
  const finalRows = [];
  for (const [, st] of state) finalRows.push([st.row, keysOf(st.clock), compsOf(st.clock), st.value]);
  finalRows.sort((x, y) => x[0] - y[0]);
  return { kind: "vc_keyset_stream", source: "reference BulletCRT.processUpdate with clocks over ordered subsets of the writers {a,b,w}, bullet.id = 'w'", writers: WRITERS, spec,
    resident, deltas, flags_b64: flags.toString("base64"), updated: Array.from(last.values()).sort((x, y) => x - y), final_rows: finalRows };
}

/* ------------------------------------------------------------------ L1 / query fixtures through the real Bullet facade */
function quiet(fn) { const l = console.log; console.log = () => {}; try { return fn(); } finally { console.log = l; } }
function newBullet(extra) {
  const Bullet = require(path.join(REF, "src", "bullet.js"));
  return quiet(() => {
    const b = new Bullet(Object.assign({ disableNetwork: true, storage: false, server: false, enableMiddleware: false,
      enableValidation: false, enableSerializer: false }, extra || {}));
    b.id = WID;
    return b;
  });
}
function genL1() {
  /* scripted node-level ops; expected store/meta after each op. enableIndexing:false so setData returns the value */
  const b = newBullet({ enableIndexing: false });
  const ops = [
    { path: "users/bob", data: { name: "Bob", age: 30, __fromNetwork: true, __vectorClock: { w: 5 } } },
    { path: "users/bob", data: { name: "Bobby", __fromNetwork: true, __vectorClock: { w: 7 } } },          /* dominate -> whole-object replace */
    { path: "users/bob", data: { name: "Old", age: 1, __fromNetwork: true, __vectorClock: { w: 6 } } },     /* historical */
    { path: "users/bob", data: { name: "Aaa", __fromNetwork: true, __vectorClock: { w: 7 } } },             /* tie: objects compare +1 -> incoming */
    { path: "users/bob", data: { name: "Aaa", __fromNetwork: true, __vectorClock: { w: 7 } } },             /* tie again, still incoming (objects never ===) */
    { path: "users/amy", data: { city: "X", __fromNetwork: true, __vectorClock: { w: 3 } } },               /* insert quirk: stored clock {w:2} */
    { path: "users/amy", data: { city: "Y", __fromNetwork: true, __vectorClock: { w: 1 } } },               /* 1 < 2 -> historical */
    { path: "users/amy", data: { city: "Z", __fromNetwork: true, __vectorClock: { w: 2 } } },               /* 2 == 2 -> tie -> incoming */
    { path: "users/cat", data: { a: 1, n: { x: 1 }, __fromNetwork: true, __vectorClock: { w: 4, p: 1 } } },
    { path: "users/cat", data: { b: 2, n: { y: 2 }, __fromNetwork: true, __vectorClock: { w: 3, q: 9 } } }, /* after insert clock is {w:2}: concurrent -> deep merge */
    { path: "users/cat", data: { b: 5, c: 3, n: { y: 1, z: 3 }, __fromNetwork: true, __vectorClock: { w: 1, p: 50 } } }, /* truly concurrent vs {w:3,q:9}: deep merge, per-leaf compare */
    { path: "cfg/theme", data: "dark" },                                                                     /* local primitive writes */
    { path: "cfg/theme", data: "light" },
    { path: "cfg/theme", data: "light" },
    { path: "cfg/n", data: 41 },
    { path: "cfg/n", data: 40 },
    /* nulls and falsy leaves: delete-vs-update is order dependent (docs/conflict-resolution.md:512-523), and reading a falsy
       leaf through _getData replaces it with {} (src/bullet.js:122-124) */
    { path: "tmp/x", data: { a: 1, __fromNetwork: true, __vectorClock: { w: 5 } } },
    { path: "tmp/x", data: null },                                                                          /* local delete: null vs object */
    { path: "tmp/x", data: { a: 2, __fromNetwork: true, __vectorClock: { w: 5 } } },                        /* object vs the autovivified {} */
    { path: "tmp/z", data: 0 },                                                                              /* falsy leaf */
    { path: "tmp/z", data: 7 },
    { path: "tmp/z", data: "" },
    { path: "tmp/z", data: false },
    { path: "tmp/arr", data: [1, 2, 3] },
    { path: "tmp/arr", data: [1, 2] },
    { path: "tmp/s", data: "b" },
    { path: "tmp/s", data: "a" },
    { path: "tmp/s", data: "c" },
  ];
  const steps = [];
  for (const op of ops) {
    const hu = JSON.parse(JSON.stringify(op.data));
    const ret = quiet(() => b.setData(op.path, op.data, false));
    steps.push({
      path: op.path, data: hu, ret: ret === undefined ? null : JSON.parse(JSON.stringify(ret)),
      store: JSON.parse(JSON.stringify(b.store)),
      clock: b.meta[op.path] ? JSON.parse(JSON.stringify(b.meta[op.path].vectorClock)) : null,
      source: b.meta[op.path] ? b.meta[op.path].source : null,
      log_len: b.log.length,
    });
  }
  return { kind: "l1_ops", source: "reference Bullet.setData -> BulletCRT.handleUpdate (src/bullet.js:139-220, src/bullet-crt.js:329-385), bullet.id='w'", steps };
}

function childKeys(nodes) { return nodes.map((n) => n.path.split("/").pop()); }

function genQueryExample() {
  /* known-answer dataset (values only) of examples/bullet-query-example.js:17-47 */
  const users = {
    user1: { name: "Alice Johnson", age: 28, active: true, role: "admin" }, user2: { name: "Bob Smith", age: 35, active: true, role: "user" },
    user3: { name: "Carol Davis", age: 42, active: false, role: "user" }, user4: { name: "Dave Wilson", age: 23, active: true, role: "editor" },
    user5: { name: "Eve Brown", age: 31, active: true, role: "user" }, user6: { name: "Frank Miller", age: 47, active: false, role: "admin" },
    user7: { name: "Grace Lee", age: 29, active: true, role: "editor" }, user8: { name: "Harry Taylor", age: 39, active: true, role: "user" },
    user9: { name: "Irene Clark", age: 26, active: false, role: "user" }, user10: { name: "Jack Roberts", age: 33, active: true, role: "admin" },
  };
  const products = {
    prod1: { name: "Laptop", price: 1200, stock: 15, category: "electronics" }, prod2: { name: "Smartphone", price: 800, stock: 25, category: "electronics" },
    prod3: { name: "Headphones", price: 150, stock: 50, category: "accessories" }, prod4: { name: "Mouse", price: 30, stock: 100, category: "accessories" },
    prod5: { name: "Keyboard", price: 80, stock: 40, category: "accessories" }, prod6: { name: "Monitor", price: 300, stock: 20, category: "electronics" },
    prod7: { name: "Desk Chair", price: 250, stock: 10, category: "furniture" }, prod8: { name: "Desk", price: 400, stock: 5, category: "furniture" },
    prod9: { name: "Printer", price: 200, stock: 8, category: "electronics" }, prod10: { name: "Camera", price: 600, stock: 12, category: "electronics" },
  };
  const b = newBullet({ enableIndexing: true });
  quiet(() => {
    for (const [k, v] of Object.entries(users)) b.get("users/" + k).put(v);
    for (const [k, v] of Object.entries(products)) b.get("products/" + k).put(v);
  });
  const q = [];
  const rec = (name, args, nodes) => q.push({ op: name, args, keys: childKeys(nodes) });
  quiet(() => {
    rec("range", ["users", "age", 30, 40], b.range("users", "age", 30, 40));
    rec("range", ["users", "age", 0, 1000], b.range("users", "age", 0, 1000));
    rec("range", ["users", "age", 40, 30], b.range("users", "age", 40, 30));
    rec("equals", ["users", "age", 42], b.equals("users", "age", 42));
    rec("equals", ["users", "age", "42"], b.equals("users", "age", "42"));
    rec("equals", ["users", "age", 99], b.equals("users", "age", 99));
    rec("range", ["products", "price", 100, 500], b.range("products", "price", 100, 500));
    rec("range", ["products", "price", 600, Infinity], b.range("products", "price", 600, "Infinity"));
    rec("range", ["products", "stock", 10, 25], b.range("products", "stock", 10, 25));
    rec("equals", ["products", "stock", 50], b.equals("products", "stock", 50));
    rec("filter", ["products", "price>=200&&stock<=12"], b.filter("products", (p) => p.price >= 200 && p.stock <= 12));
    q.push({ op: "count", args: ["users", "age", 35], n: b.query.count("users", "age", 35) });
    q.push({ op: "count", args: ["products", "price", 30], n: b.query.count("products", "price", 30) });
    q.push({ op: "range_undefined_max", args: ["users", "age", 30], keys: childKeys(b.range("users", "age", 30, undefined)) });
  });
  /* the Infinity query above used the string form to survive JSON; redo with a real Infinity */
  q[7] = { op: "range", args: ["products", "price", 600, "Infinity"], keys: quiet(() => childKeys(b.range("products", "price", 600, Infinity))) };
  return { kind: "query_example", source: "reference BulletQuery via Bullet facade (src/bullet-query.js:186-313), dataset values of examples/bullet-query-example.js:17-47",
    users, products, queries: q, indices: Object.keys(b.query.indices) };
}

function genQuerySeeded(N, seed, full) {
  const rng = xorshift32(seed);
  const b = newBullet({ enableIndexing: true });
  const ages = new Array(N), scores = new Array(N);
  quiet(() => {
    for (let i = 0; i < N; i++) {
      ages[i] = rng() % 100; scores[i] = (rng() % 200001) - 100000;
      /* network-tagged object write with scalar clock, as sync entries arrive (src/bullet-network-sync.js:551-569) */
      b.setData("n/k" + i, { age: ages[i], score: scores[i], __fromNetwork: true, __vectorClock: { w: 10 + (i % 7) } }, false);
    }
  });
  const ord = (nodes) => nodes.map((n) => parseInt(n.path.split("/").pop().slice(1), 10));
  const sum = (a) => a.reduce((x, y) => x + y, 0);
  const queries = [];
  const rec = (op, field, args, nodes) => {
    const o = ord(nodes);
    const e = { op, field, args, count: o.length, ordinal_sum: sum(o) };
    if (full) e.ordinals = o; /* reference order */
    queries.push(e);
  };
  quiet(() => {
    rec("equals", "age", [42], b.equals("n", "age", 42));
    rec("equals", "age", [0], b.equals("n", "age", 0));
    rec("equals", "age", [100], b.equals("n", "age", 100));
    rec("range", "age", [20, 30], b.range("n", "age", 20, 30));
    rec("range", "age", [0, 99], b.range("n", "age", 0, 99));
    rec("range", "age", [50, 49], b.range("n", "age", 50, 49));
    rec("range", "age", [99, 1e9], b.range("n", "age", 99, 1e9));
    rec("range", "score", [-100, 100], b.range("n", "score", -100, 100));
    rec("range", "score", [-100000, -99000], b.range("n", "score", -100000, -99000));
    rec("range", "score", [0, 100000], b.range("n", "score", 0, 100000));
    rec("equals", "score", [scores[3]], b.equals("n", "score", scores[3]));
    queries.push({ op: "count", field: "age", args: [7], count: b.query.count("n", "age", 7) });
    rec("filter_and", "age,score", [[25, 35], [0, 50000]], b.filter("n", (v) => v.age >= 25 && v.age <= 35 && v.score >= 0 && v.score <= 50000));
  });
  return { kind: "query_seeded", source: "reference BulletQuery (fresh index: queries issued after all writes)", N, seed,
    gen: "for i<N: age=rng()%100; score=rng()%200001-100000 (xorshift32, interleaved per node)", queries };
}


/* ------------------------------------------------------------------ g8: a sync chunk through the reference's own loop (N2)
 * BulletNetworkSync._processSyncEntries (src/bullet-network-sync.js:551-569) run on a real Bullet (network disabled: the method
 * only needs this.bullet). The chunk mixes the two kinds of entry the loop treats differently — objects are tagged with
 * __fromNetwork/__vectorClock and resolved against the sender's clock, primitives are passed UNTAGGED and therefore handled as
 * local writes (clock ignored, always accepted) — plus deletions and off-contract values. Expected: store and per-path clock/source. */
function genSyncChunk() {
  const Sync = require(path.join(REF, "src", "bullet-network-sync.js"));
  const b = newBullet({ enableIndexing: false });
  const chunks = [
    [ /* chunk 1: first sight of everything */
      { path: "acct/a", data: { bal: 10, seq: 1 }, vectorClock: { w: 100 } },
      { path: "acct/b", data: { bal: 20, seq: 1 }, vectorClock: { w: 100 } },
      { path: "acct/c", data: { bal: 30, seq: 1 }, vectorClock: { w: 100 } },
      { path: "cfg/limit", data: 5, vectorClock: { w: 100 } },                 /* primitive: untagged -> local write */
      { path: "cfg/name", data: "alpha", vectorClock: { w: 100 } },
      { path: "users/u1", data: { name: "Ann", age: 30 }, vectorClock: { w: 100 } },   /* string field: host path in the batch adapter */
    ],
    [ /* chunk 2: the first writes stored clock {w:2}; 150 dominates, 1 is historical */
      { path: "acct/a", data: { bal: 11, seq: 2 }, vectorClock: { w: 150 } },
      { path: "acct/b", data: { bal: 99, seq: 9 }, vectorClock: { w: 1 } },
      { path: "cfg/limit", data: 4, vectorClock: { w: 1 } },                   /* stale clock, still applied: the clock of a primitive is ignored */
      { path: "cfg/name", data: "beta", vectorClock: { w: 1 } },
      { path: "users/u1", data: { name: "Bea", age: 31 }, vectorClock: { w: 150 } },
      { path: "acct/c", deleted: true, vectorClock: { w: 150 } },              /* deletion: setData(path, null) */
      { path: "acct/d", data: { bal: 40, seq: 1 }, vectorClock: { w: 120, q: 3 } },    /* two writers: host path */
    ],
    [ /* chunk 3 */
      { path: "acct/a", data: { bal: 12, seq: 3 }, vectorClock: { w: 149 } },  /* older than 150: historical */
      { path: "acct/b", data: { bal: 21, seq: 2 }, vectorClock: { w: 200 } },
      { path: "acct/e", data: { bal: 50, seq: 1 }, vectorClock: { w: 7 } },
      { path: "acct/e", data: { bal: 51, seq: 2 }, vectorClock: { w: 8 } },    /* same node twice in one chunk */
      { path: "cfg/limit", data: 6, vectorClock: { w: 300 } },
    ],
  ];
  const fake = { bullet: b };
  quiet(() => { for (const c of chunks) Sync.prototype._processSyncEntries.call(fake, c, "peer-1"); });
  const meta = {};
  for (const k of Object.keys(b.meta)) meta[k] = { vectorClock: b.meta[k].vectorClock, source: b.meta[k].source };
  return { kind: "sync_chunk", source: "reference BulletNetworkSync._processSyncEntries on a real Bullet (id 'w', network disabled)", id: "w",
    chunks, store: JSON.parse(JSON.stringify(b.store)), meta };
}

/* ------------------------------------------------------------------ g9: NODE-level semantics of synced objects (N2/N4)
 * The same loop as g8, with exactly the entries a per-FIELD last-writer-wins table gets wrong (VERDICT r2 weak #1):
 *   - a dominating object REPLACES the node: fields it no longer carries disappear (src/bullet-crt.js:236-248);
 *   - equal clocks + object values: compare() is +1 for objects, the INCOMING object wins whatever its values (:11-15, :200-233),
 *     also against the {w:2} clock a first write leaves behind (:172-185);
 *   - a field that appears later belongs to the node's clock, not to a clock of its own;
 *   - deletions (setData(path, null), src/bullet-network-sync.js:553-555) and what an index / a query sees afterwards
 *     (_addToIndex skips null and non-object children, src/bullet-query.js:53-94);
 *   - a node first written through the host path (string field), later by an all-integer object.
 * After every chunk the reference's store and per-path clock/source are recorded; after the last one, queries on a fresh index. */
function genSyncNodeSemantics() {
  const Sync = require(path.join(REF, "src", "bullet-network-sync.js"));
  const b = newBullet({ enableIndexing: true });
  const chunks = [
    [ /* chunk 1: first sight (stored clock {w:2}); acct/t twice: the second entry TIES with that stored clock and wins with the smaller value */
      { path: "acct/a", data: { bal: 10, seq: 1 }, vectorClock: { w: 100 } },
      { path: "acct/b", data: { bal: 20, seq: 1, lim: 5 }, vectorClock: { w: 100 } },
      { path: "acct/c", data: { bal: 30 }, vectorClock: { w: 100 } },
      { path: "acct/d", data: { bal: 40, seq: 1 }, vectorClock: { w: 100 } },
      { path: "acct/t", data: { bal: 9 }, vectorClock: { w: 50 } },
      { path: "acct/t", data: { bal: 1 }, vectorClock: { w: 2 } },
      { path: "acct/g", data: { bal: 60, seq: 1 }, vectorClock: { w: 100 } },
      { path: "users/u1", data: { name: "Ann", age: 30 }, vectorClock: { w: 100 } },
      { path: "cfg/limit", data: 5, vectorClock: { w: 100 } },
    ],
    [ /* chunk 2: shrinking and growing field sets under dominating clocks, a tie on {w:2}, a deletion */
      { path: "acct/a", data: { bal: 13 }, vectorClock: { w: 151 } },
      { path: "acct/b", data: { bal: 21, seq: 2 }, vectorClock: { w: 150 } },
      { path: "acct/c", data: { bal: 31, extra: 7 }, vectorClock: { w: 150 } },
      { path: "acct/d", deleted: true, vectorClock: { w: 150 } },
      { path: "acct/t", data: { bal: 0, seq: 4 }, vectorClock: { w: 2 } },
      { path: "acct/g", data: { bal: 61 }, vectorClock: { w: 1 } },
      { path: "users/u1", data: { age: 31 }, vectorClock: { w: 150 } },
    ],
    [ /* chunk 3: equal clocks after a dominating write, a historical write, the late field under a clock below the node's, re-creation after a delete,
         one node three times (the last two tie: the later one wins) */
      { path: "acct/a", data: { bal: 1, seq: 9 }, vectorClock: { w: 151 } },
      { path: "acct/a", data: { bal: 99 }, vectorClock: { w: 150 } },
      { path: "acct/c", data: { bal: 32 }, vectorClock: { w: 3 } },
      { path: "acct/d", data: { bal: 41, seq: 2 }, vectorClock: { w: 200 } },
      { path: "acct/e", data: { bal: 50 }, vectorClock: { w: 7 } },
      { path: "acct/e", data: { bal: 51, seq: 2 }, vectorClock: { w: 8 } },
      { path: "acct/e", data: { bal: 52 }, vectorClock: { w: 8 } },
      { path: "cfg/limit", data: 6, vectorClock: { w: 300 } },
    ],
    [ /* chunk 4: delete, then a write at the clock the delete left behind (a tie: incoming wins), a second delete that stays */
      { path: "acct/b", deleted: true, vectorClock: { w: 160 } },
      { path: "acct/g", deleted: true, vectorClock: { w: 160 } },
      { path: "acct/g", data: { bal: 62, seq: 3 }, vectorClock: { w: 3 } },
      { path: "acct/h", data: { bal: 70 }, vectorClock: { w: 9 } },
      { path: "acct/h", deleted: true, vectorClock: { w: 10 } },
    ],
  ];
  const fake = { bullet: b };
  const after = [];
  quiet(() => {
    for (const c of chunks) {
      Sync.prototype._processSyncEntries.call(fake, c, "peer-1");
      const meta = {};
      for (const k of Object.keys(b.meta)) meta[k] = { vectorClock: JSON.parse(JSON.stringify(b.meta[k].vectorClock)), source: b.meta[k].source };
      after.push({ store: JSON.parse(JSON.stringify(b.store)), meta });
    }
  });
  /* queries on an index that is first built now (the fresh state: SURVEY §8(a) "Scan parity target") */
  const paths = (nodes) => nodes.map((n) => n.path);
  const queries = [];
  quiet(() => {
    queries.push({ op: "range", path: "acct", field: "bal", args: [0, 100], paths: paths(b.range("acct", "bal", 0, 100)) });
    queries.push({ op: "equals", path: "acct", field: "bal", args: [41], paths: paths(b.equals("acct", "bal", 41)) });
    queries.push({ op: "equals", path: "acct", field: "bal", args: [21], paths: paths(b.equals("acct", "bal", 21)) });
    queries.push({ op: "range", path: "acct", field: "seq", args: [0, 100], paths: paths(b.range("acct", "seq", 0, 100)) });
    queries.push({ op: "range", path: "acct", field: "lim", args: [0, 100], paths: paths(b.range("acct", "lim", 0, 100)) });
    queries.push({ op: "range", path: "acct", field: "extra", args: [0, 100], paths: paths(b.range("acct", "extra", 0, 100)) });
    queries.push({ op: "range", path: "users", field: "age", args: [0, 100], paths: paths(b.range("users", "age", 0, 100)) });
    queries.push({ op: "count", path: "acct", field: "bal", args: [52], count: b.query.count("acct", "bal", 52) });
  });
  return { kind: "sync_node_semantics", source: "reference BulletNetworkSync._processSyncEntries + BulletQuery on a real Bullet (id 'w', network disabled)", id: "w",
    chunks, after, queries };
}

/* ------------------------------------------------------------------ g10: synced objects with fields of every JSON type
 * The node-level resolution never looks at an object's fields (two objects under identical clocks: compare() is +1, src/bullet-crt.js:11-15), so the
 * batch path takes ANY plain object, not only all-integer ones. This fixture pins that on the reference's own loop: strings, nested objects, arrays,
 * booleans and null as fields; whole-node replacement dropping such fields; the ONE case where an object entry's fate depends on the stored value — an
 * identical clock against a stored STRING, compared as text ("[object Object]" < "alpha" keeps the string, "ALPHA" loses) —; ties against a stored
 * number, null (a deleted node), {} and the object an array entry turns into (the loop spreads it: src/bullet-network-sync.js:560-563). */
function genSyncMixedValues() {
  const Sync = require(path.join(REF, "src", "bullet-network-sync.js"));
  const b = newBullet({ enableIndexing: true });
  const chunks = [
    [ /* chunk 1: first sight; primitives are local writes */
      { path: "users/u1", data: { name: "Ann", age: 30, tags: ["a", "b"], addr: { city: "X", zip: 10115 }, active: true, nick: null }, vectorClock: { w: 100 } },
      { path: "users/u2", data: { name: "Bob", age: 41 }, vectorClock: { w: 100 } },
      { path: "users/u3", data: { name: "Cy", age: "52" }, vectorClock: { w: 100 } },
      { path: "cfg/name", data: "alpha", vectorClock: { w: 100 } },
      { path: "cfg/up", data: "ALPHA", vectorClock: { w: 100 } },
      { path: "cfg/list", data: [1, 2, 3], vectorClock: { w: 100 } },
      { path: "cfg/num", data: 7, vectorClock: { w: 100 } },
      { path: "m/e", data: {}, vectorClock: { w: 100 } },
    ],
    [ /* chunk 2: replacement, ties on {w:2} against every kind of stored value, a historical write */
      { path: "users/u1", data: { name: "Anna", age: 31 }, vectorClock: { w: 150 } },
      { path: "users/u2", data: { name: "Bobby" }, vectorClock: { w: 2 } },
      { path: "users/u3", data: { name: "Cyd", age: 53 }, vectorClock: { w: 1 } },
      { path: "cfg/name", data: { v: 1 }, vectorClock: { w: 3 } },            /* a primitive's local write left {w:3} (create 1, two increments) */
      { path: "cfg/up", data: { v: 1 }, vectorClock: { w: 3 } },
      { path: "cfg/num", data: { v: 2 }, vectorClock: { w: 3 } },
      { path: "cfg/list", data: { v: 3 }, vectorClock: { w: 2 } },
      { path: "m/e", data: { x: 1, note: "was empty" }, vectorClock: { w: 2 } },
    ],
    [ /* chunk 3: a deletion and a tie on the clock it leaves behind, a string where an integer was, a node created and replaced in one chunk,
         a dominating object over the stored string */
      { path: "users/u1", deleted: true, vectorClock: { w: 160 } },
      { path: "users/u1", data: { name: "Zed", age: 9 }, vectorClock: { w: 151 } },
      { path: "users/u2", data: { name: "B3", age: "x" }, vectorClock: { w: 200 } },
      { path: "users/u4", data: { name: "Dee", age: 61, addr: { city: "Y" } }, vectorClock: { w: 5 } },
      { path: "users/u4", data: { age: 62 }, vectorClock: { w: 5 } },
      { path: "cfg/name", data: { v: 2 }, vectorClock: { w: 4 } },
    ],
    [ /* chunk 4: the former string node is an ordinary node now; an integer where a string was */
      { path: "cfg/name", data: { v: 3, label: "three" }, vectorClock: { w: 4 } },
      { path: "users/u3", data: { age: 54 }, vectorClock: { w: 2 } },
      { path: "users/u5", data: { name: "Eve", age: 70, prefs: { theme: "dark", size: 12 } }, vectorClock: { w: 1 } },
      { path: "cfg/n2", data: 9, vectorClock: { w: 1 } },
      { path: "cfg/n3", data: 9, vectorClock: { w: 1 } },
    ],
    [ /* chunk 5: what a LOSING entry leaves behind. resolve() stores the merged clock of every entry in crt.vectorClocks (src/bullet-crt.js:193-198); for an
         entry that loses this is a fresh object, no longer the one meta[path] holds — so the next local write (a primitive here) increments the copy and
         dominates (cfg/n2: 5 is accepted), where without the losing entry it increments meta's own clock in place, meets "identical clocks" and is decided by
         value (cfg/n3: 5 < 9 is refused) */
      { path: "cfg/n2", data: { x: 1 }, vectorClock: { w: 1 } },
      { path: "cfg/n2", data: 5, vectorClock: { w: 1 } },
      { path: "cfg/n3", data: 5, vectorClock: { w: 1 } },
      { path: "cfg/n3", data: { x: 2 }, vectorClock: { w: 3 } },      /* the refused write has still moved cfg/n3's clock to {w:4}: {w:3} is historical now */
      /* ... and in the store: every resolution starts with _getData(path), which replaces a falsy value on the way by {} (src/bullet.js:115-129) — the null
         of a deleted node becomes {} even though the entry that looked at it loses */
      { path: "users/u6", deleted: true, vectorClock: { w: 1 } },
      { path: "users/u6", data: { name: "late" }, vectorClock: { w: 1 } },
    ],
  ];
  const fake = { bullet: b };
  const after = [];
  quiet(() => {
    for (const c of chunks) {
      Sync.prototype._processSyncEntries.call(fake, JSON.parse(JSON.stringify(c)), "peer-1");
      const meta = {};
      for (const k of Object.keys(b.meta)) meta[k] = { vectorClock: JSON.parse(JSON.stringify(b.meta[k].vectorClock)), source: b.meta[k].source };
      after.push({ store: JSON.parse(JSON.stringify(b.store)), meta });
    }
  });
  const paths = (nodes) => nodes.map((n) => n.path);
  const queries = [];
  quiet(() => {
    queries.push({ op: "range", path: "users", field: "age", args: [0, 100], paths: paths(b.range("users", "age", 0, 100)) });
    queries.push({ op: "equals", path: "users", field: "age", args: [62], paths: paths(b.equals("users", "age", 62)) });
    queries.push({ op: "equals", path: "users", field: "age", args: [41], paths: paths(b.equals("users", "age", 41)) });
    queries.push({ op: "range", path: "cfg", field: "v", args: [0, 10], paths: paths(b.range("cfg", "v", 0, 10)) });
    queries.push({ op: "range", path: "m", field: "x", args: [0, 10], paths: paths(b.range("m", "x", 0, 10)) });
    queries.push({ op: "count", path: "users", field: "age", args: [54], count: b.query.count("users", "age", 54) });
  });
  return { kind: "sync_mixed_values", source: "reference BulletNetworkSync._processSyncEntries + BulletQuery on a real Bullet (id 'w', network disabled)", id: "w",
    chunks, after, queries };
}

/* ------------------------------------------------------------------ g12: NODE-level semantics under multi-writer clocks (N4)
 * The reference's sync loop on objects whose clocks name several writers ({a, b, w}; this peer is 'w'), in every key order and with keys missing:
 * dominance with missing keys counted as 0, whole-node replacement, CONCURRENT clocks -> mergeValues (the stored object's fields, overlaid field by
 * field with compare(in, cur) >= 0 ? in : cur, new fields appended: src/bullet-crt.js:122-153) and the merged clock's key order ({...incoming}, then the
 * stored clock's other keys: :103-114), equal counters under different key sets (NOT identical: JSON.stringify :200-203 -> concurrent), the {w:2} of a
 * first sight against multi-writer clocks, several entries for one node in one chunk, deletions (a local write: the stored clock's own component is
 * incremented in place), strings and nested objects as fields. Store and per-path clock (as an ordered [key, counter] list) + source after every chunk. */
function genVcNodeSemantics() {
  const Sync = require(path.join(REF, "src", "bullet-network-sync.js"));
  const b = newBullet({ enableIndexing: true });
  const chunks = [
    [ /* chunk 1: first sight: every node stores {w:2}, whatever clock came with it */
      { path: "doc/a", data: { title: "A", rev: 1, size: 10 }, vectorClock: { a: 1, b: 0, w: 0 } },
      { path: "doc/b", data: { title: "B", rev: 1 }, vectorClock: { b: 4 } },
      { path: "doc/c", data: { rev: 1, tags: ["x"], meta: { k: 1 } }, vectorClock: { w: 1, a: 1 } },
      { path: "doc/d", data: { rev: 1, size: 5 }, vectorClock: {} },
      { path: "doc/e", data: { rev: 1 }, vectorClock: { a: 3, b: 3, w: 3 } },
      { path: "doc/f", data: { rev: 1, size: 1 }, vectorClock: { w: 9 } },
    ],
    [ /* chunk 2 against {w:2}: dominating (w >= 2 and something more), concurrent (w < 2 but another writer ahead), historical, equal counters with
         another key set (concurrent: merged), the identical clock (tie: incoming object) */
      { path: "doc/a", data: { title: "A2", rev: 2 }, vectorClock: { a: 2, w: 2 } },
      { path: "doc/b", data: { rev: 2, size: 7 }, vectorClock: { b: 5 } },
      { path: "doc/c", data: { rev: 0 }, vectorClock: { w: 1 } },
      { path: "doc/d", data: { rev: 3, extra: 1 }, vectorClock: { w: 2, a: 0 } },
      { path: "doc/e", data: { rev: 9, note: "tie" }, vectorClock: { w: 2 } },
      { path: "doc/f", data: { rev: 4 }, vectorClock: { b: 0, w: 2, a: 0 } },
    ],
    [ /* chunk 3: several entries per node in one chunk: concurrent after concurrent (the merged clock grows), then a dominating one that replaces the
         merged node, key orders that differ from the stored one */
      { path: "doc/b", data: { rev: 1, size: 9, who: "a" }, vectorClock: { a: 7 } },
      { path: "doc/b", data: { rev: 5, who: "w" }, vectorClock: { w: 3, b: 1 } },
      { path: "doc/b", data: { rev: 6 }, vectorClock: { w: 3, b: 5, a: 7 } },
      { path: "doc/b", data: { rev: 7, fin: 1 }, vectorClock: { a: 7, b: 5, w: 3 } },
      { path: "doc/a", data: { rev: 3, size: 11 }, vectorClock: { b: 1 } },
      { path: "doc/a", data: { rev: 1, size: 99, title: "old" }, vectorClock: { a: 1 } },
      { path: "doc/g", data: { rev: 1 }, vectorClock: { b: 2 } },
      { path: "doc/g", data: { rev: 2, size: 3 }, vectorClock: { b: 2 } },
      { path: "doc/g", data: { rev: 0, size: 8, name: "g" }, vectorClock: { a: 1 } },
    ],
    [ /* chunk 4: a deletion (local write: the stored clock's w component + 1), entries around it, a re-creation */
      { path: "doc/a", deleted: true, vectorClock: { w: 50 } },
      { path: "doc/a", data: { rev: 8 }, vectorClock: { a: 2, w: 3, b: 1 } },
      { path: "doc/a", data: { rev: 9, size: 1 }, vectorClock: { a: 2, b: 1, w: 3 } },
      { path: "doc/c", data: { rev: 5, size: 2 }, vectorClock: { a: 9, w: 2 } },
      { path: "doc/c", data: { size: 4, rev: 4 }, vectorClock: { b: 1, w: 2 } },
      { path: "doc/h", data: { rev: 1, size: 6 }, vectorClock: { a: 1, b: 1 } },
      { path: "doc/h", data: { rev: 2 }, vectorClock: { a: 1, b: 1, w: 1 } },
      { path: "doc/h", data: { rev: 3, size: 2 }, vectorClock: { b: 1, w: 2, a: 1 } },
    ],
    [ /* chunk 5: what a LOSING entry leaves behind in crt.vectorClocks (the merged clock, a fresh object with the incoming clock's keys first: :193-198):
         the deletion behind it — a local write — increments that copy, dominates and stores ITS key set; without a losing entry in front (doc/d) the
         deletion increments meta's own clock in place */
      { path: "doc/e", data: { rev: 0 }, vectorClock: { a: 0, w: 1 } },
      { path: "doc/e", deleted: true, vectorClock: { w: 1 } },
      { path: "doc/g", data: { rev: 9 }, vectorClock: { w: 1, b: 1 } },
      { path: "doc/g", deleted: true, vectorClock: { w: 1 } },
      { path: "doc/d", deleted: true, vectorClock: { w: 1 } },
      { path: "doc/g", data: { rev: 10, back: 1 }, vectorClock: { w: 3, b: 2, a: 1 } },
      { path: "doc/e", data: { rev: 1 }, vectorClock: { w: 1 } },      /* loses against the deleted node, whose null its _getData has turned into {} by then */
    ],
  ];
  const fake = { bullet: b };
  const after = [];
  quiet(() => {
    for (const c of chunks) {
      Sync.prototype._processSyncEntries.call(fake, JSON.parse(JSON.stringify(c)), "peer-1");
      const meta = {};
      for (const k of Object.keys(b.meta)) meta[k] = { clock: Object.keys(b.meta[k].vectorClock).map((w) => [w, b.meta[k].vectorClock[w]]), source: b.meta[k].source };
      after.push({ store: JSON.parse(JSON.stringify(b.store)), meta });
    }
  });
  const paths = (nodes) => nodes.map((n) => n.path);
  const queries = [];
  quiet(() => {
    queries.push({ op: "range", path: "doc", field: "rev", args: [0, 100], paths: paths(b.range("doc", "rev", 0, 100)) });
    queries.push({ op: "range", path: "doc", field: "size", args: [0, 100], paths: paths(b.range("doc", "size", 0, 100)) });
    queries.push({ op: "equals", path: "doc", field: "rev", args: [7], paths: paths(b.equals("doc", "rev", 7)) });
    queries.push({ op: "count", path: "doc", field: "size", args: [2], count: b.query.count("doc", "size", 2) });
  });
  return { kind: "vc_node_semantics", source: "reference BulletNetworkSync._processSyncEntries + BulletQuery on a real Bullet (id 'w', network disabled), multi-writer clocks", id: "w",
    writers: ["a", "b", "w"], chunks, after, queries };
}

/* ------------------------------------------------------------------ g7: a directory written by the reference's file storage (N3)
 * src/bullet-file-storage.js:170-210 writes store.json / meta.json / log.json; the files themselves are the fixture
 * (tests/golden/g7_storage_dir/). */
function genStorageDir(outDir) {
  const os = require("os");
  const Bullet = require(path.join(REF, "src", "bullet.js"));
  const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "bmx-g7-"));
  const b = quiet(() => new Bullet({ disableNetwork: true, storage: true, storageType: "file", storagePath: tmp, saveInterval: 0, enableIndexing: false, server: false }));
  b.id = "w";
  const rng = xorshift32(97);
  quiet(() => {
    for (let i = 0; i < 40; i++) b.setData("n/k" + i, { age: rng() % 90, score: (rng() % 2001) - 1000, __fromNetwork: true, __vectorClock: { w: 10 + (rng() % 50) } }, false);
    for (let i = 0; i < 40; i += 3) b.setData("n/k" + i, { age: rng() % 90, score: (rng() % 2001) - 1000, __fromNetwork: true, __vectorClock: { w: 100 + i } }, false);
    b.setData("cfg/title", "hello", false);
    b.setData("cfg/count", 3, false);
  });
  b.storage._saveData();
  fs.mkdirSync(outDir, { recursive: true });
  for (const f of ["store.json", "meta.json"]) fs.copyFileSync(path.join(tmp, f), path.join(outDir, f));
  if (b.storage.saveInterval) clearInterval(b.storage.saveInterval);
  console.log("wrote", outDir, "(store.json, meta.json as written by the reference's BulletFileStorage)");
}

/* ------------------------------------------------------------------ g13: INTEGER entries resolved against the sender's clock
 * The reference's sync and put paths never do this (a primitive arrives untagged and becomes a local write, src/bullet-network-sync.js:560-563), but
 * its public resolver does: crt.processUpdate(path, 5, {w: t}, currentValue, currentClock) — identical clocks are then decided BY VALUE (larger wins,
 * equal is a no-op: src/bullet-crt.js:200-233), against the {w: 2} a first sight leaves as against any other clock. This is what a direct
 * GpuCRT.mergeEntries call on integer entries has to reproduce. L0 harness (SURVEY 8(c)): the real BulletCRT, a Map path -> {value, clock}, the caller's
 * store rule (incoming || no current clock || concurrent). Entries on a path that has held an OBJECT are marked `mixed`: under identical clocks an
 * object beats any integer and any integer beats an object (compare() answers +1 for unordered pairs, :11-15), which no single row order expresses —
 * the device hands such entries back (`host`), and the fixture records what the reference does with them. */
function genIntegerEntryTies() {
  const crt = newCrt();
  const state = new Map();
  const kind = new Map();          // path -> "int" | "node": what the path has held so far (the device's rule for `mixed`)
  const chunks = [
    [ /* chunk 1: first sights (stored clock {w:2}, the entry's own clock is discarded), then ties against that {w:2}: larger, smaller, equal */
      { path: "cnt/a", data: 7, vectorClock: { w: 100 } },
      { path: "cnt/b", data: 7, vectorClock: { w: 100 } },
      { path: "cnt/c", data: 7, vectorClock: { w: 100 } },
      { path: "cnt/d", data: -5, vectorClock: { w: 1 } },
      { path: "cnt/a", data: 9, vectorClock: { w: 2 } },
      { path: "cnt/b", data: 3, vectorClock: { w: 2 } },
      { path: "cnt/c", data: 7, vectorClock: { w: 2 } },
      { path: "cnt/d", data: -6, vectorClock: { w: 2 } },
      { path: "cnt/d", data: -4, vectorClock: { w: 2 } },
    ],
    [ /* chunk 2: dominating and historical clocks, then ties on the new clock inside ONE chunk (several entries per path), negative and large values */
      { path: "cnt/a", data: 1, vectorClock: { w: 50 } },
      { path: "cnt/a", data: 0 - 1, vectorClock: { w: 50 } },
      { path: "cnt/a", data: 2, vectorClock: { w: 50 } },
      { path: "cnt/a", data: 2, vectorClock: { w: 50 } },
      { path: "cnt/b", data: 1000, vectorClock: { w: 1 } },
      { path: "cnt/c", data: 9007199254740991, vectorClock: { w: 3 } },
      { path: "cnt/c", data: 9007199254740990, vectorClock: { w: 3 } },
      { path: "cnt/c", data: -9007199254740991, vectorClock: { w: 4 } },
      { path: "cnt/e", data: 5, vectorClock: { w: 9 } },
      { path: "cnt/e", data: 6, vectorClock: { w: 2 } },
      { path: "cnt/e", data: 4, vectorClock: { w: 2 } },
    ],
    [ /* chunk 3: a tie after a tie, an equal value under a newer clock (it wins: the clock decides first), a long run on one path */
      { path: "cnt/a", data: 3, vectorClock: { w: 50 } },
      { path: "cnt/a", data: 3, vectorClock: { w: 51 } },
      { path: "cnt/a", data: 2, vectorClock: { w: 51 } },
      { path: "cnt/b", data: 8, vectorClock: { w: 2 } },
      { path: "cnt/b", data: 8, vectorClock: { w: 2 } },
      { path: "cnt/f", data: 1, vectorClock: { w: 7 } },
      { path: "cnt/f", data: 5, vectorClock: { w: 2 } },
      { path: "cnt/f", data: 3, vectorClock: { w: 2 } },
      { path: "cnt/f", data: 5, vectorClock: { w: 2 } },
      { path: "cnt/f", data: 6, vectorClock: { w: 1 } },
      { path: "cnt/f", data: 4, vectorClock: { w: 3 } },
      { path: "cnt/f", data: 9, vectorClock: { w: 3 } },
    ],
    [ /* chunk 4: nodes and integers on one path (`mixed`): handed back by the device; what the reference does with them is recorded all the same */
      { path: "mix/n", data: { v: 1 }, vectorClock: { w: 5 } },
      { path: "mix/n", data: 4, vectorClock: { w: 2 } },
      { path: "mix/i", data: 4, vectorClock: { w: 5 } },
      { path: "mix/i", data: { v: 2 }, vectorClock: { w: 2 } },
      { path: "cnt/a", data: 10, vectorClock: { w: 51 } },
    ],
  ];
  const after = [], decisions = [];
  quiet(() => {
    for (const c of chunks) {
      const ds = [];
      for (const e of c) {
        const isInt = typeof e.data === "number";
        const k = kind.get(e.path);
        if (k !== undefined && k !== (isInt ? "int" : "node")) e.mixed = true;
        const d = applyDelta(crt, state, e.path, e.vectorClock.w, JSON.parse(JSON.stringify(e.data)));
        ds.push(flagsOf(d));
        const held = state.get(e.path).value;
        kind.set(e.path, typeof held === "number" ? "int" : "node");
      }
      decisions.push(ds);
      const snap = {};
      for (const [p, v] of state) snap[p] = { value: JSON.parse(JSON.stringify(v.value)), clock: JSON.parse(JSON.stringify(v.clock)) };
      after.push(snap);
    }
  });
  return { kind: "integer_entry_ties", source: "reference BulletCRT.processUpdate over an entry list (L0 harness: Map path -> {value, clock}, store rule of src/bullet-crt.js:383), id 'w'", id: "w",
    chunks, decisions, after };
}

/* ------------------------------------------------------------------ main */
function write(name, obj) {
  const p = path.join(OUT, name);
  fs.writeFileSync(p, JSON.stringify(obj));
  console.log("wrote", p, fs.statSync(p).size, "bytes");
}
const BASE = { T0: 1000, DT: 1000, VR: 1000, VOFF: 0, F: 1, H: 16, hot_pct: 0, insert_pct: 0, unique: false, store_final: false };
const STREAMS = {
  /* config-1 shape of BASELINE.md: R=100k, D=10k, ts~U[1000,2000) / U[1000,3000), val~U[0,1000) */
  "g2_stream_cfg1_100k_10k.json": Object.assign({}, BASE, { seed: 12345, R: 100000, D: 10000 }),
  "g2_stream_unique_10k_10k.json": Object.assign({}, BASE, { seed: 777, R: 10000, D: 10000, unique: true, store_final: true }),
  "g2_stream_unique_insert_10k_4k.json": Object.assign({}, BASE, { seed: 778, R: 10000, D: 4000, unique: true, insert_pct: 10, store_final: true }),
  "g2_stream_hot30_10k_10k.json": Object.assign({}, BASE, { seed: 31337, R: 10000, D: 10000, hot_pct: 30, H: 10, store_final: true }),
  "g2_stream_insert10_1k_10k.json": Object.assign({}, BASE, { seed: 99, R: 1000, D: 10000, insert_pct: 10, store_final: true }),
  "g2_stream_ties_1k_10k.json": Object.assign({}, BASE, { seed: 5, R: 1000, D: 10000, T0: 0, DT: 2, VR: 3, VOFF: 1, insert_pct: 15, hot_pct: 20, H: 5, store_final: true }),
  "g2_stream_multifield_1k_1k.json": Object.assign({}, BASE, { seed: 2024, R: 4000, D: 4000, F: 4, insert_pct: 5, store_final: true }),
  "g2_stream_mixed_100k_10k.json": Object.assign({}, BASE, { seed: 8086, R: 100000, D: 10000, insert_pct: 10, hot_pct: 30, H: 100, VR: 2000, VOFF: 1000 }),
  "g2_stream_empty_start_0_1k.json": Object.assign({}, BASE, { seed: 3, R: 0, D: 1000, insert_pct: 100, ins_space: 300, store_final: true }),
};

function main() {
  fs.mkdirSync(OUT, { recursive: true });
  write("g1_decision_table.json", genDecisionTable());
  write("g3_sequences.json", genSequences());
  for (const [name, spec] of Object.entries(STREAMS)) write(name, runStream(spec));
  write("g6_vc_unique_2k.json", runVcStream({ seed: 61, R: 2000, D: 1500, CMAX: 4, VR: 5, VOFF: 2, insert_pct: 10, hot_pct: 0, H: 1, ins_space: 100000 }));
  write("g6_vc_dups_500.json", runVcStream({ seed: 62, R: 500, D: 4000, CMAX: 3, VR: 3, VOFF: 1, insert_pct: 15, hot_pct: 30, H: 8, ins_space: 60 }));
  write("g6_vc_empty_start.json", runVcStream({ seed: 63, R: 0, D: 1500, CMAX: 3, VR: 3, VOFF: 1, insert_pct: 100, hot_pct: 0, H: 1, ins_space: 200 }));
  write("g11_vc_keysets_2k.json", runVcKeysetStream({ seed: 111, R: 2000, D: 3000, CMAX: 4, VR: 5, VOFF: 2, insert_pct: 10, hot_pct: 0, H: 1, ins_space: 100000, full_pct: 30 }));
  write("g11_vc_keysets_hot.json", runVcKeysetStream({ seed: 112, R: 300, D: 3000, CMAX: 40, VR: 3, VOFF: 1, insert_pct: 10, hot_pct: 40, H: 6, ins_space: 40, full_pct: 20 }));
  write("g4_l1_ops.json", genL1());
  write("g5_query_example.json", genQueryExample());
  write("g5_query_seeded_2k.json", genQuerySeeded(2000, 4711, true));
  write("g5_query_seeded_100k.json", genQuerySeeded(100000, 4712, false));
  write("g8_sync_chunk.json", genSyncChunk());
  write("g9_sync_node_semantics.json", genSyncNodeSemantics());
  write("g10_sync_mixed_values.json", genSyncMixedValues());
  write("g12_vc_node_semantics.json", genVcNodeSemantics());
  write("g13_entries_integer_ties.json", genIntegerEntryTies());
  genStorageDir(path.join(OUT, "g7_storage_dir"));
}

module.exports = { genStream, rowId, rowField, xorshift32, splitmix64, fnv1a32, rowDigest };
if (require.main === module) main();
