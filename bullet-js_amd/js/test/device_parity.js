"use strict";
/*
 * device_parity.js — GPU test of the JS host over the N-API addon: GpuCRT.mergeBatch/mergeEntries and GpuQuery's
 * device indices against golden vectors produced by the real reference (tests/golden). Needs an MI355X.
 * Usage: node device_parity.js <golden dir>
 */
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const { GpuCRT, GpuQuery, attach, hash } = require("..");
const MiniBullet = require("./mini-bullet");
const gen = require(path.join(__dirname, "..", "..", "..", "oracle", "gen_golden.js"));   // test infrastructure: stream generator only

const GOLD = process.argv[2] || path.join(__dirname, "..", "..", "..", "tests", "golden");
const load = (n) => JSON.parse(fs.readFileSync(path.join(GOLD, n), "utf8"));
let checks = 0;
const STRESS_SALT = parseInt(process.env.BMX_STRESS_SALT || "0", 10);   // other seeds for the differential stress tests (default: the pinned ones)

function columns(rows, F) {
  const n = rows.length;
  const c = { id: new BigUint64Array(n), field: new Uint32Array(n), ts: new BigInt64Array(n), val: new BigInt64Array(n) };
  rows.forEach((r, i) => { c.id[i] = gen.rowId(r.row, F); c.field[i] = gen.rowField(r.row, F); c.ts[i] = BigInt(r.ts); c.val[i] = BigInt(r.val); });
  return c;
}

/* G2 streams through GpuCRT.mergeBatch */
for (const name of fs.readdirSync(GOLD).filter((f) => f.startsWith("g2_stream_")).sort()) {
  const g = load(name);
  const { resident, deltas, F } = gen.genStream(g.spec);
  const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: Math.max(4096, 2 * (g.spec.R + g.spec.D)) });
  const rc = columns(resident, F);
  crt.graph.loadRows(rc);
  const r = crt.mergeBatch(columns(deltas, F));
  assert.deepStrictEqual(Array.from(r.applied), g.winners, name + " winners");
  assert.strictEqual(r.nRows, g.n_rows_final, name + " rows");
  if (r.nConflicts === 0) assert.strictEqual(Buffer.from(r.flags).toString("base64"), g.flags_b64, name + " flags");
  else {   // duplicates in the batch: the strict mode reproduces the reference's per-delta flags exactly
    const crt2 = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: Math.max(4096, 2 * (g.spec.R + g.spec.D)) });
    crt2.graph.loadRows(rc);
    const r2 = crt2.mergeBatch(columns(deltas, F), { strictFlags: true });
    assert.strictEqual(Buffer.from(r2.flags).toString("base64"), g.flags_b64, name + " strict flags");
    assert.deepStrictEqual(Array.from(r2.applied), g.winners, name + " strict winners");
    crt2.close();
    checks += 2;
  }
  const d = crt.graph.dumpRows();
  let digest = 0n;
  for (let i = 0; i < d.id.length; i++) digest = (digest + gen.rowDigest(d.id[i], d.field[i], d.ts[i], d.val[i])) & ((1n << 64n) - 1n);
  assert.strictEqual(digest.toString(16), g.digest, name + " state digest");
  crt.close();
  checks += 4;
}

/* Page-locked column sets from the graph's pool (DeviceGraph.takeColumns -> the addon's hostColumns -> bmx_host_alloc): one allocation behind
 * four typed arrays at byte offsets, reused after giveColumns; a G2 stream merged from them gives the reference's winners and state. */
{
  const name = "g2_stream_mixed_100k_10k.json";
  const g = load(name);
  const { resident, deltas, F } = gen.genStream(g.spec);
  const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: 2 * (g.spec.R + g.spec.D) });
  const graph = crt.graph;
  const fill = (rows) => {
    const cols = graph.takeColumns(rows.length);
    rows.forEach((r, i) => { const id = gen.rowId(r.row, F); cols.set2(i, Number(id & 0xffffffffn), Number(id >> 32n), gen.rowField(r.row, F), r.ts, r.val); });
    return cols;
  };
  const rc = fill(resident);
  assert.ok(rc._pooled && rc.n >= resident.length && (rc.n & (rc.n - 1)) === 0, "pooled, power-of-two capacity");
  if (graph._pinned) assert.ok(rc.id.buffer === rc.field.buffer && rc.ts.byteOffset === 8 * rc.n && rc.field.byteOffset === 24 * rc.n, "one page-locked allocation behind the four columns");
  graph.loadRows(rc.slice(resident.length));
  graph.giveColumns(rc.slice(resident.length));
  assert.strictEqual(graph.takeColumns(resident.length), rc, "an idle column set is reused");
  graph.giveColumns(rc);
  graph.giveColumns(rc);                                     // a second give is ignored
  const dc = fill(deltas);
  assert.ok(dc !== rc && dc.id.buffer !== rc.id.buffer, "a set of its own for another size");
  const r = crt.mergeBatch(dc.slice(deltas.length));
  assert.deepStrictEqual(Array.from(r.applied), g.winners, name + " winners from page-locked columns");
  assert.strictEqual(r.nRows, g.n_rows_final);
  const d = graph.dumpRows();
  let digest = 0n;
  for (let i = 0; i < d.id.length; i++) digest = (digest + gen.rowDigest(d.id[i], d.field[i], d.ts[i], d.val[i])) & ((1n << 64n) - 1n);
  assert.strictEqual(digest.toString(16), g.digest, name + " state digest");
  assert.ok(graph.takeColumns(100)._pooled === undefined, "small sets are plain arrays");
  crt.close();
  checks += 8;
}

/* Sharded graph (bmx_comm_*): one handle owns N shards (logical shards on this box's one GPU). Winners in the caller's index
 * space, row count and state digest must equal the reference's, and so must the queries, which now run on every shard. */
for (const shards of [2, 4, 8]) {
  for (const name of ["g2_stream_hot30_10k_10k.json", "g2_stream_mixed_100k_10k.json", "g2_stream_multifield_1k_1k.json"]) {
    if (!fs.existsSync(path.join(GOLD, name))) continue;
    const g = load(name);
    const { resident, deltas, F } = gen.genStream(g.spec);
    const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: Math.max(8192, 2 * (g.spec.R + g.spec.D)), shards });
    crt.graph.loadRows(columns(resident, F));
    const r = crt.mergeBatch(columns(deltas, F));
    assert.deepStrictEqual(Array.from(r.applied), g.winners, name + " winners over " + shards + " shards");
    assert.strictEqual(r.nRows, g.n_rows_final, name + " rows");
    const d = crt.graph.dumpRows();
    let digest = 0n;
    for (let i = 0; i < d.id.length; i++) digest = (digest + gen.rowDigest(d.id[i], d.field[i], d.ts[i], d.val[i])) & ((1n << 64n) - 1n);
    assert.strictEqual(digest.toString(16), g.digest, name + " state digest over " + shards + " shards");
    crt.close();
    checks += 3;
  }
}
{
  const g = load("g5_query_seeded_2k.json");
  const rng = gen.xorshift32(g.seed);
  const b = new MiniBullet("w");
  const { query } = attach(b, { capacityRows: 1 << 16, shards: 4 });
  for (let i = 0; i < g.N; i++) {
    const age = rng() % 100, score = (rng() % 200001) - 100000;
    b.setData("n/k" + i, { age, score, __fromNetwork: true, __vectorClock: { w: 10 + (i % 7) } }, false);
  }
  const ord = (nodes) => nodes.map((n) => parseInt(n.path.split("/").pop().slice(1), 10));
  for (const q of g.queries) {
    let got;
    if (q.op === "equals") got = ord(query.equals("n", q.field, q.args[0]));
    else if (q.op === "range") got = ord(query.range("n", q.field, q.args[0], q.args[1]));
    else if (q.op === "count") { assert.strictEqual(query.count("n", q.field, q.args[0]), q.count); checks++; continue; }
    else if (q.op === "filter_and") got = ord(query.filterWhere("n", [{ field: "age", min: q.args[0][0], max: q.args[0][1] }, { field: "score", min: q.args[1][0], max: q.args[1][1] }]));
    assert.strictEqual(query.lastPath, "device", JSON.stringify(q));
    if (q.op === "filter_and") assert.deepStrictEqual(got.slice().sort((a, b) => a - b), q.ordinals.slice().sort((a, b) => a - b));
    else assert.deepStrictEqual(got, q.ordinals, "reference order over 4 shards " + JSON.stringify(q.args));
    checks++;
  }
  b.close();
}

/* G5: integer indices on the device, reference order included */
{
  const g = load("g5_query_seeded_2k.json");
  const rng = gen.xorshift32(g.seed);
  const b = new MiniBullet("w");
  const { query } = attach(b, { capacityRows: 1 << 16 });
  for (let i = 0; i < g.N; i++) {
    const age = rng() % 100, score = (rng() % 200001) - 100000;
    b.setData("n/k" + i, { age, score, __fromNetwork: true, __vectorClock: { w: 10 + (i % 7) } }, false);
  }
  const ord = (nodes) => nodes.map((n) => parseInt(n.path.split("/").pop().slice(1), 10));
  for (const q of g.queries) {
    let got;
    if (q.op === "equals") got = ord(query.equals("n", q.field, q.args[0]));
    else if (q.op === "range") got = ord(query.range("n", q.field, q.args[0], q.args[1]));
    else if (q.op === "count") { assert.strictEqual(query.count("n", q.field, q.args[0]), q.count); assert.strictEqual(query.lastPath, "device"); checks++; continue; }
    else if (q.op === "filter_and") got = ord(query.filterWhere("n", [{ field: "age", min: q.args[0][0], max: q.args[0][1] }, { field: "score", min: q.args[1][0], max: q.args[1][1] }]));
    assert.strictEqual(query.lastPath, "device", JSON.stringify(q));
    if (q.op === "filter_and") assert.deepStrictEqual(got.slice().sort((a, b) => a - b), q.ordinals.slice().sort((a, b) => a - b));
    else assert.deepStrictEqual(got, q.ordinals, "reference order " + JSON.stringify(q.args));   // exact order of the reference
    checks++;
  }
  // a write makes the index stale; the next query sees it (fresh-index semantics)
  b.setData("n/k5", { age: 42, score: 1, __fromNetwork: true, __vectorClock: { w: 999 } }, false);
  assert.ok(ord(query.equals("n", "age", 42)).includes(5));
  assert.strictEqual(query.equals("n", "age", "42").length, query.equals("n", "age", 42).length);   // "42" and 42 share a bucket
  assert.deepStrictEqual(query.range("n", "age", 30, undefined), []);
  b.close();
  checks += 3;
}

/* G5 example dataset: ints on the device, strings on the host, through the same facade */
{
  const g = load("g5_query_example.json");
  const b = new MiniBullet("w");
  const { query } = attach(b, { capacityRows: 4096 });
  for (const [k, v] of Object.entries(g.users)) b.get("users/" + k).put(v);
  for (const [k, v] of Object.entries(g.products)) b.get("products/" + k).put(v);
  const keys = (nodes) => nodes.map((n) => n.path.split("/").pop());
  for (const q of g.queries) {
    if (q.op === "range") {
      const hi = q.args[3] === "Infinity" ? Infinity : q.args[3];
      assert.deepStrictEqual(keys(query.range(q.args[0], q.args[1], q.args[2], hi)), q.keys, JSON.stringify(q.args));
      assert.strictEqual(query.lastPath, "device");
    } else if (q.op === "equals") {
      assert.deepStrictEqual(keys(query.equals(q.args[0], q.args[1], q.args[2])), q.keys, JSON.stringify(q.args));
    } else if (q.op === "count") {
      assert.strictEqual(query.count(q.args[0], q.args[1], q.args[2]), q.n);
    } else if (q.op === "range_undefined_max") {
      assert.deepStrictEqual(query.range(q.args[0], q.args[1], q.args[2], undefined), []);
    }
    checks++;
  }
  assert.deepStrictEqual(keys(query.equals("users", "role", "admin")), ["user1", "user6", "user10"]);
  assert.strictEqual(query.lastPath, "host");
  b.close();
}

/* sync-chunk adapter: every eligible entry is ONE delta on its node's clock row; off-contract entries are handed back */
{
  const b = new MiniBullet("w");
  const { crt } = attach(b, { capacityRows: 4096 });
  const entries = [
    { path: "s/a", data: { n: 1, m: 5 }, vectorClock: { w: 10 } },
    { path: "s/b", data: 7, vectorClock: { w: 3 } },
    { path: "s/a", data: { n: 2 }, vectorClock: { w: 9 } },          // older than the first (delta mode keeps the incoming clock): historical
    { path: "s/c", data: { name: "x", k: 6 }, vectorClock: { w: 4 } },   // a string field: part of the node on the host, no device row of its own
    { path: "s/d", data: { n: 1 }, vectorClock: { w: 4, q: 1 } },    // multi-writer clock: host path
    { path: "s/a", data: { n: 3 }, vectorClock: { w: 11 } },
    { path: "s/e", data: ["x"], vectorClock: { w: 4 } },             // an array, a string, an empty object: host path
    { path: "s/f", data: "x", vectorClock: { w: 4 } },
    { path: "s/g", data: {}, vectorClock: { w: 4 } },
    { path: "s/h", data: { n: 1 }, vectorClock: { q: 4 } },          // somebody else's clock: host path (the list of handed-back entries stays in entry order)
  ];
  const r = crt.mergeEntries(entries, { insertMode: "delta" });
  assert.deepStrictEqual(r.host, [4, 6, 7, 8, 9]);
  assert.deepStrictEqual(r.applied, [{ entry: 1, field: null }, { entry: 3, field: null }, { entry: 5, field: null }]);   // one winner per NODE: the object that is its final value
  assert.strictEqual(r.nRows, 3);                                    // three clock rows (the winners' value rows are still queued)
  const snap = crt.checkpoint();                                     // flushes them: s/a {n:3}, s/b 7, s/c {k:6}
  assert.deepStrictEqual(snap.filter((x) => x.field !== hash.NODE_CLOCK).map((x) => [x.path, x.field, x.ts, x.val]).sort(), [["s/a", "n", 11, 3], ["s/b", null, 3, 7], ["s/c", "k", 4, 6]]);
  b.close();
  checks += 4;
}

/* N1 + N3: winners applied to the store in one pass; checkpoint / restore of the device rows */
{
  const b = new MiniBullet("w");
  const applied = [];
  b._applyUpdate = (path, value, clock, fromNetwork) => {    // the facade hook the reference exposes (src/bullet.js:184)
    applied.push([path, value, clock.w, fromNetwork]);
    const segs = path.split("/"); let node = b.store;
    for (const s of segs.slice(0, -1)) { if (!node[s]) node[s] = {}; node = node[s]; }
    node[segs[segs.length - 1]] = value; b.meta[path] = { source: "network", vectorClock: clock };
  };
  const { crt } = attach(b, { capacityRows: 4096 });
  const entries = [
    { path: "acct/a", data: { bal: 10, seq: 1 }, vectorClock: { w: 100 } },
    { path: "acct/b", data: { bal: 20 }, vectorClock: { w: 100 } },
    { path: "acct/a", data: { bal: 15 }, vectorClock: { w: 101 } },
  ];
  const r1 = crt.mergeEntries(entries, { apply: "each" });   // the facade's own per-write hook
  // reference insert rule: a first write stores clock {w:2}; the later object (101 > 2) dominates and REPLACES the node: seq is gone
  assert.deepStrictEqual(applied, [["acct/b", { bal: 20 }, 2, true], ["acct/a", { bal: 15 }, 101, true]]);
  assert.deepStrictEqual(b.store, { acct: { a: { bal: 15 }, b: { bal: 20 } } });
  assert.strictEqual(r1.nRows, 2);
  const snap = crt.checkpoint().sort((x, y) => (x.path + x.field < y.path + y.field ? -1 : 1));
  assert.deepStrictEqual(snap.map((x) => [x.path, x.collection, x.field, x.ts, x.field === hash.NODE_CLOCK ? "seq" : x.val]),
    [["acct/a", "acct", hash.NODE_CLOCK, 101, "seq"], ["acct/a", "acct", "bal", 101, 15], ["acct/b", "acct", hash.NODE_CLOCK, 2, "seq"], ["acct/b", "acct", "bal", 2, 20]]);
  b.close();
  const b2 = new MiniBullet("w");
  const h2 = attach(b2, { capacityRows: 4096 });
  assert.strictEqual(h2.crt.restore(snap), 4);
  const again = h2.crt.mergeEntries([{ path: "acct/a", data: { bal: 1 }, vectorClock: { w: 50 } }, { path: "acct/b", data: { bal: 19 }, vectorClock: { w: 2 } }]);
  assert.deepStrictEqual(again.applied, [{ entry: 1, field: null }]);      // 50 < 101 is historical; {w:2} ties with the stored clock and the INCOMING object wins, smaller value or not
  b2.close();
  checks += 6;
}

/* N2: the reference's sync-chunk loop, batched. Fixture generated by running the REAL BulletNetworkSync._processSyncEntries
 * (tests/golden/g8_sync_chunk.json): final store, clock + source of EVERY path — including the tagging quirk (primitives arrive untagged,
 * are treated as local writes and always accepted) and a node created on the device and deleted through the host path. */
{
  const g = load("g8_sync_chunk.json");
  const b = new MiniBullet("w");
  const { crt, sync } = attach(b, { capacityRows: 4096, batchSync: true });
  for (const chunk of g.chunks) sync.processSyncEntries(JSON.parse(JSON.stringify(chunk)), "peer-1");
  assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), g.store, "store after three sync chunks");
  assert.deepStrictEqual(Object.keys(b.meta).sort(), Object.keys(g.meta).sort(), "the same paths carry a clock");
  for (const p of Object.keys(g.meta)) {
    assert.deepStrictEqual(b.meta[p].vectorClock, g.meta[p].vectorClock, "clock of " + p);
    assert.strictEqual(b.meta[p].source, g.meta[p].source, "source of " + p);
    checks++;
  }
  assert.strictEqual(b.meta["cfg/limit"].source, "local");          // the quirk: a primitive from the network is a local write
  assert.strictEqual(b.store.acct.c, null);
  assert.ok(sync.stats.deviceBatches >= 3 && sync.stats.hostEntries >= 7, JSON.stringify(sync.stats));
  // network puts (batching off: applied as they arrive, like the reference)
  sync.handlePut("peer-2", { path: "acct/a", data: { bal: 500, seq: 9, __vectorClock: { w: 151 } } });
  sync.handlePut("peer-2", { path: "acct/a", data: { bal: 1, seq: 1, __vectorClock: { w: 3 } } });
  sync.handlePut("peer-2", { path: "cfg/name", data: "gamma" });
  assert.deepStrictEqual(b.store.acct.a, { bal: 500, seq: 9 });
  assert.strictEqual(b.store.cfg.name, "gamma");
  // ... and the device followed the host-path write: a sync entry at the put's clock ties and wins, an older one is historical
  sync.processSyncEntries([{ path: "acct/a", data: { bal: 2 }, vectorClock: { w: 150 } }, { path: "acct/a", data: { bal: 3 }, vectorClock: { w: 151 } }], "peer-1");
  assert.deepStrictEqual(b.store.acct.a, { bal: 3 });
  b.close();
  checks += 7;
}

/* N2/N4: NODE-level semantics, pinned on tests/golden/g9_sync_node_semantics.json (the reference's own loop): shrinking and growing field
 * sets, ties on the stored clock (also the {w:2} of a first sight), deletions and re-creations, a node that starts on the host path. Store and
 * every path's clock + source after EVERY chunk; then the reference's query answers from a store-sourced index (reference order) and from a
 * device-sourced one (the rows the batches and the host-path mirror left on the GPU: deleted nodes and replaced fields must be gone). */
/* ... and on tests/golden/g10_sync_mixed_values.json: objects with string / nested / array / boolean / null fields take the same device path (the node's
 * clock is resolved on the GPU, the object stays on the host, only its integer fields become device rows); an object that meets a stored STRING under
 * an identical clock is the host's (compared as text by the reference). */
for (const [fixture, batchPuts, minDevice, minHost, lazyStore] of [["g9_sync_node_semantics.json", false, 23, 6], ["g9_sync_node_semantics.json", true, 23, 6],
  ["g10_sync_mixed_values.json", false, 16, 9], ["g10_sync_mixed_values.json", true, 15, 8],
  ["g9_sync_node_semantics.json", false, 23, 6, true], ["g10_sync_mixed_values.json", false, 16, 9, true], ["g10_sync_mixed_values.json", true, 15, 8, { idleSlice: 0 }]]) {   // ... and with the store following lazily (lazy-store.js)
  const g = load(fixture);
  const g9 = fixture.slice(0, fixture.indexOf("_")) + (lazyStore ? " (lazy store)" : "");
  const b = new MiniBullet(g.id);
  const { crt, query, sync } = attach(b, { capacityRows: 4096, batchSync: { batchPuts, lazyStore } });
  g.chunks.forEach((chunk, ci) => {
    if (batchPuts && ci === 2) {     // the same entries as network puts: queued, merged when the queue is flushed (here: by the next chunk's first use)
      for (const e of chunk) {
        if (e.deleted || typeof e.data !== "object") sync.processSyncEntries([JSON.parse(JSON.stringify(e))], "peer-1");
        else sync.handlePut("peer-1", { path: e.path, data: Object.assign({}, e.data, { __vectorClock: e.vectorClock }) });
      }
      sync.flush();
    } else sync.processSyncEntries(JSON.parse(JSON.stringify(chunk)), "peer-1");
    const want = g.after[ci];
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), want.store, g9 + " store after chunk " + (ci + 1));
    assert.deepStrictEqual(Object.keys(b.meta).sort(), Object.keys(want.meta).sort(), g9 + " paths with a clock after chunk " + (ci + 1));
    for (const p of Object.keys(want.meta)) {
      assert.deepStrictEqual(b.meta[p].vectorClock, want.meta[p].vectorClock, g9 + " clock of " + p + " after chunk " + (ci + 1));
      assert.strictEqual(b.meta[p].source, want.meta[p].source, g9 + " source of " + p + " after chunk " + (ci + 1));
      checks++;
    }
  });
  assert.ok(sync.stats.deviceEntries >= minDevice && sync.stats.hostEntries >= minHost, fixture + " " + JSON.stringify(sync.stats));
  const G2 = require("../gpu-query");
  const dq = new G2({ id: "w", _getData: (p) => b._getData(p), setData() {}, get: (p) => b.get(p) }, { graph: crt.graph });   // a second engine on the same device rows
  for (const q of g.queries) {
    if (q.op === "count") {
      assert.strictEqual(query.count(q.path, q.field, q.args[0]), q.count, g9 + " count");
      dq.index(q.path, q.field, { source: "device" });
      assert.strictEqual(dq.count(q.path, q.field, q.args[0]), q.count, g9 + " count (device rows)");
    } else {
      const got = q.op === "range" ? query.range(q.path, q.field, q.args[0], q.args[1]) : query.equals(q.path, q.field, q.args[0]);
      assert.deepStrictEqual(got.map((n) => n.path), q.paths, g9 + " " + q.op + " " + q.field + " (store-sourced index, reference order)");
      dq.index(q.path, q.field, { source: "device" });
      const dev = q.op === "range" ? dq.range(q.path, q.field, q.args[0], q.args[1]) : dq.equals(q.path, q.field, q.args[0]);
      assert.deepStrictEqual(dev.map((n) => n.path).sort(), q.paths.slice().sort(), g9 + " " + q.op + " " + q.field + " (device-sourced index)");
    }
    checks += 2;
  }
  b.close();
}

/* N3 (device side): a reference-written storage directory -> device rows; rows that reach the device through typed columns only
 * are folded back into store.json / meta.json on save */
{
  const os = require("os");
  const { GpuStorage } = require("..");
  const src = path.join(GOLD, "g7_storage_dir");
  const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "bmx-n3d-"));
  for (const f of ["store.json", "meta.json"]) fs.copyFileSync(path.join(src, f), path.join(tmp, f));
  const b = new MiniBullet("w");
  b.storage = new GpuStorage(b, { path: tmp, saveInterval: 0 });
  const { crt } = attach(b, { capacityRows: 4096 });
  const wantStore = JSON.parse(fs.readFileSync(path.join(src, "store.json"), "utf8")), wantMeta = JSON.parse(fs.readFileSync(path.join(src, "meta.json"), "utf8"));
  const g = crt.graph;                                               // created now: seeded from what the storage loaded
  assert.strictEqual(g.rowCount(), 40 * 3 + 1 + 3);                   // 40 nodes x {clock, age, score} + cfg/title (clock only: a string) + cfg/count (clock, value, field of cfg)
  const ids = new BigUint64Array(80), id32 = new Uint32Array(ids.buffer), fields = new Uint32Array(80);
  let k = 0;
  for (let i = 0; i < 40; i++) for (const f of ["age", "score"]) { const id = g.keys.idOf("n/k" + i); id32[2 * k] = id[0]; id32[2 * k + 1] = id[1]; fields[k] = g.keys.fieldOf("n", f); k++; }
  const rows = g.getRows(ids, fields);
  k = 0;
  for (let i = 0; i < 40; i++) for (const f of ["age", "score"]) {
    assert.strictEqual(rows.found[k], 1);
    assert.strictEqual(Number(rows.val[k]), wantStore.n["k" + i][f]);
    assert.strictEqual(Number(rows.ts[k]), wantMeta["n/k" + i].vectorClock.w);
    k++;
  }
  // a sync chunk is resolved against the restored NODE clocks: 1 is historical everywhere, 1000 dominates and replaces the node
  const r = crt.mergeEntries([{ path: "n/k0", data: { age: 1 }, vectorClock: { w: 1 } }, { path: "n/k1", data: { age: 77 }, vectorClock: { w: 1000 } }], { apply: true });
  assert.deepStrictEqual(r.applied, [{ entry: 1, field: null }]);
  // rows that arrive as typed columns only (no facade write) are folded into the files on save
  const cols = new hash.Columns(1);
  cols.set(0, g.keys.idOf("n/k2"), g.keys.fieldOf("n", "score"), 5000, 4242);
  crt.mergeBatch(cols);
  b.storage.save();
  const savedStore = JSON.parse(fs.readFileSync(path.join(tmp, "store.json"), "utf8")), savedMeta = JSON.parse(fs.readFileSync(path.join(tmp, "meta.json"), "utf8"));
  assert.strictEqual(savedStore.n.k2.score, 4242);
  assert.deepStrictEqual(savedMeta["n/k2/score"].vectorClock, { w: 5000 });
  assert.deepStrictEqual(savedStore.n.k1, { age: 77 });
  assert.deepStrictEqual(savedMeta["n/k1"].vectorClock, { w: 1000 });
  assert.deepStrictEqual(savedStore.n.k0, wantStore.n.k0);
  assert.deepStrictEqual(Object.keys(savedMeta).sort(), Object.keys(wantMeta).concat(["n/k2/score"]).sort());   // nothing else was folded: mirrored rows are recognised
  b.close();
  checks += 9 + 80;
}

/* device-sourced index: rows ingested by mergeEntries are queried without re-uploading anything from the JS store */
{
  const b = new MiniBullet("w");
  const { crt, query } = attach(b, { capacityRows: 4096 });
  const entries = [];
  for (let i = 0; i < 500; i++) entries.push({ path: "dev/n" + i, data: { age: i % 50, score: 1000 - i }, vectorClock: { w: 10 } });
  entries.push({ path: "dev/n7", data: { age: 49 }, vectorClock: { w: 11 } });     // newer clock: n7 is replaced, its age moves from 7 to 49
  crt.mergeEntries(entries, { insertMode: "delta", apply: true });
  query.index("dev", "age", { source: "device" });
  const keys = (nodes) => nodes.map((n) => n.path.split("/").pop());
  const want49 = []; for (let i = 0; i < 500; i++) if (i % 50 === 49 || i === 7) want49.push("n" + i);
  assert.deepStrictEqual(keys(query.equals("dev", "age", 49)), want49.sort());
  assert.strictEqual(query.lastPath, "device");
  assert.strictEqual(query.range("dev", "age", 0, 4).length, 50);
  assert.strictEqual(query.count("dev", "age", 7), 9);                              // n7 moved away from 7
  query.index("dev", "score", { source: "device" });
  assert.strictEqual(query.count("dev", "score", 993), 0);                          // ... and its score went with the replaced node (a tombstone on the device)
  assert.strictEqual(query.count("dev", "score", 992), 1);
  const before = crt.graph.indexRefreshCounts();
  crt.mergeEntries([{ path: "dev/n1", data: { age: 49 }, vectorClock: { w: 12 } }, { path: "dev/fresh", data: { age: 49 }, vectorClock: { w: 12 } }], { insertMode: "delta", apply: true });
  const got49 = keys(query.equals("dev", "age", 49));
  assert.ok(got49.includes("n1") && got49.includes("fresh"));                       // changed row and created row, both through the change log:
  const after = crt.graph.indexRefreshCounts();
  assert.strictEqual(after.fullBuilds, before.fullBuilds);                          // the indexes were not rebuilt from the table
  assert.ok(after.incremental > before.incremental);
  b.close();
  checks += 9;
}

/* the same device-sourced index with a VALUE-ORDERED VIEW on the device (opts.ordered: bmx_index_set_ordered — the reference's index is a Map keyed by
 * value, src/bullet-query.js:30-73): equals / range / count answer from a sorted copy of the column, which a write to the field patches (the first query
 * after it) instead of throwing it away. Same answers as the plain index whichever path replied; both store-sourced and device-sourced kinds. */
{
  const b = new MiniBullet("w");
  const { crt, query } = attach(b, { capacityRows: 8192 });
  const entries = [];
  for (let i = 0; i < 2000; i++) entries.push({ path: "ov/n" + i, data: { age: (i * 7) % 90, score: 3000 - i }, vectorClock: { w: 10 } });
  crt.mergeEntries(entries, { insertMode: "delta", apply: true });
  query.index("ov", "age", { source: "device", ordered: 2 });
  const f = crt.graph.keys.fieldOf("ov", "age");
  const keys = (nodes) => nodes.map((n) => n.path.split("/").pop()).sort();
  const wantAge = (lo, hi) => { const o = []; for (let i = 0; i < 2000; i++) { const a = b.store.ov["n" + i] && b.store.ov["n" + i].age; if (a >= lo && a <= hi) o.push("n" + i); } if (b.store.ov.extra && b.store.ov.extra.age >= lo && b.store.ov.extra.age <= hi) o.push("extra"); return o.sort(); };
  assert.deepStrictEqual(keys(query.equals("ov", "age", 35)), wantAge(35, 35));          // query 1 since the build: the column scan
  assert.strictEqual(crt.graph.indexOrderedInfo(f).valid, 0);
  assert.deepStrictEqual(keys(query.range("ov", "age", 10, 12)), wantAge(10, 12));       // query 2: the view is sorted
  let info = crt.graph.indexOrderedInfo(f);
  assert.ok(info.valid === 1 && info.sorts === 1 && info.afterQueries === 4, JSON.stringify(info));   // (a query here is a count + a fetch on the device)
  for (const [lo, hi] of [[0, 0], [0, 89], [89, 89], [40, 60], [-5, 2], [90, 200]]) assert.deepStrictEqual(keys(query.range("ov", "age", lo, hi)), wantAge(lo, hi));
  assert.strictEqual(query.count("ov", "age", 35), wantAge(35, 35).length);
  assert.strictEqual(crt.graph.indexOrderedInfo(f).sorts, 1);                            // all from the one sort
  crt.mergeEntries([{ path: "ov/n5", data: { age: 88 }, vectorClock: { w: 12 } }, { path: "ov/extra", data: { age: 35 }, vectorClock: { w: 12 } }], { insertMode: "delta", apply: true });
  assert.deepStrictEqual(keys(query.equals("ov", "age", 35)), wantAge(35, 35));          // right after the write: the refresh PATCHED the view (round 5: the reference moves the path between value buckets on the write, src/bullet-query.js:139-176), "extra" is there
  assert.strictEqual(crt.graph.indexOrderedInfo(f).valid, 1);
  assert.deepStrictEqual(keys(query.equals("ov", "age", 88)), wantAge(88, 88));
  info = crt.graph.indexOrderedInfo(f);
  assert.ok(info.valid === 1 && info.sorts === 1 && info.patches >= 1 && info.keysPatched >= 3, JSON.stringify(info));   // still the one sort: the change run (n5's old and new key, "extra") was merged in
  assert.deepStrictEqual(keys(query.range("ov", "age", 30, 40)), wantAge(30, 40));
  crt.graph.indexSetOrdered(f, 0);
  assert.deepStrictEqual(keys(query.range("ov", "age", 30, 40)), wantAge(30, 40));
  // a store-sourced index (position output) over the same rows, ordered view on: the reference's result ORDER is rebuilt on the host from any order
  query._opts.orderedIndexes = 1;                               // what attach(bullet, {orderedIndexes: 1}) sets: the default of every index built from here on
  query.index("ov", "score");
  const want = []; for (let i = 0; i < 2000; i++) { const n = b.store.ov["n" + i]; if (n && n.score >= 2900 && n.score <= 2950) want.push("ov/n" + i); }
  assert.deepStrictEqual(query.range("ov", "score", 2900, 2950).map((n) => n.path).sort(), want.sort());
  assert.strictEqual(crt.graph.indexOrderedInfo(query.indices["ov:score"].deviceField).afterQueries, 2);
  b.close();
  checks += 20;
}

/* write-through: what the host path decides (local puts, deletions) reaches the device before the next batch is resolved — node level */
{
  const b = new MiniBullet("w");
  const { crt, query } = attach(b, { capacityRows: 4096 });
  crt.mergeEntries([{ path: "wt/k1", data: { age: 10, hits: 1 }, vectorClock: { w: 4 } }], { apply: true });
  assert.deepStrictEqual(b.meta["wt/k1"].vectorClock, { w: 2 });                   // first write of an absent node: clock 2 (src/bullet-crt.js:172-185)
  b.setData("wt/k1/age", 50);                                                     // local put of a LEAF: its own path gets its own clock ({w:3}: a local first write increments twice, src/bullet-crt.js:358 + :172-185), the node's stays
  assert.deepStrictEqual(b.meta["wt/k1/age"].vectorClock, { w: 3 });
  assert.deepStrictEqual(b.meta["wt/k1"].vectorClock, { w: 2 });
  let r = crt.mergeEntries([{ path: "wt/k1", data: { age: 45 }, vectorClock: { w: 1 } }], { apply: true });
  assert.strictEqual(r.nApplied, 0);                                              // older than the node's clock: historical
  assert.strictEqual(b.store.wt.k1.age, 50);
  r = crt.mergeEntries([{ path: "wt/k1", data: { age: 40 }, vectorClock: { w: 2 } }], { apply: true });
  assert.strictEqual(r.nApplied, 1);                                              // a tie on the node's clock: the incoming object wins and replaces the node
  assert.deepStrictEqual(b.store.wt.k1, { age: 40 });
  b.setData("wt/k2", { age: 7 });                                                 // a node the device has never seen, written locally: clock {w:3}
  r = crt.mergeEntries([{ path: "wt/k2", data: { age: 9 }, vectorClock: { w: 2 } }], { apply: true });
  assert.strictEqual(r.nApplied, 0);                                              // not a first sight any more: 2 is older than the local write's clock
  r = crt.mergeEntries([{ path: "wt/k2", data: { age: 9, hits: 3 }, vectorClock: { w: 3 } }], { apply: true });
  assert.strictEqual(r.nApplied, 1);
  b.setData("wt/k1", null);                                                       // deletion (what a `deleted` sync entry becomes): the clock goes to 3
  assert.deepStrictEqual(b.meta["wt/k1"].vectorClock, { w: 3 });
  query.index("wt", "age", { source: "device" });
  assert.deepStrictEqual(query.range("wt", "age", 0, 100).map((n) => n.path), ["wt/k2"]);   // the deleted node left the device-side index
  r = crt.mergeEntries([{ path: "wt/k1", data: { age: 41 }, vectorClock: { w: 2 } }, { path: "wt/k1", data: { age: 42 }, vectorClock: { w: 3 } }], { apply: true });
  assert.deepStrictEqual(r.applied, [{ entry: 1, field: null }]);                 // 2 is older than the delete's clock, 3 ties with it and wins
  assert.deepStrictEqual(query.range("wt", "age", 0, 100).map((n) => n.path), ["wt/k1", "wt/k2"]);
  const row = crt.checkpoint().find((x) => x.path === "wt/k2" && x.field === "hits");
  assert.strictEqual(row.val, 3);
  b.close();
  checks += 14;
}

/* Q4: writes under an indexed path patch the index (dirty children -> device rows -> the device's change log) and every query equals what a
 * FRESH GpuQuery builds from the same store; the cases where only a rebuild is safe fall back to it and still agree */
{
  const GpuQuery = require("../gpu-query");
  const b = new MiniBullet("w");
  const { crt, query } = attach(b, { capacityRows: 1 << 16 });
  for (let i = 0; i < 400; i++) b.setData("users/u" + i, { age: i % 40, score: i * 3, name: "n" + i });
  query.index("users", "age"); query.index("users", "score");
  const paths = (nodes) => nodes.map((n) => n.path);
  const sameAsFresh = (tag) => {
    const b2 = new MiniBullet("w"); b2.store = b.store; b2.crt = { handleUpdate() { throw new Error("read-only twin"); } };
    const q2 = new GpuQuery(b2, { capacityRows: 1 << 16 });
    for (const [f, v] of [["age", 39], ["age", 7], ["score", 15], ["age", 1000]]) {
      assert.deepStrictEqual(paths(query.equals("users", f, v)), paths(q2.equals("users", f, v)), tag + " equals " + f + "=" + v);
      assert.strictEqual(query.count("users", f, v), q2.count("users", f, v), tag + " count " + f + "=" + v);
    }
    assert.deepStrictEqual(paths(query.range("users", "age", 0, 10)), paths(q2.range("users", "age", 0, 10)), tag + " range age");
    assert.deepStrictEqual(paths(query.range("users", "score", 100, 900)), paths(q2.range("users", "score", 100, 900)), tag + " range score");
    if (query.indices["users:age"].kind === "device" && query.indices["users:score"].kind === "device") {
      const t = [{ field: "age", min: 5, max: 20 }, { field: "score", min: 0, max: 600 }];
      assert.deepStrictEqual(paths(query.filterWhere("users", t)), paths(q2.filterWhere("users", t)), tag + " filterWhere");
    }
    q2.close();
    checks += 12;
  };
  sameAsFresh("fresh");
  const s0 = Object.assign({}, query.stats), d0 = crt.graph.indexRefreshCounts();
  b.setData("users/u5", { age: 39, score: 15, name: "n5" });                    // an existing child changes both indexed fields
  b.setData("users/u9/age", 7);                                                 // a leaf write below a child
  b.setData("users/zed", { age: 7, score: 15 });                                // new children, in creation order
  b.setData("users/yan", { age: 39, score: 450 });
  sameAsFresh("patched once");
  b.setData("users/zed/score", 451); b.setData("users/u5/age", 1000);
  crt.mergeEntries([{ path: "users/u1", data: { age: 39, score: 3 }, vectorClock: { w: 999999 } }, { path: "users/batchnew", data: { age: 7, score: 200 }, vectorClock: { w: 5 } }], { apply: true });
  sameAsFresh("patched twice, one of them through a batch");
  assert.strictEqual(query.stats.builds, s0.builds, "patched indexes were rebuilt from the store");
  assert.ok(query.stats.patches >= s0.patches + 4);
  const d1 = crt.graph.indexRefreshCounts();
  assert.strictEqual(d1.fullBuilds, d0.fullBuilds, "the device rebuilt a maintained index");
  assert.ok(d1.incremental > d0.incremental);
  // where only a rebuild gives the reference's state
  b.setData("users/123", { age: 7, score: 1 });                                 // integer-like key: enumerated first by a fresh scan
  sameAsFresh("integer-like key");
  b.setData("users/u3", null);                                                  // a child disappears
  sameAsFresh("deleted child");
  b.setData("users/nofield", { name: "x" }); sameAsFresh("child without the field");
  b.setData("users/nofield/age", 7); b.setData("users/later", { age: 7, score: 2 });
  sameAsFresh("existing child gains the field");
  b.setData("users/u7/age", "old");                                             // leaves the integer domain: the index becomes a host index
  sameAsFresh("non-integer value");
  assert.ok(query.stats.builds > s0.builds);
  b.close();
  checks += 6;
}

/* N4: K-writer vector clocks. (a) the reference's golden vectors through the addon's vc* entry points */
for (const [name, shards] of [["g6_vc_unique_2k.json", 1], ["g6_vc_dups_500.json", 1], ["g6_vc_empty_start.json", 1],
                              ["g6_vc_unique_2k.json", 3], ["g6_vc_dups_500.json", 4], ["g6_vc_empty_start.json", 2]]) {   // > 1: rows split over that many tables by owner
  const g = load(name);
  const { DeviceVcTable } = require("../device-graph");
  const t = new DeviceVcTable(g.writers, "w", { capacityRows: 1024, shards });       // small on purpose: the tables grow
  const K = g.writers.length;
  const cols = (rows) => {
    const c = new hash.VcColumns(rows.length, K);
    rows.forEach((r, i) => { const id = gen.splitmix64(BigInt(r[0]) + 1n); c.set(i, [Number(id & 0xffffffffn), Number(id >> 32n)], gen.rowField(0, 1), r[1], r[2]); });
    return c;
  };
  if (g.resident.length) t.loadRows(cols(g.resident));
  const r = t.mergeBatch(cols(g.deltas));
  assert.deepStrictEqual(Buffer.from(r.flags).toString("base64"), g.flags_b64, name + " flags");
  assert.deepStrictEqual(Array.from(r.updated), g.updated, name + " updated");
  assert.strictEqual(r.nRows, g.final_rows.length, name + " rows");
  const fin = cols(g.final_rows.map((x) => [x[0], x[1], x[2]]));
  const got = t.getRows(fin.id, fin.field);
  g.final_rows.forEach((x, i) => {
    assert.deepStrictEqual(Array.from(got.clocks.subarray(i * K, i * K + K)), x[1], name + " clock of row " + x[0]);
    assert.strictEqual(Number(got.val[i]), x[2]);
    assert.strictEqual(got.state[i], x[3] === 1 ? t.native.VC_SPARSE : t.native.VC_DENSE);
  });
  // range scan over the K-writer rows == the reference's final rows filtered on the host
  const vals = g.final_rows.map((x) => x[2]).sort((a, b2) => a - b2);
  const lo = vals[Math.floor(vals.length / 4)], hi = vals[Math.floor(3 * vals.length / 4)];
  const ids = t.scanRange(gen.rowField(0, 1), lo, hi);
  const wantIds = g.final_rows.filter((x) => x[2] >= lo && x[2] <= hi).map((x) => gen.splitmix64(BigInt(x[0]) + 1n)).sort((a, b2) => (a < b2 ? -1 : a > b2 ? 1 : 0));
  assert.deepStrictEqual(Array.from(ids).sort((a, b2) => (a < b2 ? -1 : a > b2 ? 1 : 0)), wantIds, name + " range scan");
  t.close();
  checks += 4 + g.final_rows.length;
}

/* N4 (a'): clocks over ordered SUBSETS of the writers (g11, made by the real reference): the key set travels with every clock, the stored clock's key
 * order after merges is the reference's */
for (const [name, shards] of [["g11_vc_keysets_2k.json", 1], ["g11_vc_keysets_hot.json", 1], ["g11_vc_keysets_2k.json", 3]]) {
  const g = load(name);
  const { DeviceVcTable } = require("../device-graph");
  const t = new DeviceVcTable(g.writers, "w", { capacityRows: 1024, shards });
  const K = g.writers.length;
  const ksOf = (keys) => { let ks = 0xffffffff; keys.forEach((w, i) => { ks = ((ks & ~(0xf << (4 * i))) | (w << (4 * i))) >>> 0; }); return ks; };
  const cols = (rows) => {
    const c = new hash.VcColumns(rows.length, K);
    rows.forEach((r, i) => { const id = gen.splitmix64(BigInt(r[0]) + 1n); c.set(i, [Number(id & 0xffffffffn), Number(id >> 32n)], gen.rowField(0, 1), r[2], r[3], ksOf(r[1])); });
    return c;
  };
  if (g.resident.length) t.loadRows(cols(g.resident));
  const r = t.mergeBatch(cols(g.deltas));
  assert.deepStrictEqual(Buffer.from(r.flags).toString("base64"), g.flags_b64, name + " flags");
  assert.deepStrictEqual(Array.from(r.updated), g.updated, name + " updated");
  assert.strictEqual(r.nRows, g.final_rows.length, name + " rows");
  const fin = cols(g.final_rows);
  const got = t.getRows(fin.id, fin.field);
  g.final_rows.forEach((x, i) => {
    assert.deepStrictEqual(Array.from(got.clocks.subarray(i * K, i * K + K)), x[2], name + " counters of row " + x[0]);
    assert.deepStrictEqual(hash.keysetWriters(got.keysets[i]), x[1], name + " key order of row " + x[0]);
    assert.strictEqual(Number(got.val[i]), x[3]);
  });
  t.close();
  checks += 3 + g.final_rows.length;
}

/* N4 (b): NODE-level semantics under multi-writer clocks, pinned on tests/golden/g12_vc_node_semantics.json (the reference's own sync loop): the
 * facade with GpuCRT({writers}) + the batch adapter — every object entry whose clock names only the table's writers is one delta on the node's clock
 * row of the vector-clock table; dominated / dominating / identical / concurrent are decided on the GPU, the host replays replace / mergeValues over
 * the store. Store and every path's clock — as an ORDERED key list — after every chunk, then the queries (store-sourced and device-sourced indexes). */
{
  const g = load("g12_vc_node_semantics.json");
  const b = new MiniBullet(g.id);
  const { crt, query, sync } = attach(b, { capacityRows: 4096, writers: g.writers, batchSync: {} });
  g.chunks.forEach((chunk, ci) => {
    sync.processSyncEntries(JSON.parse(JSON.stringify(chunk)), "peer-1");
    const want = g.after[ci];
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), want.store, "g12 store after chunk " + (ci + 1));
    assert.deepStrictEqual(Object.keys(b.meta).sort(), Object.keys(want.meta).sort(), "g12 paths with a clock after chunk " + (ci + 1));
    for (const p of Object.keys(want.meta)) {
      assert.deepStrictEqual(Object.keys(b.meta[p].vectorClock).map((w) => [w, b.meta[p].vectorClock[w]]), want.meta[p].clock, "g12 clock of " + p + " after chunk " + (ci + 1));
      assert.strictEqual(b.meta[p].source, want.meta[p].source, "g12 source of " + p);
      checks++;
    }
    // the device holds the same clocks (key order included)
    const ps = Object.keys(want.meta);
    crt.vcLookup(ps).forEach((c, i) => assert.deepStrictEqual(c && Object.keys(c).map((w) => [w, c[w]]), want.meta[ps[i]].clock, "g12 device clock of " + ps[i] + " after chunk " + (ci + 1)));
  });
  assert.ok(sync.stats.deviceEntries === 32 && sync.stats.hostEntries === 4, JSON.stringify(sync.stats));
  const G2 = require("../gpu-query");
  const dq = new G2({ id: "w", _getData: (p) => b._getData(p), setData() {}, get: (p) => b.get(p) }, { graph: crt.graph });
  for (const q of g.queries) {
    if (q.op === "count") {
      assert.strictEqual(query.count(q.path, q.field, q.args[0]), q.count, "g12 count");
      dq.index(q.path, q.field, { source: "device" });
      assert.strictEqual(dq.count(q.path, q.field, q.args[0]), q.count, "g12 count (device rows)");
    } else {
      const got = q.op === "range" ? query.range(q.path, q.field, q.args[0], q.args[1]) : query.equals(q.path, q.field, q.args[0]);
      assert.deepStrictEqual(got.map((n) => n.path), q.paths, "g12 " + q.op + " " + q.field + " (store-sourced index, reference order)");
      dq.index(q.path, q.field, { source: "device" });
      const dev = q.op === "range" ? dq.range(q.path, q.field, q.args[0], q.args[1]) : dq.equals(q.path, q.field, q.args[0]);
      assert.deepStrictEqual(dev.map((n) => n.path).sort(), q.paths.slice().sort(), "g12 " + q.op + " " + q.field + " (device-sourced index)");
    }
    checks += 2;
  }
  b.close();
}

/* N4 (c): seeded stress against the host resolver applied entry by entry (the reference's loop body; the host resolver is pinned on the reference by
 * host_semantics.js): 3 chunks of 1500 entries on 300 nodes — objects under clocks over random ordered subsets of {a, b, w}, now and then a writer the
 * table does not know, a primitive (a local write in the reference's loop) or a deletion, all of which take the host path and are mirrored to the
 * device. The final store and every clock (key order included) must agree after each chunk, and so must the clocks the device holds. */
{
  const WR = ["a", "b", "w"];
  const b = new MiniBullet("w");
  const { crt, sync } = attach(b, { writers: WR, capacityRows: 256, batchSync: {} });
  const twinB = new MiniBullet("w");
  twinB.crt = new GpuCRT(twinB);                                        // host resolver only
  const rng = gen.xorshift32(99 + STRESS_SALT);
  const randClock = () => {
    const order = [0, 1, 2];
    for (let i = 2; i > 0; i--) { const j = rng() % (i + 1); const t = order[i]; order[i] = order[j]; order[j] = t; }
    const c = {};
    for (const k of order.slice(0, rng() % 4)) c[WR[k]] = rng() % 4;
    return c;
  };
  const ordered = (c) => Object.keys(c).map((w) => [w, c[w]]);
  for (let round = 0; round < 3; round++) {
    const entries = [];
    for (let j = 0; j < 1500; j++) {
      const path = "vc/n" + (rng() % 300);
      const u = rng() % 100;
      if (u < 2) entries.push({ path, data: { hits: 1 }, vectorClock: { a: rng() % 3, zed: 2 } });
      else if (u < 4) entries.push({ path, data: rng() % 7, vectorClock: randClock() });
      else if (u < 5) entries.push({ path, deleted: true, vectorClock: randClock() });
      else if (u < 6) entries.push({ path, data: rng() % 2 ? [1, "a"] : (rng() % 2 ? "" : {}), vectorClock: randClock() });   // array (spread by the loop), empty string, {}
      else if (u < 8) entries.push({ path, data: { hits: rng() % 3, deep: { a: rng() % 2 }, none: null }, vectorClock: randClock() });
      else entries.push({ path, data: rng() % 3 ? { hits: (rng() % 5) - 2, level: rng() % 3 } : { hits: (rng() % 5) - 2, tag: "t" + (rng() % 3) }, vectorClock: randClock() });
    }
    for (const e of JSON.parse(JSON.stringify(entries))) {             // the loop body of src/bullet-network-sync.js:552-568
      if (e.deleted) twinB.setData(e.path, null, false);
      else twinB.setData(e.path, typeof e.data === "object" && e.data !== null ? Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }) : e.data, false);
    }
    sync.processSyncEntries(JSON.parse(JSON.stringify(entries)), "peer-1");
    for (const k of Object.keys(twinB.store.vc)) assert.deepStrictEqual(b.store.vc[k], twinB.store.vc[k], "vector mode: node vc/" + k + " after chunk " + round + ", clock " + JSON.stringify(twinB.meta["vc/" + k].vectorClock) + " vs " + JSON.stringify((b.meta["vc/" + k] || {}).vectorClock));
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), JSON.parse(JSON.stringify(twinB.store)), "vector mode: store after chunk " + round);
    const ps = Object.keys(twinB.meta).sort();
    assert.deepStrictEqual(Object.keys(b.meta).sort(), ps);
    for (const p of ps) assert.deepStrictEqual(ordered(b.meta[p].vectorClock), ordered(twinB.meta[p].vectorClock), "vector mode: clock of " + p + " after chunk " + round);
    const held = ps.filter((p) => Object.keys(twinB.meta[p].vectorClock).every((w) => WR.includes(w)));
    crt.vcLookup(held).forEach((c, i) => assert.deepStrictEqual(c && ordered(c), ordered(twinB.meta[held[i]].vectorClock), "vector mode: device clock of " + held[i]));
    checks += 2 * ps.length;
  }
  assert.ok(sync.stats.deviceEntries > 3500 && sync.stats.hostEntries > 100, JSON.stringify(sync.stats));
  assert.strictEqual(crt.vcLookup(["vc/none"])[0], null);
  assert.throws(() => new GpuCRT(new MiniBullet("z"), { writers: WR }).vcTable, (e) => e.code === "BMX_BAD_WRITERS");
  assert.throws(() => new GpuCRT(new MiniBullet("7"), { writers: ["7", "a"] }).vcTable, (e) => e.code === "BMX_BAD_WRITERS");   // integer-like ids reorder object keys
  checks += 3;
  b.close();
}

/* N4 (d): the same differential test with NOBODY named up front (writers: "auto"): the table is eight writers wide, this peer holds the first component and
 * every other writer takes a free one when a clock first names it. Ten writers send: the eight seen first are the device's, clocks naming the ninth and
 * tenth keep their paths on the host (hostOnlyInfo), and store, clocks (key order included) and the device's clock rows equal the host resolver's. */
{
  const POOL = ["w", "p1", "p2", "p3", "p4", "p5", "p6", "p7", "p8", "p9"];
  const b = new MiniBullet("w");
  const { crt, sync } = attach(b, { writers: "auto", capacityRows: 256, batchSync: {} });
  const twinB = new MiniBullet("w");
  twinB.crt = new GpuCRT(twinB);
  const rng = gen.xorshift32(4242 + STRESS_SALT);
  const ordered = (c) => Object.keys(c).map((w) => [w, c[w]]);
  for (let round = 0; round < 3; round++) {
    const live = Math.min(POOL.length, 4 + 3 * round);                 // the mesh grows: 4, 7, 10 writers are heard of
    const randClock = () => { const c = {}; const n = rng() % 4; for (let x = 0; x < n; x++) c[POOL[rng() % live]] = rng() % 4; return c; };
    const entries = [];
    for (let j = 0; j < 1200; j++) {
      const path = "au/n" + (rng() % 200);
      const u = rng() % 100;
      if (u < 3) entries.push({ path, data: rng() % 7, vectorClock: randClock() });
      else if (u < 4) entries.push({ path, deleted: true, vectorClock: randClock() });
      else entries.push({ path, data: rng() % 3 ? { hits: (rng() % 5) - 2, level: rng() % 3 } : { hits: (rng() % 5) - 2, tag: "t" + (rng() % 3) }, vectorClock: randClock() });
    }
    for (const e of JSON.parse(JSON.stringify(entries))) {
      if (e.deleted) twinB.setData(e.path, null, false);
      else twinB.setData(e.path, typeof e.data === "object" && e.data !== null ? Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }) : e.data, false);
    }
    sync.processSyncEntries(JSON.parse(JSON.stringify(entries)), "peer-1");
    assert.deepStrictEqual(JSON.parse(JSON.stringify(b.store)), JSON.parse(JSON.stringify(twinB.store)), "auto writers: store after chunk " + round);
    const ps = Object.keys(twinB.meta).sort();
    assert.deepStrictEqual(Object.keys(b.meta).sort(), ps);
    for (const p of ps) assert.deepStrictEqual(ordered(b.meta[p].vectorClock), ordered(twinB.meta[p].vectorClock), "auto writers: clock of " + p + " after chunk " + round);
    const known = crt.hostOnlyInfo().writers;
    const held = ps.filter((p) => Object.keys(twinB.meta[p].vectorClock).every((w) => known.includes(w)) && !crt._hostOnly.has(p));
    crt.vcLookup(held).forEach((c, i) => assert.deepStrictEqual(c && ordered(c), ordered(twinB.meta[held[i]].vectorClock), "auto writers: device clock of " + held[i]));
    checks += 2 * ps.length;
  }
  const info = crt.hostOnlyInfo();
  assert.strictEqual(info.writerSlots, 8);
  assert.strictEqual(info.writers.length, 8, JSON.stringify(info));
  assert.strictEqual(info.writers[0], "w");
  assert.ok(info.hostOnlyPaths > 0 && info.marked >= info.hostOnlyPaths, JSON.stringify(info));   // the ninth and tenth writer
  assert.ok(sync.stats.deviceEntries > 2000 && sync.stats.hostEntries > 100, JSON.stringify(sync.stats));
  assert.throws(() => new GpuCRT(new MiniBullet("w"), { writers: ["w"], maxWriters: 9 }).vcTable, (e) => e.code === "BMX_BAD_WRITERS");
  checks += 5;
  b.close();
}

/* N2 (stress): the same differential test in scalar mode — batch adapter on the GPU against the host resolver applied entry by entry: 3 chunks of 2000 entries
 * on 250 nodes under clocks {w: 0..7} (many ties), objects with integer and string fields, integer and string primitives (local writes in the reference's
 * loop: refused ones still move the clock), deletions, two-writer clocks (host-only paths), with and without put batching. Store, clocks, sources and the device's
 * clock rows after every chunk. */
for (const [batchPuts, lazyStore] of [[false, false], [true, false], [false, { idleSlice: 0 }], [true, true]]) {
  const b = new MiniBullet("w");
  const { crt, sync } = attach(b, { capacityRows: 256, batchSync: { batchPuts, lazyStore } });
  const twinB = new MiniBullet("w");
  twinB.crt = new GpuCRT(twinB);
  const rng = gen.xorshift32((batchPuts ? 4242 : 777) + (lazyStore ? 31 : 0) + STRESS_SALT);
  for (let round = 0; round < 3; round++) {
    const entries = [];
    for (let j = 0; j < 2000; j++) {
      const path = "st/n" + (rng() % 250);
      const u = rng() % 100, clock = { w: rng() % 8 };
      if (u < 3) entries.push({ path, data: { hits: 1 }, vectorClock: { w: rng() % 8, q: 1 } });
      else if (u < 8) entries.push({ path, data: rng() % 7, vectorClock: clock });                      // 0 among them: a falsy value _getData turns into {}
      else if (u < 10) entries.push({ path, data: rng() % 4 ? "s" + (rng() % 3) : "", vectorClock: clock });
      else if (u < 13) entries.push({ path, deleted: true, vectorClock: clock });
      else if (u < 14) entries.push({ path, data: [rng() % 3, "a"], vectorClock: clock });              // an array: the loop spreads it into an object (host path)
      else if (u < 15) entries.push({ path, data: {}, vectorClock: clock });
      else if (u < 17) entries.push({ path, data: { hits: rng() % 3, deep: { a: rng() % 2, b: [1] }, flag: rng() % 2 === 0, none: null }, vectorClock: clock });
      else entries.push({ path, data: rng() % 3 ? { hits: (rng() % 5) - 2, level: rng() % 3 } : { hits: (rng() % 5) - 2, tag: "t" + (rng() % 3) }, vectorClock: clock });
    }
    for (const e of JSON.parse(JSON.stringify(entries))) {             // the loop body of src/bullet-network-sync.js:552-568
      if (e.deleted) twinB.setData(e.path, null, false);
      else twinB.setData(e.path, typeof e.data === "object" && e.data !== null ? Object.assign({}, e.data, { __fromNetwork: true, __vectorClock: e.vectorClock }) : e.data, false);
    }
    if (batchPuts && round === 1) {                                     // the middle chunk as network puts (objects) and single-entry chunks (the rest)
      for (const e of JSON.parse(JSON.stringify(entries))) {
        if (e.deleted || typeof e.data !== "object") sync.processSyncEntries([e], "peer-1");
        else sync.handlePut("peer-1", { path: e.path, data: Object.assign({}, e.data, { __vectorClock: e.vectorClock }) });
      }
      sync.flush();
    } else sync.processSyncEntries(JSON.parse(JSON.stringify(entries)), "peer-1");
    for (const k of Object.keys(twinB.store.st)) assert.deepStrictEqual(b.store.st[k], twinB.store.st[k], "scalar stress: node st/" + k + " after chunk " + round + ", clock " + JSON.stringify(twinB.meta["st/" + k].vectorClock) + " vs " + JSON.stringify((b.meta["st/" + k] || {}).vectorClock));
    const ps = Object.keys(twinB.meta).sort();
    assert.deepStrictEqual(Object.keys(b.meta).sort(), ps);
    for (const p of ps) {
      assert.deepStrictEqual(b.meta[p].vectorClock, twinB.meta[p].vectorClock, "scalar stress: clock of " + p + " after chunk " + round);
      assert.strictEqual(b.meta[p].source, twinB.meta[p].source, "scalar stress: source of " + p + " after chunk " + round);
    }
    // the device's clock rows: ts = the stored clock's w for every path whose clock is {w: n}
    const held = ps.filter((p) => hash.scalarClock(twinB.meta[p].vectorClock, "w") >= 0);
    const ids = new BigUint64Array(held.length), fields = new Uint32Array(held.length);
    held.forEach((p, i) => { const id = crt.graph.keys.idOf(p); ids[i] = BigInt(id[0]) | (BigInt(id[1]) << 32n); fields[i] = crt.graph.keys.fieldOf("st", hash.NODE_CLOCK); });
    const rows = crt.graph.getRows(ids, fields);
    held.forEach((p, i) => { assert.ok(rows.found[i], "scalar stress: no clock row for " + p); assert.strictEqual(Number(rows.ts[i]), twinB.meta[p].vectorClock.w, "scalar stress: device clock of " + p + " after chunk " + round); });
    checks += 3 * ps.length;
  }
  assert.ok(sync.stats.deviceEntries > 3000, JSON.stringify(sync.stats));
  b.close();
}

/* lazy-store.js (opt-in, VERDICT r4 item 5): the winners of device batches are recorded and the facade's store / meta / op log follow when they are read. Against a
 * twin that applies the same chunks eagerly: nothing is folded while only batches arrive; ANY read through the facade (store, meta, log, _getData, a node's value(), a
 * host-path write, a device scan) sees exactly the eager state, op-log order and per-batch timestamps included; a registered listener switches the deferral off. */
{
  const mk = (lazyStore) => { const b = new MiniBullet("w"); const h = attach(b, { capacityRows: 8192, batchSync: { lazyStore } }); return { b, h }; };
  const L = mk({ idleSlice: 0 }), E = mk(false);
  const chunkOf = (c, n, base) => { const es = []; for (let i = 0; i < n; i++) es.push({ path: "lz/n" + ((i * 7 + base) % 900), data: { v: c * 1000 + i, tag: "t" + (i % 3) }, vectorClock: { w: 10 + c } }); return es; };
  for (let c = 0; c < 4; c++) for (const x of [L, E]) x.h.sync.processSyncEntries(chunkOf(c, 700, c * 13), "peer");
  assert.ok(L.h.lazyStore.n > 0 && L.h.lazyStore.folds === 0, "four chunks in, nothing folded yet: " + L.h.lazyStore.n);
  const pend = L.h.lazyStore.n;
  assert.strictEqual(Object.keys(L.h.lazyStore.real.store).length, 0);                         // the real store object has not been touched
  assert.deepStrictEqual(JSON.parse(JSON.stringify(L.b.store)), JSON.parse(JSON.stringify(E.b.store)));   // reading it folds everything
  assert.ok(L.h.lazyStore.n === 0 && L.h.lazyStore.folded === pend);
  assert.deepStrictEqual(Object.keys(L.b.meta).sort(), Object.keys(E.b.meta).sort());
  for (const p of Object.keys(E.b.meta)) { assert.deepStrictEqual(L.b.meta[p].vectorClock, E.b.meta[p].vectorClock, p); assert.strictEqual(L.b.meta[p].source, E.b.meta[p].source); }
  assert.deepStrictEqual(L.b.log.map((r) => [r.path, r.data.v]), E.b.log.map((r) => [r.path, r.data.v]));                    // the last 1000 operations, in arrival order
  // one more chunk, then different kinds of readers, each on a fresh pending set
  const readers = [
    (x) => x.b._getData("lz/n5"), (x) => x.b.get("lz/n6").value(), (x) => x.b.meta["lz/n7"].vectorClock, (x) => x.b.log.length,
    (x) => { x.b.setData("lz/n8", { v: -1, __fromNetwork: true, __vectorClock: { w: 500 } }); return x.b.store.lz.n8; },        // a host-path write lands BEHIND the recorded batch
    (x) => { x.h.query.index("lz", "v", { source: "device" }); return x.h.query.range("lz", "v", 4000, 4100).map((n) => n.path).sort(); },   // a device scan: the winners' value rows are there
    (x) => x.h.crt.getVectorClock("lz/n9"),
  ];
  readers.forEach((read, ri) => {
    for (const x of [L, E]) x.h.sync.processSyncEntries(chunkOf(4 + ri, 300, ri * 31), "peer");
    assert.ok(L.h.lazyStore.n > 0, "reader " + ri + ": something pending");
    const gotL = read(L), gotE = read(E);
    assert.strictEqual(L.h.lazyStore.n, 0, "reader " + ri + " folded what was pending");
    assert.deepStrictEqual(JSON.parse(JSON.stringify(gotL === undefined ? null : gotL)), JSON.parse(JSON.stringify(gotE === undefined ? null : gotE)), "reader " + ri);
    assert.deepStrictEqual(JSON.parse(JSON.stringify(L.b.store)), JSON.parse(JSON.stringify(E.b.store)), "store after reader " + ri);
    checks += 2;
  });
  // a listener wants its callback AT the write: nothing is deferred while one is registered
  L.b.listeners = { "lz/n1": [() => {}] };
  L.h.sync.processSyncEntries(chunkOf(40, 50, 0), "peer"); E.h.sync.processSyncEntries(chunkOf(40, 50, 0), "peer");
  assert.strictEqual(L.h.lazyStore.n, 0);
  assert.deepStrictEqual(JSON.parse(JSON.stringify(L.h.lazyStore.real.store)), JSON.parse(JSON.stringify(E.b.store)));
  L.b.listeners = {};
  // the documented difference: a live object handed out before a batch shows the batch once the store has been read through the facade (or the idle fold ran)
  const held = L.b.get("lz").value();
  L.h.sync.processSyncEntries([{ path: "lz/brandnew", data: { v: 1 }, vectorClock: { w: 5 } }], "peer");
  assert.strictEqual(held.brandnew, undefined);
  assert.deepStrictEqual(L.b.store.lz.brandnew, { v: 1 });
  assert.deepStrictEqual(held.brandnew, { v: 1 });                                           // the same object: the fold wrote into it
  L.b.close(); E.b.close();
  assert.ok(Object.getOwnPropertyDescriptor(L.b, "store").writable, "close() gives the plain properties back");
  checks += 12;
}

/* Integer ENTRIES through a DIRECT GpuCRT.mergeEntries call, pinned on tests/golden/g13_entries_integer_ties.json (the reference's processUpdate over
 * the same entry lists): identical clocks are decided BY VALUE — larger wins, equal is a no-op, also against the {w:2} of a first sight and for
 * several entries of one path inside one chunk (src/bullet-crt.js:200-233). On the device an integer path's clock row carries the integer itself as
 * its value. Entries that meet the other kind of value on their path (`mixed`) come back in `host`, exactly those. ADVICE r3 #2: 5 then 3 under one clock. */
{
  const g = load("g13_entries_integer_ties.json");
  const b = new MiniBullet(g.id);
  const { crt } = attach(b, { capacityRows: 4096 });
  g.chunks.forEach((chunk, ci) => {
    const entries = JSON.parse(JSON.stringify(chunk));
    const r = crt.mergeEntries(entries, { apply: true });
    const mixed = []; chunk.forEach((e, j) => { if (e.mixed) mixed.push(j); });
    assert.deepStrictEqual(Array.from(r.host), mixed, "g13 chunk " + (ci + 1) + ": exactly the entries that meet the other kind of value are handed back");
    const want = g.after[ci];
    for (const p of Object.keys(want)) {
      if (p.startsWith("mix/")) continue;               // the host path's business (the caller resolves what comes back in `host`)
      const seg = p.split("/");
      assert.strictEqual(b.store[seg[0]][seg[1]], want[p].value, "g13 value of " + p + " after chunk " + (ci + 1));
      assert.deepStrictEqual(b.meta[p].vectorClock, want[p].clock, "g13 clock of " + p + " after chunk " + (ci + 1));
      checks++;
    }
    // the device rows: clock row (ts = the clock, val = the integer) and the value row the scans read
    const ps = Object.keys(want).filter((p) => !p.startsWith("mix/"));
    const snap = crt.checkpoint();
    for (const p of ps) {
      const clk = snap.find((x) => x.path === p && x.field === hash.NODE_CLOCK), val = snap.find((x) => x.path === p && x.field === null);
      assert.ok(clk && val, "g13 device rows of " + p);
      assert.deepStrictEqual([clk.ts, clk.val, val.val], [want[p].clock.w, want[p].value, want[p].value], "g13 device rows of " + p + " after chunk " + (ci + 1));
    }
  });
  // the advisor's case, on a fresh path and directly: the same path with data 5 then 3 under the same clock keeps 5; then 5 again is a no-op; 6 wins
  const r2 = crt.mergeEntries([{ path: "adv/x", data: 5, vectorClock: { w: 7 } }, { path: "adv/x", data: 5, vectorClock: { w: 2 } }, { path: "adv/x", data: 3, vectorClock: { w: 2 } }], { apply: true });
  assert.deepStrictEqual([b.store.adv.x, Array.from(r2.appliedEntries)], [5, [0]]);
  const r3 = crt.mergeEntries([{ path: "adv/x", data: 6, vectorClock: { w: 2 } }, { path: "adv/x", data: 0, vectorClock: { w: 9 } }], { apply: true });
  assert.deepStrictEqual([b.store.adv.x, Array.from(r3.appliedEntries), Array.from(r3.host)], [6, [0], [1]]);   // (a 0 is the host's: _getData turns a stored 0 into {})
  assert.strictEqual(crt.hostOnlyInfo().integerPaths >= 7, true);
  b.close();
  checks += 4;
}

/* Promise variant: two batches in flight from the event loop's point of view, serialised inside the addon */
(async () => {
  const g = load("g2_stream_hot30_10k_10k.json");
  const { resident, deltas, F } = gen.genStream(g.spec);
  const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} }, { capacityRows: 1024 });   // small on purpose: the table grows
  crt.graph.loadRows(columns(resident, F));
  const half = deltas.length >> 1;
  let ticks = 0;
  const timer = setInterval(() => { ticks++; }, 0);
  const p1 = crt.mergeBatchAsync(columns(deltas.slice(0, half), F));
  const p2 = crt.mergeBatchAsync(columns(deltas.slice(half), F));
  const [r1, r2] = await Promise.all([p1, p2]);
  clearInterval(timer);
  assert.strictEqual(r2.nRows, g.n_rows_final);
  const d = crt.graph.dumpRows();
  let digest = 0n;
  for (let i = 0; i < d.id.length; i++) digest = (digest + gen.rowDigest(d.id[i], d.field[i], d.ts[i], d.val[i])) & ((1n << 64n) - 1n);
  assert.strictEqual(digest.toString(16), g.digest, "two sequential async halves == the whole batch");
  await assert.rejects(crt.mergeBatchAsync({ id: new BigUint64Array([2n ** 64n - 1n]), field: new Uint32Array([1]), ts: new BigInt64Array([1n]), val: new BigInt64Array([1n]) }),
    (e) => e.code === -5);
  crt.close();
  checks += 3;

  /* the ingestion pipeline (mergeEntriesAsync, two chunks in flight: chunk b + 1 is packed while chunk b is on the GPU, winners are applied one
   * chunk late): same store, same clocks, same device rows as the synchronous loop over the same chunks — first sights, ties and replaced nodes
   * spread over chunk boundaries */
  {
    let s = 777;
    const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
    const chunks = [];
    for (let c = 0; c < 6; c++) {
      const entries = [];
      for (let j = 0; j < 3000; j++) {
        const data = rnd() % 3 === 0 ? { a: rnd() % 50 } : (rnd() % 2 ? { a: rnd() % 50, b: rnd() % 9 } : { b: rnd() % 9, c: 1 });
        entries.push({ path: "p/n" + (rnd() % 2500), data, vectorClock: { w: rnd() % 12 } });     // few clocks: ties everywhere, {w:2} included
      }
      chunks.push(entries);
    }
    const run = async (pipelined) => {
      const b = new MiniBullet("w");
      const { crt: c2, query } = attach(b, { capacityRows: 1 << 15 });
      if (pipelined) await c2.mergeEntriesPipelined(chunks.map((c) => JSON.parse(JSON.stringify(c))), { apply: true });
      else for (const c of chunks) c2.mergeEntries(JSON.parse(JSON.stringify(c)), { apply: true });
      query.index("p", "a", { source: "device" });
      const out = { store: JSON.parse(JSON.stringify(b.store)), clocks: Object.keys(b.meta).sort().map((k) => [k, b.meta[k].vectorClock.w]),
        scan: query.range("p", "a", 10, 30).map((n) => n.path), rows: c2.checkpoint().filter((x) => x.field !== hash.NODE_CLOCK).map((x) => [x.path, x.field, x.ts, x.val].join()).sort() };
      b.close();
      return out;
    };
    const seq = await run(false), pip = await run(true);
    assert.deepStrictEqual(pip.store, seq.store, "pipelined ingestion: store");
    assert.deepStrictEqual(pip.clocks, seq.clocks, "pipelined ingestion: clocks");
    assert.deepStrictEqual(pip.scan, seq.scan, "pipelined ingestion: device-side scan");
    assert.deepStrictEqual(pip.rows, seq.rows, "pipelined ingestion: device value rows");
    assert.ok(seq.scan.length > 100 && seq.clocks.length > 2000);
    checks += 5;
  }
  /* the same for GENERAL vector clocks (N4): mergeEntriesAsync under `writers` is a real asynchronous merge now (vcMergeBatchAsync: a worker thread of the
   * addon, the updated rows' clocks read right behind the merge) — two chunks in flight leave the store, every clock's key ORDER and the device rows that a
   * synchronous loop over the same chunks leaves; and a clock that names a writer outside the table shows up in hostOnlyInfo() */
  {
    let s = 4242;
    const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
    const WR = ["a", "b", "w"];
    const chunks = [];
    for (let c = 0; c < 5; c++) {
      const entries = [];
      for (let j = 0; j < 2500; j++) {
        const clock = {}; const k = 1 + (rnd() % 3);
        for (let x = 0; x < k; x++) clock[WR[(x + (rnd() % 3)) % 3]] = rnd() % 4;
        const data = rnd() % 2 ? { a: rnd() % 50, note: "n" + (rnd() % 3) } : { b: rnd() % 9 };
        entries.push({ path: "v/n" + (rnd() % 1500), data, vectorClock: clock });
      }
      chunks.push(entries);
    }
    const run = async (pipelined) => {
      const b = new MiniBullet("w");
      const { crt: c2 } = attach(b, { capacityRows: 1 << 15, writers: WR });
      let ticks = 0;
      const timer = setInterval(() => { ticks++; }, 0);
      if (pipelined) await c2.mergeEntriesPipelined(chunks.map((c) => JSON.parse(JSON.stringify(c))), { apply: true });
      else for (const c of chunks) c2.mergeEntries(JSON.parse(JSON.stringify(c)), { apply: true });
      clearInterval(timer);
      const paths = Object.keys(b.meta).sort();
      const out = { store: JSON.parse(JSON.stringify(b.store)), clocks: paths.map((k) => [k, Object.keys(b.meta[k].vectorClock).map((w) => [w, b.meta[k].vectorClock[w]])]),
        device: c2.vcLookup(paths).map((c) => c && Object.keys(c).map((w) => [w, c[w]])) };
      // a clock over a writer the table does not know: its entry is the host's, and the caller can see that such paths exist
      const before = c2.hostOnlyInfo().marked;
      const r = c2.mergeEntries([{ path: "v/foreign", data: { a: 1 }, vectorClock: { zed: 3 } }], { apply: true });
      assert.deepStrictEqual(Array.from(r.host), [0]);
      b.setData("v/foreign", { a: 1, __fromNetwork: true, __vectorClock: { zed: 3 } });      // first sight: stored under {w: 2}, the sender's clock is discarded (src/bullet-crt.js:172-185)
      b.setData("v/foreign", { a: 2, __fromNetwork: true, __vectorClock: { zed: 3 } });      // concurrent with {w: 2}: the merged clock names `zed`, which the table cannot hold
      assert.deepStrictEqual(Object.keys(b.meta["v/foreign"].vectorClock).sort(), ["w", "zed"]);
      assert.ok(c2.hostOnlyInfo().marked > before && c2.hostOnlyInfo().hostOnlyPaths >= 1, JSON.stringify(c2.hostOnlyInfo()));
      b.close();
      return out;
    };
    const seq = await run(false), pip = await run(true);
    assert.deepStrictEqual(pip.store, seq.store, "pipelined vector ingestion: store");
    assert.deepStrictEqual(pip.clocks, seq.clocks, "pipelined vector ingestion: clocks with their key order");
    assert.deepStrictEqual(pip.device, seq.device, "pipelined vector ingestion: clocks on the device");
    assert.ok(seq.clocks.length > 1000);
    checks += 4;
  }
  console.log("device_parity ok:", checks, "checks");
})().catch((e) => { console.error(e); process.exit(1); });
