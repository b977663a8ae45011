"use strict";
/*
 * cpu_baseline.js — the Node.js path timed on this box (bench.py's cpu_baseline.js_twin): the per-delta loop the reference
 * runs (processUpdate per entry over a Map of {value, clock}: BASELINE.md §2 harness shape), executed by GpuCRT's
 * single-operation path — a JS implementation pinned to the reference on tests/golden (host_semantics.js). One thread.
 * Usage: node cpu_baseline.js [R] [D]   -> one JSON line
 */
const { GpuCRT } = require("..");
const R = parseInt(process.argv[2] || "1000000", 10);
const D = parseInt(process.argv[3] || "300000", 10);
let s = 12345;
const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
const crt = new GpuCRT({ id: "w", meta: {}, _getData() {} });
const state = new Map();
for (let i = 0; i < R; i++) state.set(i + "/f", { value: (rnd() % 2001) - 1000, clock: { w: 1000000 + (rnd() % 1000000) } });
const keys = new Array(D), ts = new Array(D), val = new Array(D);
for (let j = 0; j < D; j++) {
  const ins = rnd() % 100 < 10;
  keys[j] = (ins ? R + j : rnd() % R) + "/f";
  ts[j] = 1000000 + (rnd() % 2000000); val[j] = (rnd() % 2001) - 1000;
}
let applied = 0;
const t0 = process.hrtime.bigint();
for (let j = 0; j < D; j++) {
  const cur = state.get(keys[j]);
  const r = crt.processUpdate(keys[j], val[j], { w: ts[j] }, cur ? cur.value : undefined, cur ? cur.clock : undefined);
  if (r.decision.incoming || !cur || r.decision.concurrent) { state.set(keys[j], { value: r.value, clock: r.vectorClock }); applied++; }
}
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ value: D / dt, unit: "merges/s", cores: 1, resident_keys: R, deltas: D, applied, seconds: dt, node: process.version }));
