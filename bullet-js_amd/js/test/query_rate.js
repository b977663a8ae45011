"use strict";
/*
 * query_rate.js — latency of indexed queries through the whole JS host (GpuQuery -> N-API -> device -> ids or positions -> paths -> nodes), one thread.
 * R nodes under one collection, an integer field of 1000 distinct values; equals (R/1000 matches), a narrow range (R/100 matches), count.
 * Usage: node query_rate.js [R] [ordered] [device|store]     (ordered: 0 = column scans, N >= 1 / "auto" = value-ordered view on the device)   -> one JSON line. Needs an MI355X.
 */
const { attach } = require("..");
const MiniBullet = require("./mini-bullet");
const R = parseInt(process.argv[2] || "1000000", 10);
const ordered = process.argv[3] === "auto" ? "auto" : parseInt(process.argv[3] || "0", 10);
const b = new MiniBullet("w");
const { crt, query } = attach(b, { capacityRows: 2 * R });
for (let r0 = 0; r0 < R; r0 += 250000) {
  const chunk = new Array(Math.min(250000, R - r0));
  for (let i = 0; i < chunk.length; i++) { const k = r0 + i; chunk[i] = { path: "u/n" + k, data: { age: (k * 7919) % 1000, hits: k & 7 }, vectorClock: { w: 5 } }; }
  crt.mergeEntries(chunk, { insertMode: "delta", apply: true });
}
const time = (label, fn, reps) => {
  for (let i = 0; i < 5; i++) fn(i);
  const t0 = process.hrtime.bigint();
  let m = 0;
  for (let i = 0; i < reps; i++) m += fn(i);
  return { us: Number(process.hrtime.bigint() - t0) / 1e3 / reps, matches: m / reps };
};
const out = { nodes: R, ordered };
const source = process.argv[4] === "store" ? "store" : "device";       // "device": the rows mergeEntries put on the GPU; "store": children read from the JS store and uploaded
{
  const t0 = process.hrtime.bigint();
  query.index("u", "age", { source, ordered });
  const build_ms = Number(process.hrtime.bigint() - t0) / 1e6;
  const o = { index_build_ms: Math.round(build_ms * 10) / 10 };
  o.count = time("count", (i) => query.count("u", "age", i % 1000), 200);
  o.equals = time("equals", (i) => query.equals("u", "age", (i * 37) % 1000).length, 200);
  o.range_1pct = time("range", (i) => query.range("u", "age", (i * 37) % 990, (i * 37) % 990 + 9).length, 50);
  out[source + "_sourced"] = o;
}
console.log(JSON.stringify(out));
b.close();
