"use strict";
/*
 * hash.js — host-side hashing of Bullet's string keys to the fixed-width device keys (SURVEY H5).
 *   node path  -> 64-bit id, kept as two uint32 halves [lo, hi] (no BigInt on the hot path)
 *   field name -> 32-bit field hash (scoped by the parent collection, like the reference's "path:field" index key
 *                 src/bullet-query.js:31)
 * The device never sees strings; KeyDictionary remembers id -> path and detects collisions.
 */

function fnv1a32(str, seed) {
  let h = seed >>> 0;
  for (let i = 0; i < str.length; i++) {
    let c = str.charCodeAt(i);
    if (c < 0x80) {
      h = Math.imul(h ^ c, 0x01000193);
    } else {                       // UTF-16 code unit as two bytes: deterministic, no encoder allocation
      h = Math.imul(h ^ (c & 0xff), 0x01000193);
      h = Math.imul(h ^ (c >>> 8), 0x01000193);
    }
  }
  return h >>> 0;
}
function fmix32(h) {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b);
  h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35);
  h ^= h >>> 16;
  return h >>> 0;
}

/* 64-bit id of a node path as [lo, hi]; 0xFFFFFFFF:FFFFFFFF is reserved by the device */
function pathId(p) {
  const lo = fmix32(fnv1a32(p, 0x811c9dc5));
  let hi = fmix32(fnv1a32(p, 0x9747b28c) ^ lo);
  if (lo === 0xffffffff && hi === 0xffffffff) hi = 0xfffffffe;
  return [lo, hi];
}

/* 32-bit hash of (collection path, field name); 0xFFFFFFFF is reserved by the device */
function fieldId(collection, field) {
  const h = fmix32(fnv1a32(collection + ":" + (field === null || field === undefined ? "" : field), 0x811c9dc5));
  return h === 0xffffffff ? 0xfffffffe : h;
}

function idKey(lo, hi) { return hi * 4294967296 + lo <= Number.MAX_SAFE_INTEGER ? String(hi * 4294967296 + lo) : hi.toString(16) + ":" + lo.toString(16); }

/*
 * path <-> 64-bit id, both directions, in ONE open-addressing table over typed arrays (slot = {lo, hi, index+1 into `paths`}).
 * A JS Map keyed by the path string spent 0.7 us per lookup of a freshly built string (V8 hashes the string again, then chases a
 * table of a million entries); here the id that has to be computed anyway IS the hash, so a lookup is two FNV passes over the
 * string, one probe and one string comparison (which is what detects a 64-bit collision between two different paths).
 * lookup(p) leaves the id in this.lo / this.hi (no array allocated per entry); idOf(p) returns [lo, hi] for the other callers.
 */
class KeyDictionary {
  constructor(capacity = 1 << 12) {
    this.fields = new Map();   // field hash -> [collection, field name]
    this._fieldCache = new Map();   // collection -> Map(field name -> hash)
    this.paths = [];           // index -> path
    this.lo = 0; this.hi = 0;
    this.cut = -1; this.ph1 = 0; this.ph2 = 0;   // of the last lookup(): position of the path's last "/" and the 64-bit hash state of its parent prefix
    this.idx = -1;                              // ... and the path's number in this dictionary (paths[idx])
    this._alloc(capacity);
  }
  get size() { return this.paths.length; }
  _alloc(cap) {
    this._cap = cap; this._mask = cap - 1;
    this._t = new Uint32Array(cap * 4);       // [lo, hi, index + 1, -] per slot: one cache line holds four slots
  }
  _grow() {
    const old = this._t, ocap = this._cap;
    this._alloc(ocap * 2);
    const t = this._t, mask = this._mask;
    for (let s = 0; s < ocap; s++) {
      const k = old[4 * s + 2];
      if (k === 0) continue;
      const lo = old[4 * s], hi = old[4 * s + 1];
      let d = (lo ^ Math.imul(hi, 0x9E3779B1)) & mask;
      while (t[4 * d + 2] !== 0) d = (d + 1) & mask;
      t[4 * d] = lo; t[4 * d + 1] = hi; t[4 * d + 2] = k; t[4 * d + 3] = old[4 * s + 3];
    }
  }
  /* id of path p -> this.lo / this.hi; registers the path on first sight. One pass over the string gives the two halves of the id (same
   * values as pathId()) and a third, independent 32-bit hash kept in the slot: a known path is recognised by all 96 bits without touching
   * its stored string (three dependent cache misses less per lookup); two paths with the same 64-bit id but different check words are
   * the collision the device cannot represent, and are refused. */
  lookup(p) {
    let h1 = 0x811c9dc5, h2 = 0x9747b28c, h3 = 0x2f0b4a27;
    let cut = -1, p1 = 0, p2 = 0;             // by-products for the callers that need the parent: index of the last "/", hash state of the prefix in front of it
    for (let i = 0; i < p.length; i++) {
      const c = p.charCodeAt(i);
      if (c < 0x80) {
        if (c === 47) { cut = i; p1 = h1; p2 = h2; }
        h1 = Math.imul(h1 ^ c, 0x01000193); h2 = Math.imul(h2 ^ c, 0x01000193); h3 = Math.imul(h3 ^ c, 0x01000193);
      } else {
        const a = c & 0xff, b = c >>> 8;
        h1 = Math.imul(Math.imul(h1 ^ a, 0x01000193) ^ b, 0x01000193);
        h2 = Math.imul(Math.imul(h2 ^ a, 0x01000193) ^ b, 0x01000193);
        h3 = Math.imul(Math.imul(h3 ^ a, 0x01000193) ^ b, 0x01000193);
      }
    }
    const lo = fmix32(h1 >>> 0);
    let hi = fmix32((h2 >>> 0) ^ lo);
    if (lo === 0xffffffff && hi === 0xffffffff) hi = 0xfffffffe;
    const chk = fmix32((h3 >>> 0) ^ p.length);
    this.lo = lo; this.hi = hi; this.cut = cut; this.ph1 = p1; this.ph2 = p2;
    const t = this._t, mask = this._mask;
    let s = (lo ^ Math.imul(hi, 0x9E3779B1)) & mask;
    for (;;) {
      const k = t[4 * s + 2];
      if (k === 0) break;
      if (t[4 * s] === lo && t[4 * s + 1] === hi) {
        if (t[4 * s + 3] === chk) { this.idx = k - 1; return; }
        const err = new Error(`bmx: 64-bit id collision between paths '${this.paths[k - 1]}' and '${p}'`);
        err.code = "BMX_ID_COLLISION";
        throw err;
      }
      s = (s + 1) & mask;
    }
    this.paths.push(p);
    this.idx = this.paths.length - 1;
    t[4 * s] = lo; t[4 * s + 1] = hi; t[4 * s + 2] = this.paths.length; t[4 * s + 3] = chk;
    if (this.paths.length * 2 > this._cap) this._grow();
  }
  /* lookup() in two halves for callers that resolve many paths at once (GpuCRT._packEntries): hashInto(p, i) does the arithmetic for slot i of a block
   * (no table access), probeBlock(paths, m) then resolves the m slots back to back — m independent table probes the CPU can overlap, where lookup() after
   * lookup() pays one cache miss at a time (the table of a million paths does not fit the caches). Results per slot: bLo/bHi (the id), bCut/bP1/bP2 (the
   * parent's by-products, as lookup() leaves them in cut/ph1/ph2), bIdx (the path's number). */
  _block(m) {
    if (!this.bLo || this.bLo.length < m) {
      this.bLo = new Uint32Array(m); this.bHi = new Uint32Array(m); this.bChk = new Uint32Array(m); this.bP1 = new Uint32Array(m); this.bP2 = new Uint32Array(m);
      this.bCut = new Int32Array(m); this.bIdx = new Int32Array(m);
    }
  }
  hashInto(p, i) {
    let h1 = 0x811c9dc5, h2 = 0x9747b28c, h3 = 0x2f0b4a27;
    let cut = -1, p1 = 0, p2 = 0;
    for (let x = 0; x < p.length; x++) {
      const c = p.charCodeAt(x);
      if (c < 0x80) {
        if (c === 47) { cut = x; p1 = h1; p2 = h2; }
        h1 = Math.imul(h1 ^ c, 0x01000193); h2 = Math.imul(h2 ^ c, 0x01000193); h3 = Math.imul(h3 ^ c, 0x01000193);
      } else {
        const a = c & 0xff, b = c >>> 8;
        h1 = Math.imul(Math.imul(h1 ^ a, 0x01000193) ^ b, 0x01000193);
        h2 = Math.imul(Math.imul(h2 ^ a, 0x01000193) ^ b, 0x01000193);
        h3 = Math.imul(Math.imul(h3 ^ a, 0x01000193) ^ b, 0x01000193);
      }
    }
    const lo = fmix32(h1 >>> 0);
    let hi = fmix32((h2 >>> 0) ^ lo);
    if (lo === 0xffffffff && hi === 0xffffffff) hi = 0xfffffffe;
    this.bLo[i] = lo; this.bHi[i] = hi; this.bChk[i] = fmix32((h3 >>> 0) ^ p.length); this.bCut[i] = cut; this.bP1[i] = p1; this.bP2[i] = p2;
  }
  probeBlock(paths, m) {
    const bLo = this.bLo, bHi = this.bHi, bChk = this.bChk, bIdx = this.bIdx;
    let t = this._t, mask = this._mask;
    for (let i = 0; i < m; i++) {
      const lo = bLo[i], hi = bHi[i];
      let s = (lo ^ Math.imul(hi, 0x9E3779B1)) & mask;
      for (;;) {
        const k = t[4 * s + 2];
        if (k === 0) {                                         // first sight: register the path (as lookup() does)
          const p = paths[i];
          this.paths.push(p);
          bIdx[i] = this.paths.length - 1;
          t[4 * s] = lo; t[4 * s + 1] = hi; t[4 * s + 2] = this.paths.length; t[4 * s + 3] = bChk[i];
          if (this.paths.length * 2 > this._cap) { this._grow(); t = this._t; mask = this._mask; }
          break;
        }
        if (t[4 * s] === lo && t[4 * s + 1] === hi) {
          if (t[4 * s + 3] === bChk[i]) { bIdx[i] = k - 1; break; }
          const err = new Error(`bmx: 64-bit id collision between paths '${this.paths[k - 1]}' and '${paths[i]}'`);
          err.code = "BMX_ID_COLLISION";
          throw err;
        }
        s = (s + 1) & mask;
      }
    }
  }
  /* the number of a KNOWN path (paths[number]), -1 for a path this dictionary has never seen; registers nothing */
  find(p) {
    const before = this.paths.length;
    // (the arithmetic of lookup(); a path that is not there is probed to its empty slot and left out)
    let h1 = 0x811c9dc5, h2 = 0x9747b28c, h3 = 0x2f0b4a27;
    for (let i = 0; i < p.length; i++) {
      const c = p.charCodeAt(i);
      if (c < 0x80) { h1 = Math.imul(h1 ^ c, 0x01000193); h2 = Math.imul(h2 ^ c, 0x01000193); h3 = Math.imul(h3 ^ c, 0x01000193); }
      else {
        const a = c & 0xff, b = c >>> 8;
        h1 = Math.imul(Math.imul(h1 ^ a, 0x01000193) ^ b, 0x01000193);
        h2 = Math.imul(Math.imul(h2 ^ a, 0x01000193) ^ b, 0x01000193);
        h3 = Math.imul(Math.imul(h3 ^ a, 0x01000193) ^ b, 0x01000193);
      }
    }
    const lo = fmix32(h1 >>> 0);
    let hi = fmix32((h2 >>> 0) ^ lo);
    if (lo === 0xffffffff && hi === 0xffffffff) hi = 0xfffffffe;
    const chk = fmix32((h3 >>> 0) ^ p.length);
    const t = this._t, mask = this._mask;
    let s = (lo ^ Math.imul(hi, 0x9E3779B1)) & mask;
    for (;;) {
      const k = t[4 * s + 2];
      if (k === 0) return -1;
      if (t[4 * s] === lo && t[4 * s + 1] === hi) return t[4 * s + 3] === chk && before === this.paths.length ? k - 1 : -1;
      s = (s + 1) & mask;
    }
  }
  idOf(p) { this.lookup(p); return [this.lo, this.hi]; }
  pathOf(lo, hi) {
    const t = this._t, mask = this._mask;
    let s = (lo ^ Math.imul(hi, 0x9E3779B1)) & mask;
    for (;;) {
      const k = t[4 * s + 2];
      if (k === 0) return undefined;
      if (t[4 * s] === lo && t[4 * s + 1] === hi) return this.paths[k - 1];
      s = (s + 1) & mask;
    }
  }
  fieldOf(collection, field) {
    // (collection, field) -> hash, cached: a sync chunk names the same few fields of the same few collections over and over
    const fkey = field === null || field === undefined ? "" : field;
    let per = this._fieldCache.get(collection);
    if (per === undefined) { per = new Map(); this._fieldCache.set(collection, per); }
    const cached = per.get(fkey);
    if (cached !== undefined) return cached;
    const h = this._fieldOfSlow(collection, field);
    per.set(fkey, h);
    return h;
  }
  _fieldOfSlow(collection, field) {
    const h = fieldId(collection, field);
    const known = this.fields.get(h);
    if (known === undefined) this.fields.set(h, [collection, field === undefined ? null : field]);
    else if (known[0] !== collection || known[1] !== (field === undefined ? null : field)) {
      const err = new Error(`bmx: 32-bit field hash collision between '${known[0]}:${known[1]}' and '${collection}:${field}'`);
      err.code = "BMX_FIELD_COLLISION";
      throw err;
    }
    return h;
  }
}

/* typed-column builder: id as BigUint64Array written through a Uint32Array view */
class Columns {
  /* backing: {id, field, ts, val} typed arrays of n rows to build in (page-locked ones from the addon's hostColumns(n)); own arrays otherwise */
  constructor(n, backing) {
    this.n = n;
    this.id = backing ? backing.id : new BigUint64Array(n);
    this.field = backing ? backing.field : new Uint32Array(n);
    this.ts = backing ? backing.ts : new BigInt64Array(n);
    this.val = backing ? backing.val : new BigInt64Array(n);
    this._id32 = new Uint32Array(this.id.buffer, this.id.byteOffset, 2 * n);
    this._ts32 = new Uint32Array(this.ts.buffer, this.ts.byteOffset, 2 * n);
    this._val32 = new Uint32Array(this.val.buffer, this.val.byteOffset, 2 * n);
  }
  /* ts and val are safe integers (|x| <= 2^53-1): written as two 32-bit halves, no BigInt allocated per element */
  set(i, idPair, field, ts, val) {
    this._id32[2 * i] = idPair[0]; this._id32[2 * i + 1] = idPair[1];
    this.field[i] = field;
    let hi = Math.floor(ts / 4294967296);
    this._ts32[2 * i] = ts - hi * 4294967296; this._ts32[2 * i + 1] = hi;
    hi = Math.floor(val / 4294967296);
    this._val32[2 * i] = val - hi * 4294967296; this._val32[2 * i + 1] = hi;      // a negative hi wraps to its two's complement in the Uint32Array
  }
  /* same as set() with the id as two numbers (KeyDictionary.lookup leaves it in .lo / .hi) */
  set2(i, lo, hi32, field, ts, val) {
    this._id32[2 * i] = lo; this._id32[2 * i + 1] = hi32;
    this.field[i] = field;
    let hi = Math.floor(ts / 4294967296);
    this._ts32[2 * i] = ts - hi * 4294967296; this._ts32[2 * i + 1] = hi;
    hi = Math.floor(val / 4294967296);
    this._val32[2 * i] = val - hi * 4294967296; this._val32[2 * i + 1] = hi;
  }
  slice(n) {
    if (n === this.n) return this;
    const c = Object.create(Columns.prototype);
    c._root = this._root || this;                  // the allocation a pool takes back (DeviceGraph.giveColumns)
    c.n = n; c.id = this.id.subarray(0, n); c._id32 = this._id32.subarray(0, 2 * n);
    c.field = this.field.subarray(0, n); c.ts = this.ts.subarray(0, n); c.val = this.val.subarray(0, n);
    c._ts32 = this._ts32.subarray(0, 2 * n); c._val32 = this._val32.subarray(0, 2 * n);
    return c;
  }
}

const MAX_SAFE = Number.MAX_SAFE_INTEGER;
/* Field name of a node's CLOCK row on the device: (node id, fieldOf(collection, NODE_CLOCK)) holds ts = the clock the reference keeps in
 * meta[path].vectorClock (src/bullet.js:196-201) and val = the arrival number of the write that set it (so that a tie on ts goes to the LATER
 * write, as compare() does for two objects: src/bullet-crt.js:11-15). No JSON field name starts with U+0000. */
const NODE_CLOCK = "\u0000clock";
/* BMX_VAL_DELETED (include/bmx.h): the tombstone value of bmx_put_rows. -2^63 is exact as a double and splits into the halves 0x80000000:00000000 */
const VAL_DELETED = -9223372036854775808;
function isDeviceInt(v) { return typeof v === "number" && Number.isInteger(v) && v <= MAX_SAFE && v >= -MAX_SAFE; }
/* a clock the device understands: exactly one component, owned by `writer`, a non-negative safe integer */
function scalarClock(clock, writer) {
  if (!clock || typeof clock !== "object") return -1;
  let seen = 0;
  for (const k in clock) { if (k !== writer || ++seen > 1) return -1; }     // no key array allocated per entry
  if (seen !== 1) return -1;
  const t = clock[writer];
  return isDeviceInt(t) && t >= 0 ? t : -1;
}

/* N4 column builder: like Columns, with K uint32 clock components per row instead of one timestamp */
class VcColumns {
  constructor(n, K) {
    this.n = n; this.K = K;
    this.id = new BigUint64Array(n);
    this._id32 = new Uint32Array(this.id.buffer);
    this.field = new Uint32Array(n);
    this.clocks = new Uint32Array(n * K);
    this.keysets = new Uint32Array(n).fill(keysetDense(K));   // which writers each clock names, in which order (include/bmx.h): all K unless set
    this.val = new BigInt64Array(n);
    this._val32 = new Uint32Array(this.val.buffer);
  }
  set(i, idPair, field, comps, val, keyset) {
    this._id32[2 * i] = idPair[0]; this._id32[2 * i + 1] = idPair[1];
    this.field[i] = field;
    this.clocks.set(comps, i * this.K);
    if (keyset !== undefined) this.keysets[i] = keyset;
    this.val[i] = BigInt(val);
  }
  /* the id as two numbers, the counters taken from comps[0..K), val a safe integer written as two halves (no BigInt per row) */
  set2(i, lo, hi32, field, comps, keyset, val) {
    this._id32[2 * i] = lo; this._id32[2 * i + 1] = hi32;
    this.field[i] = field;
    const K = this.K, o = i * K;
    for (let k = 0; k < K; k++) this.clocks[o + k] = comps[k];
    this.keysets[i] = keyset;
    const hi = Math.floor(val / 4294967296);
    this._val32[2 * i] = val - hi * 4294967296; this._val32[2 * i + 1] = hi;
  }
  setKey(i, lo, hi32, field) { this._id32[2 * i] = lo; this._id32[2 * i + 1] = hi32; this.field[i] = field; }
  setVal(i, val) { const hi = Math.floor(val / 4294967296); this._val32[2 * i] = val - hi * 4294967296; this._val32[2 * i + 1] = hi; }
  slice(n) {
    if (n === this.n) return this;
    const c = Object.create(VcColumns.prototype);
    c.n = n; c.K = this.K; c.id = this.id.subarray(0, n); c._id32 = this._id32.subarray(0, 2 * n); c.field = this.field.subarray(0, n);
    c.clocks = this.clocks.subarray(0, n * this.K); c.keysets = this.keysets.subarray(0, n); c.val = this.val.subarray(0, n); c._val32 = this._val32.subarray(0, 2 * n);
    return c;
  }
}

/* key-set word of the N4 table (include/bmx.h bmx_vc_keyset): eight 4-bit writer indices in the clock object's key order, 0xF = end */
const KEYSET_NONE = 0xffffffff;
function keysetDense(K) { let ks = KEYSET_NONE; for (let k = 0; k < K; k++) ks = ((ks & ~(0xf << (4 * k))) | (k << (4 * k))) >>> 0; return ks; }
function keysetWriters(ks) { const out = []; for (let i = 0; i < 8; i++) { const w = (ks >>> (4 * i)) & 0xf; if (w === 0xf) break; out.push(w); } return out; }
/* clock object -> key-set word, its counters written into comps[0..K) (zero for writers it does not name); -1 when the clock names somebody outside
 * `index` (Map writer id -> position) or a counter is not a uint32: such a clock stays on the host */
function clockKeyset(clock, index, comps) {
  if (!clock || typeof clock !== "object") return -1;
  comps.fill(0);
  let ks = KEYSET_NONE, i = 0;
  for (const w in clock) {
    if (!Object.prototype.hasOwnProperty.call(clock, w)) continue;
    const k = index.get(w), c = clock[w];
    if (k === undefined || i >= 8 || typeof c !== "number" || !Number.isInteger(c) || c < 0 || c > 0xffffffff) return -1;
    comps[k] = c;
    ks = (ks & ~(0xf << (4 * i))) | (k << (4 * i));
    i++;
  }
  return ks >>> 0;
}

module.exports = { pathId, fieldId, idKey, KeyDictionary, Columns, VcColumns, isDeviceInt, scalarClock, clockKeyset, keysetDense, keysetWriters, KEYSET_NONE, fnv1a32, NODE_CLOCK, VAL_DELETED };
