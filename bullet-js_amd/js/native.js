"use strict";
/*
 * native.js — loads the N-API addon (bullet-js_amd/bmx.node -> libbmx.so -> HIP kernels).
 * There is no JavaScript or CPU implementation of the batch/scan path behind this: if the addon is not
 * built, or no MI355X is present, the batch entry points throw.
 */
const path = require("path");

let addon = null;
let loadError = null;
try {
  addon = require(path.join(__dirname, "..", "bmx.node"));
} catch (e) {
  loadError = e;
}

function requireNative() {
  if (!addon) {
    const err = new Error("bmx: native addon not available (build with `make -C bullet-js_amd`): " + (loadError && loadError.message));
    err.code = "BMX_NO_ADDON";
    throw err;
  }
  return addon;
}

module.exports = { requireNative, available: () => addon !== null };
