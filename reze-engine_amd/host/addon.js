'use strict'
/*
 * addon.js — loads the N-API addon (reze_deform.node, built by csrc/Makefile from napi_addon.c on
 * top of libreze_deform.so). There is no JavaScript or CPU fallback for the deformation path: if
 * the addon is missing the require throws with build instructions, and Engine.init() rethrows —
 * the same contract as the reference's "WebGPU is not supported" error (engine.ts:160-163).
 */
const path = require('path')

let native = null
let loadError = null
try {
  native = require(path.join(__dirname, '..', 'reze_deform.node'))
} catch (e) {
  loadError = e
}

function requireAddon() {
  if (!native) {
    throw new Error('reze_deform.node is not available (' + (loadError && loadError.message) + '). Build it with ' +
      '`make -C reze-engine_amd/csrc` (hipcc --offload-arch=gfx950 + gcc); there is no CPU fallback.')
  }
  return native
}

module.exports = { requireAddon, isAvailable: () => native !== null }
