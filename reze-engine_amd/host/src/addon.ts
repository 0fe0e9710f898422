'use strict'
/*
 * Source: host/src/addon.ts (TypeScript). host/addon.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * addon.js — loads the N-API addon (reze_deform.node, built by csrc/Makefile from napi_addon.c on
 * top of libreze_deform.so). There is no JavaScript or CPU fallback for the deformation path: if
 * the addon is missing the require throws with build instructions, and Engine.init() rethrows —
 * the same contract as the reference's "WebGPU is not supported" error (engine.ts:160-163).
 */
import * as path from 'path'

import type { DeformAddon } from './types'
let native: DeformAddon | null = null
let loadError: Error | null = null
try {
  native = require(path.join(__dirname, '..', 'reze_deform.node'))
} catch (e) {
  loadError = e
}

function requireAddon(): DeformAddon {
  if (!native) {
    throw new Error('reze_deform.node is not available (' + (loadError && loadError.message) + '). Build it with ' +
      '`make -C reze-engine_amd/csrc` (hipcc --offload-arch=gfx950 + gcc); there is no CPU fallback.')
  }
  return native
}

const isAvailable = () => native !== null

export { requireAddon, isAvailable }
