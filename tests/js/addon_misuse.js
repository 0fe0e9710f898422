'use strict'
/* Every export of reze_deform.node called with missing, mistyped and nonsensical arguments: a JS exception or a value is
 * fine, a crash is not (the process must reach the last line). Runs without a GPU: no call gets a live context. */
const path = require('path')
const a = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host', 'addon.js')).requireAddon()
const junk = [[], [undefined], [null, null, null], [1, 2, 3, 4, 5, 6], ['x', {}, []], [{}, new Float32Array(3), new Uint8Array(2), new Uint16Array(1)],
  [new Float32Array(0)], [[{}, {}], 10, 0], [Symbol.iterator], [() => 1, NaN, Infinity, -1]]
let thrown = 0, returned = 0
for (const name of Object.keys(a).sort()) {
  if (typeof a[name] !== 'function') continue
  for (const args of junk) {
    try { a[name](...args); returned++ } catch (e) { if (!(e instanceof Error)) throw new Error(name + ' threw a non-Error'); thrown++ }
  }
}
console.log(JSON.stringify({ functions: Object.keys(a).length, thrown, returned, alive: true }))
