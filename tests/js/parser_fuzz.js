'use strict'
/* Corrupt-input robustness of the host parsers: truncations, byte flips and wild counts near the header of a valid
 * PMX / VMD must end in a clean parse or a thrown Error — never a hang, a crash or a runaway allocation.
 * usage: node parser_fuzz.js <model.pmx> <motion.vmd> */
const path = require('path'), fs = require('fs')
const { PmxLoader, VMDLoader } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const base = fs.readFileSync(process.argv[2]), vbase = fs.readFileSync(process.argv[3])
let s = 12345
const rnd = () => { s = (s * 1664525 + 1013904223) >>> 0; return s / 4294967296 }
const out = { ok: 0, thrown: 0, slow: 0, notError: 0 }
const quiet = console.warn; console.warn = () => {}; const qe = console.error; console.error = () => {}
for (const [buf0, fn] of [[base, (b) => PmxLoader.loadFromBuffer(b)], [vbase, (b) => VMDLoader.loadFromBuffer(b)]]) {
  for (let it = 0; it < 300; it++) {
    const b = Buffer.from(buf0)
    let use = b
    if (it % 3 === 0) use = b.slice(0, Math.floor(rnd() * b.length))
    else if (it % 3 === 1) for (let k = 0; k < 4; k++) b[Math.floor(rnd() * b.length)] = Math.floor(rnd() * 256)
    else b.writeUInt32LE(Math.floor(rnd() * 0xffffffff) >>> 0, Math.floor(rnd() * Math.min(b.length - 4, 400)))
    const t0 = Date.now()
    try { fn(use); out.ok++ } catch (e) { out.thrown++; if (!(e instanceof Error)) out.notError++ }
    if (Date.now() - t0 > 1000) out.slow++
  }
}
console.warn = quiet; console.error = qe
console.log(JSON.stringify(out))
