'use strict'
/* DEV-CONTAINER ONLY (needs /root/reference): the reference's own Model (type-erased copy under argv[2], made by
 * tools/ref_erased_run.py's eraser — never stored in the repo) and this build's host/model.js are loaded with the same
 * real PMX and driven with the same random rotateBones / evaluatePose sequence on the same fake clock. After every
 * evaluatePose the local rotations and the world matrices must be IDENTICAL bit for bit.
 *   node ref_diff_fuzz.js <erased dir> <model.pmx> <seed> */
const fs = require('fs'), path = require('path')
const [erased, pmx, seedArg] = process.argv.slice(2)
let now = 1000
global.performance = { now: () => now }
global.fetch = (p) => Promise.resolve({ arrayBuffer: () => { const b = fs.readFileSync(p); return Promise.resolve(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)) } })
const ref = { PmxLoader: require(path.join(erased, 'pmx-loader')).PmxLoader, Quat: require(path.join(erased, 'math')).Quat }
const mine = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
let state = (parseInt(seedArg, 10) >>> 0) || 1
const rnd = () => { state |= 0; state = (state + 0x6D2B79F5) | 0; let t = Math.imul(state ^ (state >>> 15), 1 | state); t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296 }
const same = (a, b) => { if (a.length !== b.length) return false; const x = new Uint32Array(a.buffer, a.byteOffset, a.length), y = new Uint32Array(b.buffer, b.byteOffset, b.length); for (let i = 0; i < x.length; i++) if (x[i] !== y[i]) return i; return true }
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const R = await ref.PmxLoader.load(pmx)
  const M = await mine.PmxLoader.load(pmx)
  M.setClock(() => now)
  const names = R.getSkeleton().bones.map((b) => b.name)
  let evals = 0
  for (let step = 0; step < 120; step++) {
    const r = rnd()
    if (r < 0.5) {
      const k = 1 + Math.floor(rnd() * 6), bs = [], qa = []
      for (let i = 0; i < k; i++) {
        bs.push(rnd() < 0.05 ? 'no-such-bone' : names[Math.floor(rnd() * names.length)])
        const v = [rnd() - 0.5, rnd() - 0.5, rnd() - 0.5, rnd() - 0.3], n = Math.hypot(...v)
        qa.push(rnd() < 0.1 ? v : v.map((x) => x / n))          // now and then an un-normalised quaternion
      }
      const dur = [0, 1, 50, 333, 1000, undefined][Math.floor(rnd() * 6)]
      R.rotateBones(bs, qa.map((q) => new ref.Quat(q[0], q[1], q[2], q[3])), dur)
      M.rotateBones(bs, qa.map((q) => new mine.Quat(q[0], q[1], q[2], q[3])), dur)
    } else if (r < 0.7) {
      now += [0, 0.5, 16.7, 100, 400, 2000][Math.floor(rnd() * 6)]
    } else {
      R.evaluatePose(); M.evaluatePose()
      const a = same(R.runtimeSkeleton.localRotations, M.runtimeSkeleton.localRotations), b = same(R.getBoneWorldMatrices(), M.getBoneWorldMatrices())
      if (a !== true || b !== true) { console.error('DIVERGED at step ' + step + ' (t=' + now + '): localRotations ' + a + ', world ' + b); process.exit(1) }
      evals++
    }
  }
  console.warn = quiet
  console.log(JSON.stringify({ evals, bones: names.length }))
})().catch((e) => { console.error(e); process.exit(1) })
