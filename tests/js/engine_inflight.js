'use strict'
/* { framesInFlight: 2 } through the N-API boundary on the GPU: the same PMX driven through the same poses by a plain Engine and by
 * one that alternates frames between its context and a fork (rz_fork). Every frame's mesh / hull / bounds must be bit-identical.
 * usage: node engine_inflight.js <model.pmx> <motion.vmd> <out.json> <deviceFK 0|1> */
const fs = require('fs'), path = require('path')
const { Engine, Quat } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const [pmx, vmd, out, fk] = process.argv.slice(2)
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const opts = { realtime: false, morphLayout: 'sparse', deviceFK: fk === '1', deviceSampling: fk === '1', outline: true, bounds: true, autotune: true }
  const A = new Engine(null, opts), B = new Engine(null, Object.assign({ framesInFlight: 2 }, opts))
  for (const e of [A, B]) { await e.init(); await e.loadModel(pmx); await e.loadAnimation(vmd) }
  const names = A.currentModel.getBoneNames()
  const res = { frames: 0, mismatches: [], forked: false, moved: 0 }
  let first = null
  const same = (x, y) => { if (x.length !== y.length) return false; for (let i = 0; i < x.length; i++) if (x[i] !== y[i] && !(x[i] !== x[i] && y[i] !== y[i])) return false; return true }
  for (let k = 0; k < 9; k++) {
    for (const e of [A, B]) {
      if (k === 2) e.rotateBones([names[1], names[2]], [new Quat(0.2, 0.1, -0.1, 0.96), new Quat(-0.3, 0.0, 0.2, 0.93)], 300)
      if (k === 3) e.setMorphWeights(['grp', 'blink', 'twist'], [0.6, 0.3, 0.4])
      if (k >= 6) e.seekFrame(3.5 * k); else e.step(k * 80)
    }
    const a = A.getDeformed(), b = B.getDeformed()
    if (!first) first = Float32Array.from(a.positions)
    for (let i = 0; i < first.length; i++) res.moved = Math.max(res.moved, Math.abs(a.positions[i] - first[i]))
    if (!same(a.positions, b.positions) || !same(a.normals, b.normals)) res.mismatches.push('mesh@' + k)
    if (!same(A.getOutlineHull(), B.getOutlineHull())) res.mismatches.push('hull@' + k)
    const ba = A.getBounds(), bb = B.getBounds()
    if (JSON.stringify(ba) !== JSON.stringify(bb)) res.mismatches.push('bounds@' + k)
    res.frames++
  }
  res.forked = !!B.shards[0].fork
  res.lastOnFork = B.shards[0].last === B.shards[0].fork
  A.dispose(); B.dispose()
  fs.writeFileSync(out, JSON.stringify(res))
  console.warn = quiet
})().catch((e) => { console.error(e); process.exit(1) })
