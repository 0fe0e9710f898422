'use strict'
/* Randomised walk over the host Engine API on the GPU (test infrastructure; uses the JS oracle):
 *   node engine_fuzz.js <model.pmx> <motion.vmd> <seed>
 * Random constructor options (shards, morph layout, device FK / sampling, outline, bounds, gather, autotune), then
 * random calls — rotateBones, setMorphWeights, playAnimation / stopAnimation, step(t), seekFrame(f) — and after every
 * rendered frame the deformed mesh is compared with oracle/js/skin_f32.js fed with the pose the HOST model holds
 * (device-sampling engines: the pose a fresh host model + VMDSampler evaluate at that frame). Exit code 1 on mismatch. */
const fs = require('fs'), path = require('path')
const root = path.join(__dirname, '..', '..')
const { Engine, Quat, PmxLoader, VMDLoader, VMDSampler } = require(path.join(root, 'reze-engine_amd', 'host'))
const oracle = require(path.join(root, 'oracle', 'js', 'skin_f32.js'))
const [pmx, vmd, seedArg] = process.argv.slice(2)
let state = (parseInt(seedArg, 10) >>> 0) || 1
const rnd = () => { state |= 0; state = (state + 0x6D2B79F5) | 0; let t = Math.imul(state ^ (state >>> 15), 1 | state); t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296 }
const pick = (a) => a[Math.floor(rnd() * a.length)]
const quat = () => { const v = [rnd() - 0.5, rnd() - 0.5, rnd() - 0.5, rnd() + 0.2]; const n = Math.hypot(...v); return new Quat(v[0] / n, v[1] / n, v[2] / n, v[3] / n) }

function expected(model) { // dense oracle frame from a host model's current pose + effective morph weights
  const V = model.getVertexCount(), B = model.getSkeleton().bones.length, vd = model.getVertices(), sk = model.getSkinning()
  const pos = new Float32Array(V * 3), nrm = new Float32Array(V * 3)
  for (let v = 0; v < V; v++) for (let k = 0; k < 3; k++) { pos[v * 3 + k] = vd[v * 8 + k]; nrm[v * 3 + k] = vd[v * 8 + 3 + k] }
  const mo = model.getMorphs(), M = mo ? mo.names.length : 0
  let deltas = null
  if (M > 0) {
    deltas = new Float32Array(M * V * 3)
    for (let m = 0; m < M; m++) for (let e = mo.offsets[m]; e < mo.offsets[m + 1]; e++) {
      const v = mo.vertexIndex[e]
      if (v < V) for (let k = 0; k < 3; k++) deltas[(m * V + v) * 3 + k] = Math.fround(deltas[(m * V + v) * 3 + k] + mo.deltas[e * 3 + k])
    }
  }
  const skin = oracle.palette(model.getBoneWorldMatrices(), model.getBoneInverseBindMatrices(), B, new Float32Array(B * 16))
  const op = new Float32Array(V * 3), on = new Float32Array(V * 3)
  oracle.deformRange(0, V, V, M, pos, nrm, sk.joints, sk.weights, skin, deltas, M > 0 ? model.getEffectiveMorphWeights() : null, op, on)
  return { op, on }
}

function compare(got, want, what) {
  let worst = 0
  const V = want.op.length / 3
  for (let v = 0; v < V; v++) {
    const dx = got.positions[v * 3] - want.op[v * 3], dy = got.positions[v * 3 + 1] - want.op[v * 3 + 1], dz = got.positions[v * 3 + 2] - want.op[v * 3 + 2]
    const n = Math.max(1, Math.hypot(want.op[v * 3], want.op[v * 3 + 1], want.op[v * 3 + 2]))
    const e = Math.hypot(dx, dy, dz) / n
    const en = Math.hypot(got.normals[v * 3] - want.on[v * 3], got.normals[v * 3 + 1] - want.on[v * 3 + 1], got.normals[v * 3 + 2] - want.on[v * 3 + 2])
    if (!(e <= 1e-4) || !(en <= 1e-4)) { console.error('MISMATCH ' + what + ' vertex ' + v + ' pos err ' + e + ' nrm err ' + en); process.exit(1) }
    worst = Math.max(worst, e)
  }
  return worst
}

;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const nShards = pick([1, 1, 2, 3])
  const deviceFK = rnd() < 0.5, deviceSampling = deviceFK && rnd() < 0.5
  const opt = {
    realtime: false, morphLayout: pick(['sparse', 'dense']), devices: new Array(nShards).fill(0), deviceFK, deviceSampling,
    outline: rnd() < 0.3, bounds: rnd() < 0.3, gather: nShards > 1 && rnd() < 0.5 ? 'direct' : false, autotune: rnd() < 0.3,
  }
  const engine = new Engine(null, opt)
  await engine.init(); await engine.loadModel(pmx); await engine.loadAnimation(vmd)
  let model = engine.currentModel
  const names = model.getBoneNames(), morphNames = model.getMorphNames()
  // device-sampling engines are stateless per seek: their reference is a fresh host model posed by the host sampler
  const shadow = await PmxLoader.load(pmx)
  const sampler = new VMDSampler(await VMDLoader.load(vmd))
  let t = 0, frames = 0, worst = 0
  const log = []
  for (let stepNo = 0; stepNo < 40; stepNo++) {
    const r = rnd()
    let rendered = null
    if (deviceSampling && nShards === 1 && !opt.outline && !opt.bounds && r < 0.25) {
      // a crowd: n copies, each at its own frame; every instance against the shadow model at that frame
      const n = 1 + Math.floor(rnd() * 4)
      engine.setInstanceCount(n)
      const fs2 = Array.from({ length: n }, () => rnd() * 40 - 2)
      log.push('crowd ' + fs2.map((x) => x.toFixed(1)).join('/'))
      engine.seekFrame(fs2)
      for (let k = 0; k < n; k++) {
        shadow.applySampledFrame(sampler, fs2[k]); shadow.evaluatePose()
        worst = Math.max(worst, compare(engine.getDeformed(k), expected(shadow), JSON.stringify(opt) + ' crowd instance ' + k + ' after ' + log.slice(-6).join(' | ')))
      }
      engine.setInstanceCount(1)
      frames++
    } else if (rnd() < 0.04) {
      log.push('reload'); await engine.loadModel(pmx)       // the same file again: fresh model, fresh device buffers
      model = engine.currentModel
    } else if (deviceSampling || r < 0.2) {
      const f = rnd() * 40 - 2
      log.push('seek ' + f.toFixed(2))
      engine.seekFrame(f)
      if (deviceSampling) { shadow.applySampledFrame(sampler, f); shadow.evaluatePose(); rendered = shadow }
      else { if (deviceFK) model.computeWorldMatrices(); rendered = model }
    } else if (r < 0.4) {
      const k = 1 + Math.floor(rnd() * 3), bs = [], qs = []
      for (let i = 0; i < k; i++) { bs.push(pick(names)); qs.push(quat()) }
      if (rnd() < 0.2) bs.push('no-such-bone'), qs.push(quat())
      log.push('rotate ' + bs.join(','))
      engine.rotateBones(bs, qs, pick([0, 50, 400, undefined]))
    } else if (r < 0.55 && morphNames.length > 0) {
      const k = 1 + Math.floor(rnd() * 3), ms = [], ws = []
      for (let i = 0; i < k; i++) { ms.push(pick(morphNames)); ws.push(Math.round(rnd() * 100) / 100) }
      log.push('morph ' + ms.join(','))
      engine.setMorphWeights(ms, ws)
    } else if (r < 0.62) {
      log.push('play'); engine.playAnimation()
    } else if (r < 0.67) {
      log.push('stop'); engine.stopAnimation()
    } else {
      t += pick([0, 16.7, 33.4, 250, 1000])
      log.push('step ' + t)
      engine.step(t)
      if (deviceFK) model.computeWorldMatrices()
      rendered = model
    }
    if (rendered) {
      worst = Math.max(worst, compare(engine.getDeformed(), expected(rendered), JSON.stringify(opt) + ' after ' + log.slice(-8).join(' | ')))
      frames++
    }
  }
  engine.dispose()
  console.warn = quiet
  console.log(JSON.stringify({ frames, worst, opt }))
})().catch((e) => { console.error(e); process.exit(1) })
