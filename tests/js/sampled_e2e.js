'use strict'
/* Device-side motion sampling through the N-API boundary: the same PMX + VMD posed at the same frames by
 *   A: new Engine(null, { realtime:false })                         host sampler (vmd-sampler.js) + host FK
 *   B: new Engine(null, { deviceFK:true, deviceSampling:true })     rz_upload_animation + rz_set_pose_sampled
 * usage: node sampled_e2e.js <model.pmx> <motion.vmd> <outdir> <morphLayout> <devices>
 * Dumps positions / normals / world matrices of both for pytest to compare. */
const fs = require('fs'), path = require('path')
const { Engine } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const [pmx, vmd, out, layout, devs] = process.argv.slice(2)
const dump = (name, ta) => fs.writeFileSync(path.join(out, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength))
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const devices = (devs || '0').split(',').map(Number)
  const A = new Engine(null, { realtime: false, morphLayout: layout, devices })
  const B = new Engine(null, { realtime: false, morphLayout: layout, devices, deviceFK: true, deviceSampling: true })
  for (const e of [A, B]) { await e.init(); await e.loadModel(pmx); await e.loadAnimation(vmd) }
  const frames = [0, 3.5, 11.5, 15, 22.75, 30, 99]
  fs.writeFileSync(path.join(out, 'frames.json'), JSON.stringify(frames))
  frames.forEach((f, i) => {
    A.seekFrame(f); B.seekFrame(f)
    const a = A.getDeformed(), b = B.getDeformed()
    dump('a_pos_' + i + '.f32', a.positions); dump('a_nrm_' + i + '.f32', a.normals)
    dump('b_pos_' + i + '.f32', b.positions); dump('b_nrm_' + i + '.f32', b.normals)
    dump('a_world_' + i + '.f32', A.currentModel.getBoneWorldMatrices())
    const gw = new Float32Array(A.currentModel.getBoneWorldMatrices().length)
    B.native.readWorld(B.ctx, 0, gw)
    dump('b_world_' + i + '.f32', gw)
    dump('a_mw_' + i + '.f32', A.currentModel.getEffectiveMorphWeights())
  })
  // a crowd: three copies of the model, each at its own frame, posed + deformed by one launch chain
  if (devices.length === 1) {
    const crowd = [3.5, 22.75, 11.5]
    B.setInstanceCount(crowd.length)
    B.seekFrame(crowd)
    crowd.forEach((f, k) => { const d = B.getDeformed(k); dump('crowd_pos_' + frames.indexOf(f) + '.f32', d.positions); dump('crowd_nrm_' + frames.indexOf(f) + '.f32', d.normals) })
  }
  A.dispose(); B.dispose()
  console.warn = quiet
})().catch((e) => { console.error(e); process.exit(1) })
