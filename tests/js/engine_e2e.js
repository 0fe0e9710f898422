'use strict'
/* End-to-end through the N-API boundary: Engine (host JS) -> reze_deform.node -> libreze_deform.so -> MI355X.
 * usage: node engine_e2e.js <model.pmx> <motion.vmd|-> <outdir> <morphLayout>
 * Dumps, for three deterministic clock steps, the exact inputs the GPU consumed (world matrices, effective morph
 * weights) and the deformed output it produced; pytest recomputes the frame with the CPU oracle from those inputs. */
const fs = require('fs'), path = require('path')
const { Engine, Quat } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const [pmx, vmd, out, layout, devs] = process.argv.slice(2)
const dump = (name, ta) => fs.writeFileSync(path.join(out, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength))
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const deviceFK = /:fk/.test(devs || '')
  const direct = /:direct/.test(devs || '')      // peer-direct gather: every shard stores into shard 0's gathered buffer
  const devices = (devs || '0').replace(':fk', '').replace(':direct', '').split(',').map(Number)   // '0,0' = two contexts (two vertex shards) on one GPU
  const engine = new Engine(null, { realtime: false, morphLayout: layout || 'sparse', ambient: 0.8, devices, deviceFK, outline: true, bounds: true,
    gather: direct ? 'direct' : false, autotune: true })
  await engine.init()
  await engine.loadModel(pmx)
  const model = engine.currentModel
  dump('vertices.f32', model.getVertices()); dump('joints.u16', model.getSkinning().joints)
  dump('weights.u8', model.getSkinning().weights); dump('invbind.f32', model.getSkeleton().inverseBindMatrices)
  const mo = model.getMorphs()
  dump('morph_offsets.u32', mo.offsets); dump('morph_vidx.u32', mo.vertexIndex); dump('morph_deltas.f32', mo.deltas)
  if (vmd !== '-') { await engine.loadAnimation(vmd); engine.playAnimation() }
  const names = model.getBoneNames()
  const steps = [0, 250, 1000, -1]        // -1: a frame-indexed seek (MMD interpolation, bone translations) instead of a clock step
  for (let s = 0; s < steps.length; s++) {
    if (s === 1) {
      engine.rotateBones([names[1], names[2]], [new Quat(0.2, 0.1, -0.1, 0.96), new Quat(-0.3, 0.0, 0.2, 0.93)], 500)
      engine.setMorphWeights(['grp', 'blink'], [0.6, 0.3])      // a group morph fans out onto its vertex morphs
    }
    if (steps[s] < 0) { if (vmd === '-') break; engine.seekFrame(11.5) } else engine.step(steps[s])
    const d = engine.getDeformed()
    if (deviceFK) {   // the host did not solve the hierarchy this frame: do it now for the oracle, and fetch the GPU's solve
      model.computeWorldMatrices()
      const gw = new Float32Array(model.getBoneWorldMatrices().length)
      engine.native.readWorld(engine.ctx, 0, gw)
      dump('gpuworld_' + s + '.f32', gw)
    }
    dump('world_' + s + '.f32', model.getBoneWorldMatrices())
    dump('mw_' + s + '.f32', model.getEffectiveMorphWeights())
    dump('pos_' + s + '.f32', d.positions); dump('nrm_' + s + '.f32', d.normals)
    dump('hull_' + s + '.f32', engine.getOutlineHull())
    const bb = engine.getBounds(); dump('bounds_' + s + '.f32', Float32Array.from(bb.min.concat(bb.max)))
    if (s === 0) dump('edge.f32', engine.edgeScale)
  }
  const t = engine.measure(20)
  const st = engine.getStats()
  fs.writeFileSync(path.join(out, 'stats.json'), JSON.stringify({ t, st, names: mo.names }))
  engine.dispose()
  console.warn = quiet
})().catch((e) => { console.error(e); process.exit(1) })
