'use strict'
/* Per-frame cost through the reference-language host (Node -> N-API -> C ABI -> MI355X): the same PMX + VMD played by
 *   host      : JS motion sampling + JS hierarchy solve, world matrices uploaded       (the reference's division of labour)
 *   deviceFK  : JS motion sampling, local rotations / translations uploaded, hierarchy on the GPU
 *   sampled   : one float per frame; sampling + hierarchy + deformation on the GPU
 *   sampled2 / deviceFK2 : the same with { framesInFlight: 2 } (frames alternate between the context and an rz_fork of it)
 * usage: node tools/node_frame_bench.js <model.pmx> <motion.vmd> [frames=2000] */
const path = require('path')
const { performance } = require('perf_hooks')
const { Engine } = require(path.join(__dirname, '..', 'reze-engine_amd', 'host'))
const [pmx, vmd, nArg] = process.argv.slice(2)
const N = parseInt(nArg || '2000', 10)
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  const out = {}
  for (const [name, opt] of [['host', {}], ['deviceFK', { deviceFK: true }], ['sampled', { deviceFK: true, deviceSampling: true }],
    ['deviceFK2', { deviceFK: true, framesInFlight: 2 }], ['sampled2', { deviceFK: true, deviceSampling: true, framesInFlight: 2 }]]) {
    const e = new Engine(null, Object.assign({ realtime: false, morphLayout: 'sparse', autotune: true }, opt))
    await e.init(); await e.loadModel(pmx); await e.loadAnimation(vmd)
    for (let i = 0; i < 200; i++) e.seekFrame((i * 0.37) % 60)
    e.native.sync(e.ctx)
    const t0 = performance.now()
    for (let i = 0; i < N; i++) e.seekFrame((i * 0.37) % 60)
    e.native.sync(e.ctx)
    if (e.shards[0].fork) e.native.sync(e.shards[0].fork)
    const us = (performance.now() - t0) * 1000 / N
    const t = e.measure(200)
    out[name] = { usPerFrame: +us.toFixed(2), gpuFrameUs: +(t.frameMs * 1000).toFixed(2), verts: e.currentModel.getVertexCount(), bones: e.currentModel.getSkeleton().bones.length }
    e.dispose()
  }
  console.warn = quiet
  console.log(JSON.stringify(out))
})().catch((e) => { console.error(e); process.exit(1) })
