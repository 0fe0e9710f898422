'use strict'
/* CPU test of { framesInFlight: 2 } with a recording stand-in for the native addon: consecutive frames alternate between the
 * context and ONE fork of it, reads go to the context of the frame rendered last, the fork is destroyed before static data is
 * replaced (loadModel, a new motion, a new instance count) and before the lender, bone overrides reach both contexts. */
const path = require('path')
const { Engine, Model } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const calls = []
let nextId = 1
const tag = (c) => (c && c.id) || '?'
const native = {
  create: () => ({ id: 'ctx' + nextId++ }), destroy: (c) => calls.push(['destroy', tag(c)]), fork: (c) => { const f = { id: 'fork' + nextId++ + '<' + tag(c) }; calls.push(['fork', tag(c), f.id]); return f },
  uploadMesh: (c) => calls.push(['uploadMesh', tag(c)]), uploadSkeleton: () => {}, uploadSkeletonTopology: () => {}, uploadBoneMorphs: () => {},
  setPose: (c) => calls.push(['setPose', tag(c)]), setPoseLocal: (c) => calls.push(['setPoseLocal', tag(c)]), deform: (c) => calls.push(['deform', tag(c)]),
  overrideWorld: (c, b) => calls.push(['overrideWorld', tag(c), b && Array.from(b)]), autotune: (c) => calls.push(['autotune', tag(c)]),
  read: (c) => calls.push(['read', tag(c)]), shardRange: (v) => [0, v],
}
const bones = ['root', 'a', 'b'].map((name, i) => ({ name, parentIndex: i - 1, bindTranslation: [0, 1, 0], children: [] }))
const mk = () => new Model(new Float32Array(8), new Uint32Array(3), [], [], { bones, inverseBindMatrices: new Float32Array(48) },
  { joints: new Uint16Array(4), weights: Uint8Array.from([255, 0, 0, 0]) }, [], [], null)
;(async () => {
  const out = {}
  const e = new Engine(null, { realtime: false, framesInFlight: 2, deviceFK: true, autotune: true })
  e.native = native; e.ctx = native.create(0); e.shards = [{ ctx: e.ctx, begin: 0, count: 1, fork: null, last: null, flip: 0 }]
  await e.setupModelBuffers(mk())
  for (let k = 0; k < 5; k++) { e.step(k * 10); e.getDeformed() }
  out.frames = calls.filter((c) => c[0] === 'deform' || c[0] === 'fork' || c[0] === 'autotune' || c[0] === 'read').map((c) => c.join(':'))
  calls.length = 0
  e.setBoneWorldOverrides([1], new Float32Array(16))
  out.overrides = calls.filter((c) => c[0] === 'overrideWorld').map((c) => c[1])
  calls.length = 0
  await e.setupModelBuffers(mk())              // a new model: the fork goes first, a new one is made on the next frame and gets the overrides
  e.step(100); e.step(110)
  out.reload = calls.map((c) => c.slice(0, 2).join(':'))
  calls.length = 0
  e.dispose()
  out.dispose = calls.map((c) => c.join(':'))
  let refused = false
  try { const m = new Engine(null, { framesInFlight: 2, devices: [0, 1] }); await m.init() } catch (err) { refused = /single GPU/.test(err.message) }
  out.multiGpuRefused = refused
  console.log(JSON.stringify(out))
})().catch((err) => { console.error(err); process.exit(1) })
