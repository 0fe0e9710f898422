'use strict'
/* CPU test of the physics hand-off seam (engine.ts:2375-2391) with a recording stand-in for the native addon:
 *   host FK:   { physics } — step(dt, worldMats, inverseBind) is called between evaluatePose() and setPose, its in-place
 *              edit of the world matrices is what reaches the GPU, dt follows the engine clock;
 *   device FK: setBoneWorldOverrides(bones, matrices, instances?) reaches overrideWorld on every shard; host-FK engines refuse it. */
const path = require('path')
const { Engine, Model } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const calls = []
const native = {
  create: () => ({}), destroy: () => {}, uploadMesh: () => {}, uploadSkeleton: () => {}, uploadSkeletonTopology: () => calls.push(['topology']), uploadBoneMorphs: () => calls.push(['boneMorphs']),
  setPose: (c, w) => calls.push(['setPose', Array.from(w)]), setPoseLocal: () => calls.push(['setPoseLocal']), deform: () => calls.push(['deform']),
  overrideWorld: (c, b, w, i) => calls.push(['overrideWorld', b && Array.from(b), w && Array.from(w), i && Array.from(i)]),
  read: () => {}, shardRange: (v) => [0, v],
}
const bones = ['root', 'a', 'b'].map((name, i) => ({ name, parentIndex: i - 1, bindTranslation: [0, 1, 0], children: [] }))
const mk = () => new Model(new Float32Array(8), new Uint32Array(3), [], [], { bones, inverseBindMatrices: Float32Array.from({ length: 48 }, (_, k) => k) },
  { joints: new Uint16Array(4), weights: Uint8Array.from([255, 0, 0, 0]) }, [], [], null)
;(async () => {
  const out = {}
  const seen = []
  const physics = { step(dt, world, ib) { seen.push({ dt, n: world.length, ib0: ib[5], before: world[29] }); world.set([9, 9, 9, 1], 28) } }   // bone 1's translation column
  const e = new Engine(null, { realtime: false, physics })
  e.native = native; e.ctx = {}; e.shards = [{ ctx: e.ctx, begin: 0, count: 1 }]
  await e.setupModelBuffers(mk())
  e.step(0); e.step(50)
  out.seen = seen
  out.poses = calls.filter((c) => c[0] === 'setPose').map((c) => c[1].slice(28, 32))
  out.order = calls.map((c) => c[0])
  let refused = false
  try { e.setBoneWorldOverrides([1], new Float32Array(16)) } catch (err) { refused = /deviceFK/.test(err.message) }
  out.hostRefusesOverrides = refused
  calls.length = 0
  const d = new Engine(null, { realtime: false, deviceFK: true, physics })       // physics option is a host-FK seam: ignored here
  d.native = native; d.ctx = {}; d.shards = [{ ctx: d.ctx, begin: 0, count: 1 }, { ctx: {}, begin: 1, count: 1 }]
  await d.setupModelBuffers(mk())
  const m = Float32Array.from({ length: 32 }, (_, k) => k * 0.5)
  d.setBoneWorldOverrides([2, 0], m, [0, 0])
  d.step(0)
  d.setBoneWorldOverrides([], null)
  out.device = calls.filter((c) => c[0] === 'overrideWorld')
  out.deviceOrder = calls.map((c) => c[0])
  out.physicsCallsOnDeviceFK = seen.length - 2
  console.log(JSON.stringify(out))
})().catch((err) => { console.error(err); process.exit(1) })
