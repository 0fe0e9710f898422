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
let live = 0
if (process.argv[2] === '--live') { // on a GPU box: the same with a LIVE context in front, a mesh + skeleton + pose behind it
  const ctx = a.create(0)
  const V = 300, B = 5
  const mesh = new Float32Array(V * 8).map((_, i) => Math.sin(i)), joints = new Uint16Array(V * 4), weights = new Uint8Array(V * 4).fill(64)
  const ib = new Float32Array(B * 16); for (let b = 0; b < B; b++) for (let k = 0; k < 4; k++) ib[b * 16 + k * 5] = 1
  a.uploadMesh(ctx, mesh, joints, weights); a.uploadSkeleton(ctx, ib); a.setPose(ctx, ib, null); a.deform(ctx)
  const tails = [[], [undefined], [null, null], [1, 2, 3], ['x', {}], [new Float32Array(3)], [new Float32Array(3), new Float32Array(1e6)], [new Uint32Array(2), new Uint32Array(1), new Float32Array(2)],
    [-1, -1, -1, -1], [70000, 70000, new Float32Array(3), new Float32Array(3)], [0, 0, 70000, new Float32Array(3), new Float32Array(3)], [{ trackBone: new Int32Array(2), keyOff: new Uint32Array([0, 5, 9]) }],
    [[ctx, ctx], 300, 0], ['grid_cap', 'x'], [new Uint8Array(7)]]
  for (const name of Object.keys(a).sort()) {
    if (typeof a[name] !== 'function' || name === 'destroy' || name === 'create') continue
    for (const args of tails) {
      if ((name === 'deformN' || name === 'timeFrames' || name === 'autotune') && typeof args[0] === 'number' && args[0] > 1000) continue // a legal, just very long, request
      const t0 = Date.now()
      try {
        const r = a[name](ctx, ...args); returned++
        if (name === 'fork') a.destroy(r) // a fork freezes the lender's static data while it lives
      } catch (e) { if (!(e instanceof Error)) throw new Error(name + ' threw a non-Error'); thrown++ }
      if (Date.now() - t0 > 300) console.error('SLOW ' + name + ' tail #' + tails.indexOf(args) + ' took ' + (Date.now() - t0) + ' ms')
      live++
    }
  }
  // uploadMorphsDense must hold exactly M*V*3 floats for THIS context's mesh: a shorter array (wrong shard, stale model)
  // would be read past its end by the C ABI's copy (round-1 review)
  for (const [M, len] of [[2, 6], [2, 2 * (V - 1) * 3], [3, 2 * V * 3]]) {
    let threw = false
    try { a.uploadMorphsDense(ctx, M, new Float32Array(len)) } catch (e) { threw = e instanceof Error }
    if (!threw) throw new Error('uploadMorphsDense accepted ' + len + ' floats for M=' + M + ' V=' + V)
  }
  a.uploadMorphsDense(ctx, 2, new Float32Array(2 * V * 3)); a.uploadMorphsDense(ctx, 0, null)
  let threwOv = false
  try { a.overrideWorld(ctx, new Uint32Array([1]), new Float32Array(15), null) } catch (e) { threwOv = e instanceof Error }
  if (!threwOv) throw new Error('overrideWorld accepted a short matrix array')
  let threwBm = 0
  for (const args of [[new Uint32Array([0]), new Uint32Array([0, 1]), new Float32Array(3), new Float32Array(4)],
    [new Uint32Array([0]), new Uint32Array([0]), new Float32Array(2), new Float32Array(4)], [new Uint32Array([0]), null, null, null],
    [new Uint32Array([0]), new Uint32Array([0]), new Float32Array(3), new Float32Array(4)] /* no topology on this context */]) {
    try { a.uploadBoneMorphs(ctx, ...args) } catch (e) { threwBm += e instanceof Error ? 1 : 0 }
  }
  if (threwBm !== 4) throw new Error('uploadBoneMorphs accepted ' + (4 - threwBm) + ' malformed calls')
  a.uploadBoneMorphs(ctx, null, null, null, null)   // n = 0 clears, always legal
  // forks: static uploads are refused on both sides while one lives, the lender cannot go first
  const fk = a.fork(ctx)
  let frozen = 0
  for (const c of [ctx, fk]) { try { a.uploadSkeleton(c, ib) } catch (e) { frozen += e instanceof Error ? 1 : 0 } }
  try { a.fork(fk) } catch (e) { frozen += e instanceof Error ? 1 : 0 }
  if (frozen !== 3) throw new Error('a live fork must freeze static data on both sides and cannot be forked itself (' + frozen + '/3)')
  a.setPose(ctx, ib, null); a.setPose(fk, ib, null); a.deformPair(ctx, fk, 5); a.sync(ctx); a.sync(fk)
  // destroying the lender while its fork lives is REFUSED — and must neither kill the handle nor leak the context (round-2 advisor finding)
  let refused = false
  try { a.destroy(ctx) } catch (e) { refused = e instanceof Error }
  if (!refused) throw new Error('destroy(lender) with a live fork must throw')
  a.setPose(ctx, ib, null); a.deform(ctx); a.sync(ctx)   // the handle is still the live context
  a.destroy(fk)
  a.uploadSkeleton(ctx, ib)                          // the lender owns its data again
  a.setPose(ctx, ib, null); a.deform(ctx)           // still usable
  // the launch-shape search as a table (rz_autotune_measure / _pick / _apply), and the communicator query without a communicator
  const tab = a.autotuneMeasure(ctx, 5)
  if (!Array.isArray(tab) || tab.length < 2 || !(tab[0].ms > 0) || tab[0].sameAs !== -1) throw new Error('autotuneMeasure: bad table ' + JSON.stringify(tab && tab[0]))
  const pk = a.autotunePick(tab)
  if (!(pk >= 0 && pk < tab.length)) throw new Error('autotunePick out of range')
  a.autotuneApply(ctx, tab[pk]); a.deform(ctx)
  let bad = 0
  try { a.autotuneApply(ctx, { morphSplit: 3 }) } catch (e) { bad += e instanceof Error ? 1 : 0 }
  try { a.autotunePick([]) } catch (e) { bad += e instanceof Error ? 1 : 0 }
  try { a.commInfo(ctx) } catch (e) { bad += e instanceof Error ? 1 : 0 }
  if (bad !== 3) throw new Error('malformed autotune entries / commInfo without a communicator must throw (' + bad + '/3)')
  const pos = new Float32Array(V * 3), nrm = new Float32Array(V * 3)
  a.read(ctx, 0, 0, V, pos, nrm)
  if (!pos.every(Number.isFinite)) throw new Error('context damaged by the misuse')
  // caller-written poses (ABI 7): mapPose hands out views over the pinned ring slot, commitPose detaches them; the frame is setPose's frame
  const w2 = ib.map((x, i) => (i % 16 === 12 ? 0.25 * (1 + (i >> 4)) : x))      // translations along x
  a.setPose(ctx, w2, null); a.deform(ctx)
  const want = new Float32Array(V * 3)
  a.read(ctx, 0, 0, V, want, null)
  a.setPose(ctx, ib, null); a.deform(ctx)
  const m = a.mapPose(ctx, 0)
  if (!(m.matrices instanceof Float32Array) || m.matrices.length !== B * 16 || m.morphWeights !== null) throw new Error('mapPose: bad views')
  m.matrices.set(w2)
  a.commitPose(ctx); a.deform(ctx)
  if (m.matrices.length !== 0) throw new Error('commitPose must detach the mapped views (length ' + m.matrices.length + ')')
  const got = new Float32Array(V * 3)
  a.read(ctx, 0, 0, V, got, null)
  for (let i = 0; i < V * 3; i++) if (got[i] !== want[i]) throw new Error('mapped pose: vertex float ' + i + ' differs from setPose')
  let mapBad = 0
  try { a.commitPose(ctx) } catch (e) { mapBad += e instanceof Error ? 1 : 0 }           // nothing mapped
  try { a.mapPose(ctx, 1) } catch (e) { mapBad += e instanceof Error ? 1 : 0 }           // rows are for poses of more than 256 KB
  try { a.mapPose(ctx, 7) } catch (e) { mapBad += e instanceof Error ? 1 : 0 }
  if (mapBad !== 3) throw new Error('commitPose without a mapping / rows for one character / an unknown layout must throw (' + mapBad + '/3)')
  const stale = a.mapPose(ctx, 0)
  a.destroy(ctx)
  if (stale.matrices.length !== 0) throw new Error('destroy must detach a mapped view')
  const ir = a.instanceRange(100, 8, 7)
  if (ir[0] !== 91 || ir[1] !== 9) throw new Error('instanceRange(100, 8, 7) = ' + ir)
}
console.log(JSON.stringify({ functions: Object.keys(a).length, thrown, returned, live, alive: true }))
