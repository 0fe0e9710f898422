'use strict'
/* Test helper: runs THIS build's host JS (reze-engine_amd/host) and dumps arrays for pytest.
 * usage: node host_dump.js <mode> <args...>
 *   parse <pmx> <outdir>                 -> vertices/joints/weights/invbind/indices + info.json (+ morphs)
 *   pose  <fixture.json> <outdir>        -> FK from a skeleton description: world matrices for pose0 + tweens
 *   vmd   <vmd> <out.json>
 *   bonemorph <pmx> <spec.json> <outdir> -> world matrices with / without the morph weights of the spec applied
 */
const fs = require('fs'), path = require('path')
const host = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const { Model, PmxLoader, VMDLoader, Quat } = host
const dump = (p, ta) => fs.writeFileSync(p, Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength))
const mode = process.argv[2]
const quiet = console.warn; console.warn = () => {}

if (mode === 'parse') {
  const m = PmxLoader.loadFromBuffer(fs.readFileSync(process.argv[3]))
  const out = process.argv[4]
  dump(path.join(out, 'vertices.f32'), m.getVertices()); dump(path.join(out, 'joints.u16'), m.getSkinning().joints)
  dump(path.join(out, 'weights.u8'), m.getSkinning().weights); dump(path.join(out, 'invbind.f32'), m.getSkeleton().inverseBindMatrices)
  dump(path.join(out, 'indices.u32'), m.getIndices())
  const bones = m.getSkeleton().bones, mo = m.getMorphs()
  const info = { verts: m.getVertexCount(), indices: m.getIndices().length, bones: bones.length,
    append: bones.filter((b) => b.appendRotate || b.appendMove).length, materials: m.getMaterials().length,
    rigidbodies: m.getRigidbodies().length, joints: m.getJoints().length, boneNames: bones.map((b) => b.name),
    parents: bones.map((b) => b.parentIndex), bind: bones.map((b) => b.bindTranslation) }
  if (mo) {
    info.morphNames = mo.names; info.morphTypes = Array.from(mo.types); info.morphGroups = mo.groups
    dump(path.join(out, 'morph_offsets.u32'), mo.offsets); dump(path.join(out, 'morph_vidx.u32'), mo.vertexIndex)
    dump(path.join(out, 'morph_deltas.f32'), mo.deltas)
    const be = mo.boneEntries
    info.uvMorph = { morph: Array.from(mo.uvEntries.morph), vertex: Array.from(mo.uvEntries.vertex), delta: Array.from(mo.uvEntries.delta) }
    info.boneMorph = { morph: Array.from(be.morph), bone: Array.from(be.bone), translation: Array.from(be.translation), rotation: Array.from(be.rotation) }
  }
  fs.writeFileSync(path.join(out, 'info.json'), JSON.stringify(info))
} else if (mode === 'pose') {
  const fx = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'))
  const out = process.argv[4]
  const n = fx.parents.length
  const bones = []
  for (let i = 0; i < n; i++) {
    bones.push({ name: fx.names[i], parentIndex: fx.parents[i], bindTranslation: fx.bind[i], children: [],
      appendParentIndex: fx.appendParent[i] < 0 && !fx.appendRotate[i] && !fx.appendMove[i] ? undefined : fx.appendParent[i],
      appendRatio: fx.appendRotate[i] || fx.appendMove[i] ? fx.appendRatio[i] : undefined,
      appendRotate: fx.appendRotate[i], appendMove: fx.appendMove[i] })
  }
  const model = new Model(new Float32Array(8), new Uint32Array(3), [], [], { bones, inverseBindMatrices: new Float32Array(n * 16) },
    { joints: new Uint16Array(4), weights: new Uint8Array(4) })
  let now = 0
  model.setClock(() => now)
  // pose 0: set every local rotation directly (what rotateBones(..., 0) leaves behind)
  const names = fx.names, quats = []
  for (let i = 0; i < n; i++) quats.push(new Quat(fx.localRot[i][0], fx.localRot[i][1], fx.localRot[i][2], fx.localRot[i][3]))
  model.rotateBones(names, quats, 0)
  model.evaluatePose()
  dump(path.join(out, 'world_pose0.f32'), model.getBoneWorldMatrices())
  dump(path.join(out, 'localrot_pose0.f32'), model.runtimeSkeleton.localRotations)
  now = 1000
  model.rotateBones(fx.tweenBones, fx.tweenQuats.map((q) => new Quat(q[0], q[1], q[2], q[3])), 400)
  now = 1150
  model.evaluatePose()
  dump(path.join(out, 'world_tween150.f32'), model.getBoneWorldMatrices())
  dump(path.join(out, 'localrot_tween150.f32'), model.runtimeSkeleton.localRotations)
  now = 1500
  model.evaluatePose()
  dump(path.join(out, 'world_tween500.f32'), model.getBoneWorldMatrices())
} else if (mode === 'bonemorph') {
  const m = PmxLoader.loadFromBuffer(fs.readFileSync(process.argv[3]))
  const spec = JSON.parse(fs.readFileSync(process.argv[4], 'utf8'))
  const out = process.argv[5]
  const bones = m.getSkeleton().bones
  m.setClock(() => 0)
  m.rotateBones(bones.map((b) => b.name), spec.rot.map((q) => new Quat(q[0], q[1], q[2], q[3])), 0)
  m.evaluatePose()
  dump(path.join(out, 'world_unmorphed.f32'), m.getBoneWorldMatrices())
  dump(path.join(out, 'uv_rest.f32'), m.getMorphedUVs())
  m.setMorphWeights(Object.keys(spec.weights), Object.values(spec.weights))
  dump(path.join(out, 'effective.f32'), m.getEffectiveMorphWeights())
  dump(path.join(out, 'uv_morphed.f32'), m.getMorphedUVs())
  m.evaluatePose()
  dump(path.join(out, 'world_morphed.f32'), m.getBoneWorldMatrices())
  dump(path.join(out, 'localrot_after.f32'), m.runtimeSkeleton.localRotations)
  fs.writeFileSync(path.join(out, 'bm_info.json'), JSON.stringify({ parents: bones.map((b) => b.parentIndex), bind: bones.map((b) => b.bindTranslation),
    appendParent: bones.map((b) => (b.appendRotate && b.appendParentIndex !== undefined && b.appendParentIndex !== null ? b.appendParentIndex : -1)),
    appendRatio: bones.map((b) => (b.appendRatio === undefined || b.appendRatio === null ? 1 : b.appendRatio)) }))
} else if (mode === 'vmd') {
  const k = VMDLoader.loadFromBuffer(fs.readFileSync(process.argv[3]))
  fs.writeFileSync(process.argv[4], JSON.stringify({
    keyTimes: k.map((f) => [Math.round(f.time * 30), f.boneFrames.length]),
    frames: k.map((f) => ({ time: f.time, bones: f.boneFrames.map((b) => ({ name: b.boneName, frame: b.frame,
      rot: [b.rotation.x, b.rotation.y, b.rotation.z, b.rotation.w], pos: [b.position.x, b.position.y, b.position.z] })) })),
    morphFrames: k.morphFrames }))
} else {
  console.error('unknown mode'); process.exit(2)
}
console.warn = quiet
