'use strict'
/*
 * Source: host/src/model.ts (TypeScript). host/model.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * model.js — scene data + CPU pose solve (stays on the host, north star: "the PMX/VMD parsers and
 * CPU bone-hierarchy solve stay in TypeScript").
 *
 * Mirrors the reference's class Model (engine/src/model.ts:70-421): same constructor argument
 * order, same getters (getVertices / getSkinning / getSkeleton / getBoneWorldMatrices / ...),
 * same rotateBones() tween semantics (:246-315), updateRotationTweens (:158-194) and forward
 * kinematics with MMD append-rotation (:330-420), same typed-array state (SkeletonRuntime,
 * :52-59). World matrices come out bit-identical to the reference because every intermediate
 * matrix is stored to Float32Array at the same points.
 *
 * Additions the reference lacks (SURVEY §8b): an injectable clock (the reference reads
 * performance.now(), model.ts:160,249 — not reproducible frame for frame) and vertex-morph state
 * (names, sparse targets, weights, group-morph flattening) feeding the fused GPU kernel.
 */
import { Quat, easeInOut, kernels } from './math'
import type { Bone, Material, MorphSet, NumArray, PosedLocals, RotTweenState, Skeleton, SkeletonRuntime, Skinning, Texture } from './types'
import type { VMDSampler } from './vmd-sampler'
const { slerpInto, mulInto, quatToMatInto, identityInto } = kernels

const VERTEX_STRIDE = 8 // floats per vertex: x y z nx ny nz u v  (model.ts:4, :196-200)

let defaultClock: () => number
try {
  const { performance } = require('perf_hooks')
  defaultClock = () => performance.now()
} catch (e) {
  defaultClock = () => Date.now()
}

class Model {
  vertexData: Float32Array
  vertexCount: number
  indexData: Uint32Array
  textures: Texture[]
  materials: Material[]
  skeleton: Skeleton
  skinning: Skinning
  rigidbodies: unknown[]
  joints: unknown[]
  clock: () => number
  runtimeSkeleton: SkeletonRuntime
  rotTweenState: RotTweenState
  solveOrder: number[]
  _rot: Float32Array
  _app: Float32Array
  _tmpA: Float32Array
  _tmpB: Float32Array
  _tr: Float32Array
  _q: number[]
  applyLocalTranslations: boolean
  morphs: MorphSet | null
  morphWeights: Float32Array
  effectiveMorphWeights: Float32Array
  morphNameIndex: Record<string, number>
  hasBoneMorphs: boolean
  poseRotations: Float32Array
  poseTranslations: Float32Array
  _uv?: Float32Array
  _uvWeights?: Float32Array
  /**
   * @param {Float32Array} vertexData interleaved 8 floats / vertex
   * @param {Uint32Array} indexData
   * @param {Array} textures
   * @param {Array} materials
   * @param {{bones: Array, inverseBindMatrices: Float32Array}} skeleton
   * @param {{joints: Uint16Array, weights: Uint8Array}} skinning
   * @param {Array} [rigidbodies]
   * @param {Array} [joints]
   * @param {object|null} [morphs] { names, types, offsets:Uint32Array(M+1), vertexIndex:Uint32Array,
   *                                 deltas:Float32Array(E*3), groups: Array<Array<[child, ratio]>|null> }
   */
  constructor(vertexData: Float32Array, indexData: Uint32Array, textures: Texture[], materials: Material[], skeleton: Skeleton, skinning: Skinning, rigidbodies?: unknown[], joints?: unknown[], morphs?: MorphSet | null) {
    if (!skeleton || !skeleton.bones || skeleton.bones.length === 0) throw new Error('Model has no bones')
    this.vertexData = vertexData
    this.vertexCount = vertexData.length / VERTEX_STRIDE
    this.indexData = indexData
    this.textures = textures || []
    this.materials = materials || []
    this.skeleton = skeleton
    this.skinning = skinning
    this.rigidbodies = rigidbodies || []
    this.joints = joints || []
    this.clock = defaultClock

    const n = skeleton.bones.length
    const nameIndex = {}
    for (let i = 0; i < n; i++) nameIndex[skeleton.bones[i].name] = i
    const localRotations = new Float32Array(n * 4)
    for (let i = 0; i < n; i++) localRotations[i * 4 + 3] = 1
    this.runtimeSkeleton = {
      nameIndex,
      localRotations, // quat per bone (x,y,z,w)
      localTranslations: new Float32Array(n * 3),
      worldMatrices: new Float32Array(n * 16),
      computedBones: new Array(n).fill(false),
    }
    this.rotTweenState = {
      active: new Uint8Array(n),
      startQuat: new Float32Array(n * 4),
      targetQuat: new Float32Array(n * 4),
      startTimeMs: new Float32Array(n),
      durationMs: new Float32Array(n),
    }
    // parent-first evaluation order (the reference recurses; the result per bone is the same)
    this.solveOrder = Model.parentFirstOrder(skeleton.bones)
    // scratch matrices for the FK (no per-bone allocation)
    this._rot = new Float32Array(16)
    this._app = new Float32Array(16)
    this._tmpA = new Float32Array(16)
    this._tmpB = new Float32Array(16)
    this._tr = new Float32Array(16)
    this._q = [0, 0, 0, 1]

    // Frame-sampled VMD playback may also move bones (センター etc.). The reference never writes localTranslations and
    // applies them only through append-move (model.ts:388-393), so this stays off unless a sampler turns it on.
    this.applyLocalTranslations = false

    this.morphs = morphs || null
    const m = this.morphs ? this.morphs.names.length : 0
    this.morphWeights = new Float32Array(m) // as set by the user / animation (includes group morphs)
    this.effectiveMorphWeights = new Float32Array(m) // group morphs flattened onto their children
    this.morphNameIndex = {}
    for (let i = 0; i < m; i++) this.morphNameIndex[this.morphs.names[i]] = i
    // PMX bone morphs (type 2): the local pose the hierarchy solve sees = the runtime's local pose with them folded in
    this.hasBoneMorphs = !!(this.morphs && this.morphs.boneEntries && this.morphs.boneEntries.morph.length > 0)
    this.poseRotations = new Float32Array(this.hasBoneMorphs ? n * 4 : 0)
    this.poseTranslations = new Float32Array(this.hasBoneMorphs ? n * 3 : 0)
  }

  static parentFirstOrder(bones: Bone[]): number[] {
    const n = bones.length
    const state = new Uint8Array(n)
    const order = []
    for (let i = 0; i < n; i++) {
      if (state[i]) continue
      const chain = []
      let b = i
      while (b >= 0 && b < n && !state[b]) { state[b] = 1; chain.push(b); b = bones[b].parentIndex }
      for (let k = chain.length - 1; k >= 0; k--) order.push(chain[k])
    }
    return order
  }

  /** Replace performance.now() (deterministic tests / offline stepping). */
  setClock(fn: (() => number) | null): void { this.clock = fn || defaultClock }

  // ---- static data getters (model.ts:196-238) ----
  getVertices(): Float32Array { return this.vertexData }
  getTextures(): Texture[] { return this.textures }
  getMaterials(): Material[] { return this.materials }
  getVertexCount(): number { return this.vertexCount }
  getIndices(): Uint32Array { return this.indexData }
  getSkeleton(): Skeleton { return this.skeleton }
  getSkinning(): Skinning { return this.skinning }
  getRigidbodies(): unknown[] { return this.rigidbodies }
  getJoints(): unknown[] { return this.joints }
  getBoneNames(): string[] { return this.skeleton.bones.map((b) => b.name) }
  getBoneWorldMatrices(): Float32Array { return this.runtimeSkeleton.worldMatrices }
  getBoneInverseBindMatrices(): Float32Array { return this.skeleton.inverseBindMatrices }

  // ---- pose API ----
  _tweenValue(idx: number, now: number, out: NumArray): number {
    const st = this.rotTweenState
    const qi = idx * 4
    const dur = Math.max(1, st.durationMs[idx])
    const t = Math.max(0, Math.min(1, (now - st.startTimeMs[idx]) / dur))
    slerpInto(out, st.startQuat[qi], st.startQuat[qi + 1], st.startQuat[qi + 2], st.startQuat[qi + 3],
      st.targetQuat[qi], st.targetQuat[qi + 1], st.targetQuat[qi + 2], st.targetQuat[qi + 3], easeInOut(t))
    return t
  }

  /** model.ts:246-315 — immediate set (durationMs 0/undefined) or arm a quadratic-ease slerp tween. */
  rotateBones(names: string[], quats: Quat[], durationMs?: number): void {
    const st = this.rotTweenState
    const rot = this.runtimeSkeleton.localRotations
    const nameIndex = this.runtimeSkeleton.nameIndex
    const now = this.clock()
    const dur = durationMs && durationMs > 0 ? durationMs : 0
    const cur = this._q
    for (let i = 0; i < names.length; i++) {
      const found = nameIndex[names[i]]
      const idx = found === undefined || found === null ? -1 : found
      if (idx < 0 || idx >= this.skeleton.bones.length) continue
      const q = quats[i].normalize()
      const qi = idx * 4
      if (dur === 0) {
        rot[qi] = q.x; rot[qi + 1] = q.y; rot[qi + 2] = q.z; rot[qi + 3] = q.w
        st.active[idx] = 0
        continue
      }
      // start from where the bone is now: the running tween's interpolated value, else the stored rotation
      let sx = rot[qi], sy = rot[qi + 1], sz = rot[qi + 2], sw = rot[qi + 3]
      if (st.active[idx] === 1) {
        this._tweenValue(idx, now, cur)
        sx = cur[0]; sy = cur[1]; sz = cur[2]; sw = cur[3]
      }
      st.startQuat[qi] = sx; st.startQuat[qi + 1] = sy; st.startQuat[qi + 2] = sz; st.startQuat[qi + 3] = sw
      st.targetQuat[qi] = q.x; st.targetQuat[qi + 1] = q.y; st.targetQuat[qi + 2] = q.z; st.targetQuat[qi + 3] = q.w
      st.startTimeMs[idx] = now
      st.durationMs[idx] = dur
      st.active[idx] = 1
    }
  }

  /** model.ts:158-194 */
  updateRotationTweens(): void {
    const st = this.rotTweenState
    const rot = this.runtimeSkeleton.localRotations
    const now = this.clock()
    const cur = this._q
    for (let i = 0, n = this.skeleton.bones.length; i < n; i++) {
      if (st.active[i] !== 1) continue
      const t = this._tweenValue(i, now, cur)
      const qi = i * 4
      rot[qi] = cur[0]; rot[qi + 1] = cur[1]; rot[qi + 2] = cur[2]; rot[qi + 3] = cur[3]
      if (t >= 1) st.active[i] = 0
    }
  }

  /** model.ts:325-328 */
  evaluatePose(): void {
    this.updateRotationTweens()
    this.computeWorldMatrices()
  }

  /**
   * The local pose the hierarchy solve consumes. Without bone morphs (or with all of them at weight 0) these ARE the
   * runtime arrays (model.ts:55-56). A PMX bone morph (type 2; the reference only skips the section,
   * pmx-loader.ts:489-497, so the semantics are this build's, the usual MMD ones) with effective weight w adds
   * w * translation to its bone's local translation and right-multiplies the local rotation by
   * Quat.slerp(identity, rotation, w) (math.ts:156-189, :77-85); entries fold in ascending morph order; the runtime arrays
   * (tween / animation state) are left untouched. GPU twin: fk_solve's bone-morph pass (csrc/deform_kernels.hip).
   */
  posedLocals(): PosedLocals {
    const rs = this.runtimeSkeleton
    const raw = { rot: rs.localRotations, tra: rs.localTranslations, moved: false }
    if (!this.hasBoneMorphs) return raw
    const w = this.getEffectiveMorphWeights()
    const be = this.morphs.boneEntries
    const n = be.morph.length
    let any = false
    for (let k = 0; k < n && !any; k++) any = w[be.morph[k]] !== 0
    if (!any) return raw
    const rot = this.poseRotations, tra = this.poseTranslations
    rot.set(rs.localRotations); tra.set(rs.localTranslations)
    const s = this._q
    for (let k = 0; k < n; k++) {
      const wk = w[be.morph[k]]
      if (wk === 0) continue
      const b = be.bone[k], qi = b * 4, ti = b * 3
      tra[ti] += wk * be.translation[k * 3]; tra[ti + 1] += wk * be.translation[k * 3 + 1]; tra[ti + 2] += wk * be.translation[k * 3 + 2]
      slerpInto(s, 0, 0, 0, 1, be.rotation[k * 4], be.rotation[k * 4 + 1], be.rotation[k * 4 + 2], be.rotation[k * 4 + 3], wk)
      const x = rot[qi], y = rot[qi + 1], z = rot[qi + 2], ww = rot[qi + 3] // Hamilton product local * s
      rot[qi] = ww * s[0] + x * s[3] + y * s[2] - z * s[1]
      rot[qi + 1] = ww * s[1] - x * s[2] + y * s[3] + z * s[0]
      rot[qi + 2] = ww * s[2] + x * s[1] - y * s[0] + z * s[3]
      rot[qi + 3] = ww * s[3] - x * s[0] - y * s[1] - z * s[2]
    }
    return { rot, tra, moved: true }
  }

  /**
   * model.ts:330-420. Per bone: R = fromQuat(q); append-rotation R = fromQuat(slerp(I, +-q_append,
   * |ratio|)) * R when appendRotate && valid parent && |clamp(ratio,-1,1)| > 1e-6; append-move only
   * inside that branch; L = T(bind) * R * T(add); W = W_parent * L.
   */
  computeWorldMatrices(): void {
    const bones = this.skeleton.bones
    const n = bones.length
    const { rot, tra, moved } = this.posedLocals()
    const useT = this.applyLocalTranslations || moved
    const world = this.runtimeSkeleton.worldMatrices
    const R = this._rot, A = this._app, T = this._tr, X = this._tmpA, L = this._tmpB
    const q = this._q
    for (let k = 0; k < n; k++) {
      const i = this.solveOrder[k]
      const b = bones[i]
      if (b.parentIndex >= n) console.warn('[RZM] bone ' + i + ' parent out of range: ' + b.parentIndex)
      const qi = i * 4
      quatToMatInto(R, 0, rot[qi], rot[qi + 1], rot[qi + 2], rot[qi + 3])
      let ax = 0, ay = 0, az = 0
      let rotM = R
      const ap = b.appendParentIndex
      if (b.appendRotate && ap !== undefined && ap !== null && ap >= 0 && ap < n) {
        const ratio = b.appendRatio === undefined || b.appendRatio === null ? 1 : Math.max(-1, Math.min(1, b.appendRatio))
        if (Math.abs(ratio) > 1e-6) {
          const aq = ap * 4
          let qx = rot[aq], qy = rot[aq + 1], qz = rot[aq + 2]
          const qw = rot[aq + 3]
          if (ratio < 0) { qx = -qx; qy = -qy; qz = -qz }
          slerpInto(q, 0, 0, 0, 1, qx, qy, qz, qw, ratio < 0 ? -ratio : ratio)
          quatToMatInto(A, 0, q[0], q[1], q[2], q[3])
          mulInto(X, 0, A, 0, R, 0)
          rotM = X
          if (b.appendMove) {
            const r = b.appendRatio === undefined || b.appendRatio === null ? 1 : b.appendRatio
            ax = tra[ap * 3] * r; ay = tra[ap * 3 + 1] * r; az = tra[ap * 3 + 2] * r
          }
        }
      }
      // L = T(bind) * rotM * T(add), each product stored as f32 like the reference's Mat4.multiply chain
      identityInto(T, 0)
      T[12] += b.bindTranslation[0]; T[13] += b.bindTranslation[1]; T[14] += b.bindTranslation[2]
      if (useT) { T[12] += tra[i * 3]; T[13] += tra[i * 3 + 1]; T[14] += tra[i * 3 + 2] }
      mulInto(L, 0, T, 0, rotM, 0)
      identityInto(T, 0)
      T[12] += ax; T[13] += ay; T[14] += az
      const L2 = rotM === X ? R : X // a free scratch matrix
      mulInto(L2, 0, L, 0, T, 0)
      const wo = i * 16
      if (b.parentIndex >= 0) {
        mulInto(T, 0, world, b.parentIndex * 16, L2, 0)
        world.set(T, wo)
      } else {
        world.set(L2, wo)
      }
    }
    this.runtimeSkeleton.computedBones.fill(true)
  }

  /** Pose every bone the sampler keys at `frame` (rotation + translation), and every morph it keys. Un-keyed bones keep their state. */
  applySampledFrame(sampler: VMDSampler, frame: number): void {
    const rot = this.runtimeSkeleton.localRotations, tra = this.runtimeSkeleton.localTranslations
    this.applyLocalTranslations = true
    for (const name of sampler.boneNames()) {
      const idx = this.runtimeSkeleton.nameIndex[name]
      if (idx === undefined) continue
      const s = sampler.sampleBone(name, frame)
      rot[idx * 4] = s.rotation[0]; rot[idx * 4 + 1] = s.rotation[1]; rot[idx * 4 + 2] = s.rotation[2]; rot[idx * 4 + 3] = s.rotation[3]
      tra[idx * 3] = s.position[0]; tra[idx * 3 + 1] = s.position[1]; tra[idx * 3 + 2] = s.position[2]
      this.rotTweenState.active[idx] = 0
    }
    for (const name of sampler.morphNames()) {
      const w = sampler.sampleMorph(name, frame)
      if (w !== null) this.setMorphWeights([name], [w])
    }
  }

  // ---- morphs (no reference counterpart; PMX layout per pmx-loader.ts:471-488) ----
  getMorphNames(): string[] { return this.morphs ? this.morphs.names.slice() : [] }
  getMorphCount(): number { return this.morphs ? this.morphs.names.length : 0 }
  getMorphs(): MorphSet | null { return this.morphs }

  /** names or indices + weights; unknown names are ignored like unknown bones in rotateBones. */
  setMorphWeights(namesOrIndices: Array<string | number>, weights: ArrayLike<number>): void {
    if (!this.morphs) return
    for (let i = 0; i < namesOrIndices.length; i++) {
      const key = namesOrIndices[i]
      const idx = typeof key === 'number' ? key : this.morphNameIndex[key]
      if (idx === undefined || idx < 0 || idx >= this.morphWeights.length) continue
      this.morphWeights[idx] = weights[i]
    }
  }

  getMorphWeights(): Float32Array { return this.morphWeights }

  /**
   * Weights the deformation consumes: vertex morphs (type 1) and bone morphs (type 2) keep their own weight plus, for
   * every group morph (type 0) that lists them, w_group * ratio (pmx-loader.ts:479-482). Other morph types
   * (UV / material / flip / impulse) touch neither positions nor bones and contribute nothing.
   */
  getEffectiveMorphWeights(): Float32Array { return this._flattenGroups(this.effectiveMorphWeights, 1, 2) }

  // own weight of the morphs whose type lies in [lo, hi], plus what group morphs feed them
  _flattenGroups(out: Float32Array, lo: number, hi: number): Float32Array {
    if (!this.morphs) return out
    const { types, groups } = this.morphs
    const w = this.morphWeights
    for (let i = 0; i < out.length; i++) out[i] = types[i] >= lo && types[i] <= hi ? w[i] : 0
    for (let g = 0; g < out.length; g++) {
      if (types[g] !== 0 || w[g] === 0 || !groups[g]) continue
      for (const [child, ratio] of groups[g]) {
        if (child >= 0 && child < out.length && types[child] >= lo && types[child] <= hi) out[child] += w[g] * ratio
      }
    }
    return out
  }

  /**
   * Texture coordinates with the UV morphs (PMX type 3) applied: uv + sum over entries of w_morph * (du, dv), ascending
   * morph order, f32. UVs never pass through the deformation kernel — vs() forwards them untouched (engine.ts:273) and a
   * renderer binds them from the static vertex buffer — so this is a sparse host-side update of a V x 2 array, not GPU
   * work. The reference has no counterpart (its loader skips the section, pmx-loader.ts:498-507).
   */
  getMorphedUVs(): Float32Array {
    const V = this.vertexCount
    if (!this._uv) this._uv = new Float32Array(V * 2)
    const uv = this._uv, vd = this.vertexData
    for (let v = 0; v < V; v++) { uv[v * 2] = vd[v * VERTEX_STRIDE + 6]; uv[v * 2 + 1] = vd[v * VERTEX_STRIDE + 7] }
    const ue = this.morphs && this.morphs.uvEntries
    if (!ue || ue.morph.length === 0) return uv
    if (!this._uvWeights) this._uvWeights = new Float32Array(this.morphWeights.length)
    const w = this._flattenGroups(this._uvWeights, 3, 3)
    for (let k = 0; k < ue.morph.length; k++) {
      const wk = w[ue.morph[k]]
      if (wk === 0) continue
      const v = ue.vertex[k]
      uv[v * 2] += wk * ue.delta[k * 2]; uv[v * 2 + 1] += wk * ue.delta[k * 2 + 1]
    }
    return uv
  }
}

export { Model, VERTEX_STRIDE }
