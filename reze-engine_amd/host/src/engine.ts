'use strict'
/*
 * Source: host/src/engine.ts (TypeScript). host/engine.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * engine.js — the Engine class with the reference's public surface (engine/src/engine.ts:35-185,
 * 1419-1725): new Engine(canvas|null, options?), init(), loadModel(), loadAnimation(),
 * playAnimation(), stopAnimation(), rotateBones(), render(), runRenderLoop(), stopRenderLoop(),
 * getStats(), dispose(). What changes is what sits behind it: the WebGPU buffer uploads of
 * setupModelBuffers() (:1728-1832) and the per-frame writeBuffer + palette dispatch of
 * updateModelPose() (:2375-2402), and the skinning that the reference re-runs inside every vertex
 * shader invocation (:253-272), become calls into the HIP addon: upload once, then per frame
 * evaluatePose() on the CPU -> setPose(world matrices, morph weights) -> deform() on the MI355X.
 * Rendering (pipelines, bloom, camera, textures) and physics are out of scope and absent.
 *
 * Additions (SURVEY §8b): setMorphWeights(), getDeformed(), step(timeMs) — a deterministic clock
 * replacing performance.now()/window.setTimeout so a VMD can be stepped reproducibly — and
 * options { device, morphLayout: 'sparse'|'dense', realtime }.
 */
import { Quat, Vec3 } from './math'
import { PmxLoader } from './pmx-loader'
import { VMDLoader } from './vmd-loader'
import { VMDSampler } from './vmd-sampler'
import { requireAddon } from './addon'

import type { Bounds, DeformAddon, DeformContext, DeformedMesh, EngineOptions, EngineStats, PhysicsLike, Shard, Timer, Timing, VMDKeyFrames } from './types'
import type { Model } from './model'
let wallClock: () => number
try {
  const { performance } = require('perf_hooks')
  wallClock = () => performance.now()
} catch (e) {
  wallClock = () => Date.now()
}

class Engine {
  canvas: unknown
  ambient: number
  bloomIntensity: number
  rimLightIntensity: number
  cameraDistance: number
  cameraTarget: Vec3
  device: number
  devices: number[]
  deviceFK: boolean
  outline: boolean
  bounds: boolean
  gather: boolean | 'direct'
  morphLayout: 'sparse' | 'dense'
  realtime: boolean
  physics: PhysicsLike | null
  lastPhysicsTime: number | null
  native: DeformAddon | null
  deviceSampling: boolean
  instances: number
  animationOnDevice: VMDKeyFrames | null
  animationFor: Model | null
  framesInFlight: 1 | 2
  overrides: [Uint32Array, Float32Array, Uint32Array | null] | null
  autotune: boolean
  tuned: boolean
  ctx: DeformContext | null
  shards: Shard[]
  currentModel: Model | null
  animationFrames: VMDKeyFrames
  hasAnimation: boolean
  playingAnimation: boolean
  timers: Timer[]
  nextTimerId: number
  animationTimers: number[]
  breathingTimer: number | null
  breathingBaseRotations: Map<string, Quat>
  nowMs: number
  loopHandle: unknown
  renderLoopCallback: (() => void) | null
  stats: EngineStats
  frameTimeSamples: number[]
  frameTimeSum: number
  framesSinceLastUpdate: number
  lastFpsUpdate: number
  outPos: Float32Array | null
  outNrm: Float32Array | null
  outHull?: Float32Array
  edgeScale?: Float32Array
  modelDir?: string
  sampler?: VMDSampler
  samplerFor?: VMDKeyFrames
  constructor(canvas: unknown, options?: EngineOptions) {
    this.canvas = canvas || null // accepted for signature compatibility; nothing is drawn
    const o = options || {}
    // render-only options are accepted and kept so existing call sites keep working (engine.ts:145-154)
    this.ambient = o.ambient === undefined ? 1.0 : o.ambient
    this.bloomIntensity = o.bloomIntensity === undefined ? 0.12 : o.bloomIntensity
    this.rimLightIntensity = o.rimLightIntensity === undefined ? 0.45 : o.rimLightIntensity
    this.cameraDistance = o.cameraDistance === undefined ? 26.6 : o.cameraDistance
    this.cameraTarget = o.cameraTarget === undefined ? new Vec3(0, 12.5, 0) : o.cameraTarget
    this.device = o.device === undefined ? 0 : o.device
    // devices: [0,1,...]  one context per GPU in this process; the mesh is vertex-sharded across them (SURVEY §8e)
    this.devices = Array.isArray(o.devices) && o.devices.length > 0 ? o.devices.slice() : [this.device]
    this.deviceFK = o.deviceFK === true // forward kinematics on the GPU: upload local rotations instead of world matrices
    this.outline = o.outline === true // also produce the outline pass's inverted hull (engine.ts:458-461) every frame
    this.bounds = o.bounds === true // also reduce the deformed mesh's bounding box every frame
    // gather: true = RCCL all-gather of the deformed mesh after every frame (needs distinct GPUs);
    // 'direct' = every shard's kernel stores straight into GPU devices[0]'s buffer over xGMI (no collective)
    this.gather = o.gather === 'direct' ? 'direct' : o.gather === true
    this.morphLayout = o.morphLayout || 'sparse'
    this.realtime = o.realtime !== false // false: time only advances through step()
    // physics hand-off (engine.ts:2379-2381): an object with the reference Physics' step(dt, worldMats, inverseBind),
    // called between evaluatePose() and the world-matrix upload; it may overwrite world matrices in place. Physics
    // itself (Bullet via @fred3d/ammo) is out of scope — this is only the seam it plugs into.
    this.physics = o.physics || null
    this.lastPhysicsTime = null
    this.native = null
    // deviceSampling (needs deviceFK): seekFrame() sends one float — the frame — and the motion is sampled on the GPU
    this.deviceSampling = o.deviceSampling === true && o.deviceFK === true
    this.instances = 1
    this.animationOnDevice = null
    // framesInFlight: 2 = consecutive frames alternate between the context and a fork of it (rz_fork: the fork borrows the
    // static buffers, owns its stream, pose slots and outputs), so the tail of frame f overlaps the launch ramp of frame
    // f + 1 — what a WebGPU queue does with consecutive command buffers (engine.ts:2124-2136 submits one per frame).
    // getDeformed() / getOutlineHull() / getBounds() read the frame rendered last. Single GPU, no gather.
    this.framesInFlight = o.framesInFlight === 2 ? 2 : 1
    this.overrides = null
    this.autotune = o.autotune === true // search launch shapes once, on the first rendered frame (rz_autotune)
    this.tuned = false
    this.ctx = null // context of shard 0 (the only one on a single GPU)
    this.shards = [] // [{ ctx, begin, count }]
    this.currentModel = null
    this.animationFrames = []
    this.hasAnimation = false
    this.playingAnimation = false
    this.timers = [] // { due, fn, id, handle }
    this.nextTimerId = 1
    this.animationTimers = []
    this.breathingTimer = null
    this.breathingBaseRotations = new Map()
    this.nowMs = 0 // manual clock value
    this.loopHandle = null
    this.renderLoopCallback = null
    this.stats = { fps: 0, frameTime: 0, gpuMemory: 0, deformMs: 0, vertsPerSec: 0, hbmGBps: 0 }
    this.frameTimeSamples = []
    this.frameTimeSum = 0
    this.framesSinceLastUpdate = 0
    this.lastFpsUpdate = 0
    this.outPos = null
    this.outNrm = null
  }

  now(): number { return this.realtime ? wallClock() : this.nowMs }

  // ---- lifecycle ----
  /** engine.ts:157-185: acquire the device. Throws when the addon or an MI355X is not available. */
  async init(): Promise<void> {
    if (this.framesInFlight === 2 && (this.devices.length > 1 || this.gather)) throw new Error('framesInFlight: 2 needs a single GPU and no gather')
    this.native = requireAddon()
    this.shards = this.devices.map((d) => ({ ctx: this.native.create(d), begin: 0, count: 0, fork: null, last: null, flip: 0 }))
    this.ctx = this.shards[0].ctx
    this.lastFpsUpdate = this.now()
  }

  dispose(): void {
    this.stopRenderLoop()
    this.stopAnimation()
    this.stopBreathing()
    this.dropForks()
    for (const s of this.shards) this.native.destroy(s.ctx)
    this.shards = []
    this.ctx = null
  }

  // ---- frames in flight ----
  /** Forks borrow the lender's static buffers: they go before anything static is replaced, and before the lender. */
  dropForks(): void {
    for (const s of this.shards) {
      if (s.fork) { this.native.destroy(s.fork); s.fork = null }
      s.last = null
      s.flip = 0
    }
  }

  /** The context the NEXT frame of shard `s` runs on. */
  frameContext(s: Shard): DeformContext {
    if (this.framesInFlight !== 2 || (this.autotune && !this.tuned)) { s.last = s.ctx; return s.ctx }
    if (!s.fork) { // made after the launch-shape search, so that it inherits the tuned plan
      s.fork = this.native.fork(s.ctx)
      if (this.overrides) this.native.overrideWorld(s.fork, this.overrides[0], this.overrides[1], this.overrides[2])
    }
    s.flip ^= 1
    s.last = s.flip ? s.fork : s.ctx
    return s.last
  }

  // ---- timers (window.setTimeout replacement that also works on a manual clock) ----
  setTimer(fn: () => void, delayMs: number): number {
    const t = { due: this.now() + delayMs, fn, id: this.nextTimerId++, handle: null }
    if (this.realtime) t.handle = setTimeout(() => { this.dropTimer(t.id); fn() }, delayMs)
    this.timers.push(t)
    return t.id
  }

  dropTimer(id: number): void {
    const i = this.timers.findIndex((t) => t.id === id)
    if (i >= 0) { if (this.timers[i].handle) clearTimeout(this.timers[i].handle); this.timers.splice(i, 1) }
  }

  fireDueTimers(): void {
    for (;;) { // earliest first; a callback may schedule more
      let best = -1
      for (let i = 0; i < this.timers.length; i++) {
        if (this.timers[i].due <= this.nowMs && (best < 0 || this.timers[i].due < this.timers[best].due)) best = i
      }
      if (best < 0) return
      const t = this.timers.splice(best, 1)[0]
      t.fn()
    }
  }

  // ---- model ----
  /** engine.ts:1704-1721 */
  async loadModel(path: string): Promise<void> {
    const parts = path.split('/')
    parts.pop()
    this.modelDir = parts.join('/') + '/'
    const model = await PmxLoader.load(path)
    await this.setupModelBuffers(model)
  }

  /** engine.ts:1728-1832: one-off static upload (vertex / joints / weights / inverse bind [+ morph targets]). */
  async setupModelBuffers(model: Model): Promise<void> {
    if (!this.ctx) throw new Error('Engine.init() has not been called')
    this.dropForks()
    this.overrides = null // they name bones of the previous model; the library drops its copy with the skeleton (rz_upload_skeleton)
    this.currentModel = model
    model.setClock(() => this.now())
    const n = this.native, skinning = model.getSkinning(), skeleton = model.getSkeleton()
    const morphs = model.getMorphs()
    const V = model.getVertexCount()
    const G = this.shards.length
    for (let r = 0; r < G; r++) {
      const s = this.shards[r]
      const range = n.shardRange(V, G, r)
      s.begin = range[0]; s.count = range[1]
      if (s.count === 0) continue
      const b = s.begin, e = s.begin + s.count
      // static data is cut once; subarray() views are zero-copy into the addon
      n.uploadMesh(s.ctx, model.getVertices().subarray(b * 8, e * 8), skinning.joints.subarray(b * 4, e * 4), skinning.weights.subarray(b * 4, e * 4))
      n.uploadSkeleton(s.ctx, skeleton.inverseBindMatrices)
      if (this.deviceFK) {
        const B = skeleton.bones.length
        const parents = new Int32Array(B), bind = new Float32Array(B * 3), ap = new Int32Array(B).fill(-1), ar = new Float32Array(B).fill(1)
        const am = new Uint8Array(B)
        skeleton.bones.forEach((bn, i) => {
          parents[i] = bn.parentIndex
          bind.set(bn.bindTranslation, i * 3)
          if (bn.appendRotate && bn.appendParentIndex !== undefined && bn.appendParentIndex !== null) {
            ap[i] = bn.appendParentIndex
            ar[i] = bn.appendRatio === undefined || bn.appendRatio === null ? 1 : bn.appendRatio
            am[i] = bn.appendMove ? 1 : 0
          }
        })
        n.uploadSkeletonTopology(s.ctx, parents, bind, ap, ar, am)
      }
      if (morphs && morphs.names.length > 0) {
        const M = morphs.names.length
        if (this.morphLayout === 'dense') {
          const dense = new Float32Array(M * s.count * 3)
          for (let m = 0; m < M; m++) {
            for (let k = morphs.offsets[m]; k < morphs.offsets[m + 1]; k++) {
              const v = morphs.vertexIndex[k]
              if (v < b || v >= e) continue
              const d = (m * s.count + (v - b)) * 3
              dense[d] += morphs.deltas[k * 3]; dense[d + 1] += morphs.deltas[k * 3 + 1]; dense[d + 2] += morphs.deltas[k * 3 + 2]
            }
          }
          n.uploadMorphsDense(s.ctx, M, dense)
        } else if (G === 1) {
          n.uploadMorphsSparse(s.ctx, morphs.offsets, morphs.vertexIndex, morphs.deltas)
        } else { // re-base the sparse entries that fall inside this shard
          const off = new Uint32Array(M + 1), vi = [], dl = []
          for (let m = 0; m < M; m++) {
            for (let k = morphs.offsets[m]; k < morphs.offsets[m + 1]; k++) {
              const v = morphs.vertexIndex[k]
              if (v >= b && v < e) { vi.push(v - b); dl.push(morphs.deltas[k * 3], morphs.deltas[k * 3 + 1], morphs.deltas[k * 3 + 2]) }
            }
            off[m + 1] = vi.length
          }
          n.uploadMorphsSparse(s.ctx, off, Uint32Array.from(vi), Float32Array.from(dl))
        }
        // PMX bone morphs: with the hierarchy solved on the GPU they are folded there (host FK: Model.posedLocals())
        const be = morphs.boneEntries
        if (this.deviceFK && be && be.morph.length > 0) n.uploadBoneMorphs(s.ctx, be.morph, be.bone, be.translation, be.rotation)
      }
    }
    if (this.outline) {
      // per-vertex edge size = edgeSize of the material whose index range draws the vertex (edge flag 0x10), else 0
      const edge = new Float32Array(V)
      const indices = model.getIndices()
      let first = 0
      for (const mat of model.getMaterials()) {
        const size = (mat.edgeFlag & 0x10) !== 0 ? mat.edgeSize : 0
        for (let k = first; k < first + mat.vertexCount && k < indices.length; k++) edge[indices[k]] = size
        first += mat.vertexCount
      }
      this.edgeScale = edge
      for (const s of this.shards) if (s.count > 0) n.uploadEdgeScale(s.ctx, edge.subarray(s.begin, s.begin + s.count))
      this.outHull = new Float32Array(V * 3)
    }
    if (this.bounds) for (const s of this.shards) if (s.count > 0) n.enableAabb(s.ctx, true)
    if (this.gather === 'direct' && G > 1) n.gatherDirect(this.shards.map((s) => s.ctx), V, 0)
    else if (this.gather && G > 1) n.commInitAll(this.shards.map((s) => s.ctx), V)
    this.tuned = false
    this.outPos = new Float32Array(V * 3)
    this.outNrm = new Float32Array(V * 3)
    this.stats.gpuMemory = Math.round(((V * 60 + skeleton.bones.length * 176 +
      (morphs ? morphs.vertexIndex.length * 16 + V * 4 : 0)) / 1024 / 1024) * 100) / 100
  }

  rotateBones(bones: string[], rotations: Quat[], durationMs?: number): void {
    if (this.currentModel) this.currentModel.rotateBones(bones, rotations, durationMs)
  }

  /** uv with the UV morphs (PMX type 3) applied, V x 2 — host-side: UVs never pass through the deformation kernel (engine.ts:273) */
  getMorphedUVs(): Float32Array { return this.currentModel ? this.currentModel.getMorphedUVs() : new Float32Array(0) }

  setMorphWeights(namesOrIndices: Array<string | number>, weights: ArrayLike<number>): void {
    if (this.currentModel) this.currentModel.setMorphWeights(namesOrIndices, weights)
  }

  // ---- animation ("VMD step", engine.ts:1419-1662) ----
  async loadAnimation(path: string): Promise<void> {
    this.animationFrames = await VMDLoader.load(path)
    this.hasAnimation = true
  }

  playAnimation(options?: { breathBones?: string[]; breathDuration?: number; breathRanges?: Record<string, number> }): void {
    if (this.animationFrames.length === 0) return
    this.stopAnimation()
    this.stopBreathing()
    this.playingAnimation = true
    const opt = options || {}
    let breathBones = []
    let breathRanges
    const enableBreath = opt.breathBones !== undefined && opt.breathBones !== null
    if (enableBreath) {
      if (Array.isArray(opt.breathBones)) breathBones = opt.breathBones
      else { breathBones = Object.keys(opt.breathBones); breathRanges = opt.breathBones }
    }
    const breathDuration = opt.breathDuration === undefined || opt.breathDuration === null ? 4000 : opt.breathDuration

    // per-bone key lists in time order
    const byBone = new Map()
    for (const kf of this.animationFrames) {
      for (const bf of kf.boneFrames) {
        if (!byBone.has(bf.boneName)) byBone.set(bf.boneName, [])
        byBone.get(bf.boneName).push({ boneName: bf.boneName, time: kf.time, rotation: bf.rotation })
      }
    }
    for (const keys of byBone.values()) keys.sort((a, b) => a.time - b.time)

    if (this.currentModel) {
      // time-0 keys apply instantly; every bone without one snaps to identity (:1474-1505)
      const names0 = [], rots0 = [], has0 = new Set()
      for (const [name, keys] of byBone.entries()) {
        if (keys.length > 0 && keys[0].time === 0) { names0.push(name); rots0.push(keys[0].rotation); has0.add(name) }
      }
      if (names0.length > 0) this.rotateBones(names0, rots0, 0)
      const reset = this.currentModel.getSkeleton().bones.map((b) => b.name).filter((n) => !has0.has(n))
      if (reset.length > 0) this.rotateBones(reset, reset.map(() => new Quat(0, 0, 0, 1)), 0)
    }

    // later keys: a tween from the previous key's time lasting until this key's time (:1527-1553)
    for (const keys of byBone.values()) {
      for (let i = 0; i < keys.length; i++) {
        const k = keys[i]
        if (k.time === 0) continue
        const prev = i > 0 ? keys[i - 1] : null
        const durationMs = (prev ? k.time - prev.time : k.time) * 1000
        const delayMs = (prev ? prev.time : 0) * 1000
        if (delayMs <= 0) this.rotateBones([k.boneName], [k.rotation], durationMs)
        else this.animationTimers.push(this.setTimer(() => this.rotateBones([k.boneName], [k.rotation], durationMs), delayMs))
      }
    }

    // morph keys (no reference counterpart: its VMD loader never reads the block): step to each key's weight
    const mf = this.animationFrames.morphFrames || []
    for (const f of mf) {
      const apply = () => this.setMorphWeights([f.morphName], [f.weight])
      if (f.time <= 0) apply(); else this.animationTimers.push(this.setTimer(apply, f.time * 1000))
    }

    if (enableBreath && this.currentModel) {
      let maxTime = 0
      for (const kf of this.animationFrames) if (kf.time > maxTime) maxTime = kf.time
      const last = new Map()
      for (const bone of breathBones) {
        const keys = byBone.get(bone)
        if (keys && keys.length > 0) last.set(bone, keys[keys.length - 1].rotation)
      }
      this.breathingTimer = this.setTimer(() => this.startBreathing(breathBones, last, breathRanges, breathDuration), maxTime * 1000 + 200)
    }
  }

  stopAnimation(): void {
    for (const id of this.animationTimers) this.dropTimer(id)
    this.animationTimers = []
    this.playingAnimation = false
  }

  stopBreathing(): void {
    if (this.breathingTimer !== null) { this.dropTimer(this.breathingTimer); this.breathingTimer = null }
    this.breathingBaseRotations.clear()
  }

  startBreathing(bones: string[], baseRotations: Map<string, Quat>, rotationRanges?: Record<string, number>, durationMs?: number): void {
    if (!this.currentModel) return
    for (const b of bones) if (baseRotations.has(b)) this.breathingBaseRotations.set(b, baseRotations.get(b))
    const half = (durationMs === undefined ? 4000 : durationMs) / 2
    const swing = (inhale) => {
      if (!this.currentModel) return
      const names = [], quats = []
      for (const b of bones) {
        const base = this.breathingBaseRotations.get(b)
        if (!base) continue
        const range = rotationRanges && rotationRanges[b] !== undefined && rotationRanges[b] !== null ? rotationRanges[b] : 0.02
        names.push(b)
        quats.push(base.multiply(Quat.fromEuler(inhale ? range : -range, 0, 0)))
      }
      if (names.length > 0) this.rotateBones(names, quats, half)
      this.breathingTimer = this.setTimer(() => swing(!inhale), half)
    }
    swing(false)
  }

  // ---- per frame ----
  /** engine.ts:2124-2136 + 2375-2402 minus the draw calls: pose on the CPU, deformation on the GPU. */
  render(): void {
    if (!this.currentModel || !this.ctx) return
    const t0 = wallClock()
    const model = this.currentModel
    const gpuFK = this.deviceFK
    // VMD bone translations (seekFrame) travel with the rotations; the reference itself never writes localTranslations
    const tra = gpuFK && model.applyLocalTranslations ? model.runtimeSkeleton.localTranslations : null
    if (gpuFK) model.updateRotationTweens() // tweens stay on the host; the hierarchy solve moves to the GPU
    else model.evaluatePose()
    if (this.physics && !gpuFK) { // updateModelPose(): physics.step(deltaTime, worldMats, inverseBind) mutates in place
      const now = this.now()
      const dt = this.lastPhysicsTime === null ? 0 : (now - this.lastPhysicsTime) / 1000
      this.lastPhysicsTime = now
      this.physics.step(dt, model.getBoneWorldMatrices(), model.getBoneInverseBindMatrices())
    }
    const mw = model.getMorphCount() > 0 ? model.getEffectiveMorphWeights() : null
    // per-frame inputs are replicated to every shard (16-22 KB); launches are asynchronous, so the GPUs run concurrently
    for (const s of this.shards) {
      if (s.count === 0) continue
      const c = this.frameContext(s)
      if (gpuFK) this.native.setPoseLocal(c, model.runtimeSkeleton.localRotations, mw, tra)
      else this.native.setPose(c, model.getBoneWorldMatrices(), mw)
      this.native.deform(c)
    }
    if (this.autotune && !this.tuned) { // the first frame supplied a pose: time the candidate launch shapes once per shard
      for (const s of this.shards) if (s.count > 0) this.native.autotune(s.ctx, 0)
      this.tuned = true
    }
    if (this.gather === 'direct' && this.shards.length > 1) this.native.gatherFence(this.ctx)
    else if (this.gather && this.shards.length > 1) this.native.allgatherAll(this.shards.map((s) => s.ctx), true)
    this.updateStats(wallClock() - t0)
  }

  /**
   * Frame-indexed playback (MMD semantics: Bezier-warped slerp / lerp between keys, bone translation, linear morph
   * keys) of the loaded animation: pose the model at `frame` (30 fps, fractional allowed) and deform one frame.
   * Independent of playAnimation()'s wall-clock tweens, which mirror the reference.
   */
  seekFrame(frame: number | ArrayLike<number>): void {
    if (!this.currentModel) return
    if (!this.sampler || this.samplerFor !== this.animationFrames) {
      this.sampler = new VMDSampler(this.animationFrames)
      this.samplerFor = this.animationFrames
    }
    if (this.deviceSampling) return this.seekFrameOnDevice(frame)
    this.currentModel.applySampledFrame(this.sampler, frame)
    this.render()
  }

  /** seekFrame with { deviceFK, deviceSampling }: upload the flattened motion once, then one float per frame. */
  seekFrameOnDevice(frame: number | ArrayLike<number>): void {
    const model = this.currentModel
    if (this.animationOnDevice !== this.animationFrames || this.animationFor !== model) {
      const flat = this.sampler.flatten(model.runtimeSkeleton.nameIndex, model.getMorphCount() > 0 ? model.getMorphs() : null)
      this.dropForks() // the motion is static data
      for (const s of this.shards) if (s.count > 0) this.native.uploadAnimation(s.ctx, flat)
      this.animationOnDevice = this.animationFrames
      this.animationFor = model
    }
    const t0 = wallClock()
    // a crowd (setInstanceCount) takes one frame per instance; a single number poses every instance at that frame
    const f = typeof frame === 'number' ? new Float32Array(this.instances).fill(frame) : Float32Array.from(frame)
    if (f.length !== this.instances) throw new Error('seekFrame: ' + f.length + ' frames for ' + this.instances + ' instances')
    for (const s of this.shards) {
      if (s.count === 0) continue
      const c = this.frameContext(s)
      this.native.setPoseSampled(c, f)
      this.native.deform(c)
    }
    if (this.autotune && !this.tuned) { // as in render(): the first frame supplied a pose
      for (const s of this.shards) if (s.count > 0) this.native.autotune(s.ctx, 0)
      this.tuned = true
    }
    if (this.gather === 'direct' && this.shards.length > 1) this.native.gatherFence(this.ctx)
    else if (this.gather && this.shards.length > 1) this.native.allgatherAll(this.shards.map((s) => s.ctx), true)
    this.updateStats(wallClock() - t0)
  }

  /**
   * Physics hand-off when the hierarchy is solved on the GPU ({ deviceFK }): the world matrices of physics-driven bones
   * (what physics.ts:715-751 writes with boneWorldMatrices.set(values, boneIndex * 16)) replace the solved ones after
   * the hierarchy solve in every following frame, until the next call; an empty list clears. Children keep the
   * matrices solved from the un-overridden parent, as in the reference. `instances` (optional) names the crowd member
   * of each entry. On the host-FK path use the { physics } option instead — there the host owns the world matrices.
   */
  setBoneWorldOverrides(boneIndices: ArrayLike<number>, worldMatrices: ArrayLike<number>, instances?: ArrayLike<number>): void {
    if (!this.ctx) throw new Error('Engine.init() has not been called')
    if (!this.deviceFK) throw new Error('setBoneWorldOverrides needs new Engine(canvas, { deviceFK: true }); with host FK pass { physics }')
    const b = boneIndices && boneIndices.length ? Uint32Array.from(boneIndices) : null
    const w = b ? (worldMatrices instanceof Float32Array ? worldMatrices : Float32Array.from(worldMatrices)) : null
    const i = b && instances ? Uint32Array.from(instances) : null
    this.overrides = b ? [b, w, i] : null
    for (const s of this.shards) {
      if (s.count === 0) continue
      this.native.overrideWorld(s.ctx, b, w, i)
      if (s.fork) this.native.overrideWorld(s.fork, b, w, i)
    }
  }

  /** Deterministic stepping: move the clock to timeMs, fire the timers that came due, render one frame. */
  step(timeMs: number): void {
    if (this.realtime) throw new Error('step() needs new Engine(canvas, { realtime: false })')
    this.nowMs = timeMs
    this.fireDueTimers()
    this.render()
  }

  /**
   * A crowd of `n` independently posed copies of the loaded model (BASELINE config 4; the reference draws one model).
   * Needs { deviceFK, deviceSampling } — every instance is posed on the GPU at its own frame of the loaded motion,
   * seekFrame([f0, f1, ...]) — and a single GPU (instancing and vertex sharding are exclusive).
   */
  setInstanceCount(n: number): void {
    if (!this.ctx || !this.currentModel) throw new Error('no model loaded')
    if (!this.deviceSampling) throw new Error('setInstanceCount needs new Engine(canvas, { deviceFK: true, deviceSampling: true })')
    if (this.shards.length > 1) throw new Error('instancing and vertex sharding are exclusive')
    if (this.outline || this.bounds) throw new Error('the outline hull and bounds are single-instance consumers')
    this.dropForks() // a fork takes its instance count from the lender when it is made
    if (n !== this.instances) this.overrides = null // (instance, bone) pairs of the old crowd: rz_set_instances dropped them on the lender too
    this.native.setInstances(this.ctx, n)
    this.instances = n
    this.tuned = false
  }

  /** Blocking readback of the deformed mesh (the values the reference's vs() only ever feeds the rasteriser). */
  getDeformed(instance?: number): DeformedMesh {
    if (!this.ctx || !this.currentModel) throw new Error('no model loaded')
    if (instance !== undefined && instance !== 0) {
      if (!(instance > 0 && instance < this.instances)) throw new Error('instance ' + instance + ' out of range')
      const V = this.currentModel.getVertexCount()
      const pos = new Float32Array(V * 3), nrm = new Float32Array(V * 3)
      this.native.read(this.shards[0].last || this.ctx, instance, 0, V, pos, nrm)
      return { positions: pos, normals: nrm }
    }
    if (this.gather && this.shards.length > 1) { // shard 0's GPU holds the whole mesh (all-gather or peer-direct stores)
      this.native.readGathered(this.ctx, 0, this.currentModel.getVertexCount(), this.outPos, this.outNrm)
      return { positions: this.outPos, normals: this.outNrm }
    }
    for (const s of this.shards) {
      if (s.count === 0) continue
      this.native.read(s.last || s.ctx, 0, 0, s.count, this.outPos.subarray(s.begin * 3, (s.begin + s.count) * 3), this.outNrm.subarray(s.begin * 3, (s.begin + s.count) * 3))
    }
    return { positions: this.outPos, normals: this.outNrm }
  }

  /** Inverted-hull positions the outline pipeline draws: worldPos + worldNormal * edgeSize * 0.01 (needs { outline: true }). */
  getOutlineHull(): Float32Array {
    if (!this.outline) throw new Error('new Engine(canvas, { outline: true }) enables the outline hull')
    for (const s of this.shards) if (s.count > 0) this.native.readHull(s.last || s.ctx, 0, 0, s.count, this.outHull.subarray(s.begin * 3, (s.begin + s.count) * 3))
    return this.outHull
  }

  /** Axis-aligned bounds { min: [x,y,z], max: [x,y,z] } of the last deformed frame (needs { bounds: true }). */
  getBounds(): Bounds {
    if (!this.bounds) throw new Error('new Engine(canvas, { bounds: true }) enables the bounding-box reduction')
    const b = new Float32Array(6)
    const min = [Infinity, Infinity, Infinity], max = [-Infinity, -Infinity, -Infinity]
    for (const s of this.shards) {
      if (s.count === 0) continue
      this.native.readAabb(s.last || s.ctx, 0, b)
      for (let k = 0; k < 3; k++) { min[k] = Math.min(min[k], b[k]); max[k] = Math.max(max[k], b[3 + k]) }
    }
    return { min, max }
  }

  runRenderLoop(callback?: () => void): void {
    this.renderLoopCallback = callback || null
    const tick = () => {
      this.render()
      if (this.renderLoopCallback) this.renderLoopCallback()
      this.loopHandle = setTimeout(tick, 0) // no requestAnimationFrame in Node: free-running
    }
    this.loopHandle = setTimeout(tick, 0)
  }

  stopRenderLoop(): void {
    if (this.loopHandle !== null) { clearTimeout(this.loopHandle); this.loopHandle = null }
    this.renderLoopCallback = null
  }

  /** engine.ts:2423-2445 (60-sample moving average, 1 Hz fps) + deformation figures. */
  updateStats(frameTime: number): void {
    this.frameTimeSamples.push(frameTime)
    this.frameTimeSum += frameTime
    if (this.frameTimeSamples.length > 60) this.frameTimeSum -= this.frameTimeSamples.shift()
    this.stats.frameTime = Math.round((this.frameTimeSum / this.frameTimeSamples.length) * 100) / 100
    const now = wallClock()
    this.framesSinceLastUpdate++
    const elapsed = now - this.lastFpsUpdate
    if (elapsed >= 1000) {
      this.stats.fps = Math.round((this.framesSinceLastUpdate / elapsed) * 1000)
      this.framesSinceLastUpdate = 0
      this.lastFpsUpdate = now
    }
  }

  /** Time `frames` back-to-back frames of the current pose on the GPU (HIP events) and fold them into getStats(). */
  measure(frames?: number): Timing {
    const t = this.native.timeFrames(this.ctx, frames || 100)
    this.stats.deformMs = t.frameMs
    this.stats.vertsPerSec = t.vertsPerFrame / (t.frameMs * 1e-3)
    this.stats.hbmGBps = t.algorithmicBytesPerFrame / (t.deformKernelMs * 1e-3) / 1e9
    return t
  }

  getStats(): EngineStats { return Object.assign({}, this.stats) }
}

export { Engine }
