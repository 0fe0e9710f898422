'use strict'
/*
 * Source: host/src/vmd-sampler.ts (TypeScript). host/vmd-sampler.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * vmd-sampler.js — frame-indexed VMD sampling (SURVEY §8f rank 2). No reference counterpart: the reference's
 * loader drops position + the 64 interpolation bytes (engine/src/vmd-loader.ts:129-140), never reads the morph
 * block, and its player replays keys as wall-clock eased tweens (engine.ts:1527-1553). This sampler evaluates a
 * motion at an arbitrary (fractional) frame the way MMD defines it:
 *   rotation   slerp between the surrounding keys, parameter warped by the key's R Bezier curve
 *   position   per-axis lerp, each axis warped by its own X / Y / Z Bezier curve
 *   morph      linear between the surrounding morph keys
 * Interpolation block layout (first 16 of the 64 bytes; values / 127):
 *   [X_x1, Y_x1, Z_x1, R_x1,  X_y1, Y_y1, Z_y1, R_y1,  X_x2, Y_x2, Z_x2, R_x2,  X_y2, Y_y2, Z_y2, R_y2]
 * i.e. curve c in {X,Y,Z,R} = cubic Bezier through (0,0), (x1,y1), (x2,y2), (1,1), stored on the LATER key.
 */
import { kernels } from './math'

// y(x) of the cubic Bezier (0,0) (x1,y1) (x2,y2) (1,1): solve x(t) = x by bisection-refined Newton, return y(t)
import type { BoneFrame, BoneSample, FlatMotion, MorphFrame, MorphSet, Triple, VMDKeyFrames } from './types'
function bezier(x: number, x1: number, y1: number, x2: number, y2: number): number {
  if (x <= 0) return 0
  if (x >= 1) return 1
  if (x1 === y1 && x2 === y2) return x // the default 20,20,107,107 curve is the identity
  let lo = 0, hi = 1, t = x
  for (let i = 0; i < 32; i++) {
    const s = 1 - t
    const fx = 3 * s * s * t * x1 + 3 * s * t * t * x2 + t * t * t - x
    if (Math.abs(fx) < 1e-9) break
    if (fx > 0) hi = t; else lo = t
    const dfx = 3 * s * s * x1 + 6 * s * t * (x2 - x1) + 3 * t * t * (1 - x2)
    const tn = dfx !== 0 ? t - fx / dfx : (lo + hi) / 2
    t = tn > lo && tn < hi ? tn : (lo + hi) / 2
  }
  const s = 1 - t
  return 3 * s * s * t * y1 + 3 * s * t * t * y2 + t * t * t
}

class VMDSampler {
  bones: Map<string, BoneFrame[]>
  morphs: Map<string, MorphFrame[]>
  lastFrame: number
  _q: number[]
  /** @param keyFrames result of VMDLoader.load / loadFromBuffer (array + .morphFrames) */
  constructor(keyFrames: VMDKeyFrames) {
    this.bones = new Map() // name -> keys sorted by frame
    for (const kf of keyFrames) {
      for (const b of kf.boneFrames) {
        if (!this.bones.has(b.boneName)) this.bones.set(b.boneName, [])
        this.bones.get(b.boneName).push(b)
      }
    }
    for (const keys of this.bones.values()) keys.sort((a, b) => a.frame - b.frame)
    this.morphs = new Map()
    for (const m of keyFrames.morphFrames || []) {
      if (!this.morphs.has(m.morphName)) this.morphs.set(m.morphName, [])
      this.morphs.get(m.morphName).push(m)
    }
    for (const keys of this.morphs.values()) keys.sort((a, b) => a.frame - b.frame)
    this.lastFrame = 0
    for (const keys of this.bones.values()) this.lastFrame = Math.max(this.lastFrame, keys[keys.length - 1].frame)
    for (const keys of this.morphs.values()) this.lastFrame = Math.max(this.lastFrame, keys[keys.length - 1].frame)
    this._q = [0, 0, 0, 1]
  }

  static span(keys: Array<{ frame: number }>, frame: number): Triple { // index i with keys[i].frame <= frame < keys[i+1].frame (clamped)
    let lo = 0, hi = keys.length - 1
    if (frame <= keys[0].frame) return [0, 0, 0]
    if (frame >= keys[hi].frame) return [hi, hi, 0]
    while (hi - lo > 1) { const mid = (lo + hi) >> 1; if (keys[mid].frame <= frame) lo = mid; else hi = mid }
    return [lo, hi, (frame - keys[lo].frame) / (keys[hi].frame - keys[lo].frame)]
  }

  /** -> { rotation: [x,y,z,w], position: [x,y,z] } of one bone at `frame`, or null when the motion does not key it */
  sampleBone(name: string, frame: number): BoneSample | null {
    const keys = this.bones.get(name)
    if (!keys) return null
    const [i0, i1, x] = VMDSampler.span(keys, frame)
    const a = keys[i0], b = keys[i1]
    if (i0 === i1) return { rotation: [a.rotation.x, a.rotation.y, a.rotation.z, a.rotation.w], position: [a.position.x, a.position.y, a.position.z] }
    const ip = b.interpolation
    const curve = (c) => (ip ? bezier(x, ip[c] / 127, ip[c + 4] / 127, ip[c + 8] / 127, ip[c + 12] / 127) : x)
    const q = kernels.slerpInto(this._q, a.rotation.x, a.rotation.y, a.rotation.z, a.rotation.w,
      b.rotation.x, b.rotation.y, b.rotation.z, b.rotation.w, curve(3))
    const tx = curve(0), ty = curve(1), tz = curve(2)
    return {
      rotation: [q[0], q[1], q[2], q[3]],
      position: [a.position.x + (b.position.x - a.position.x) * tx, a.position.y + (b.position.y - a.position.y) * ty,
        a.position.z + (b.position.z - a.position.z) * tz],
    }
  }

  sampleMorph(name: string, frame: number): number | null {
    const keys = this.morphs.get(name)
    if (!keys) return null
    const [i0, i1, x] = VMDSampler.span(keys, frame)
    return keys[i0].weight + (keys[i1].weight - keys[i0].weight) * x
  }

  /**
   * Flatten the motion for the device sampler (rz_upload_animation): one track per bone of `boneNameIndex` the motion
   * keys, one track per keyed morph, and for every morph of the model the tracks that feed its GPU weight in the order
   * Model.getEffectiveMorphWeights adds them — its own track (vertex morphs only), then the group morphs that list it,
   * ascending. `morphs` is Model.getMorphs() ({ names, types, groups }) or null.
   */
  flatten(boneNameIndex: Record<string, number>, morphs: Pick<MorphSet, 'names' | 'types' | 'groups'> | null): FlatMotion {
    const tracks = []
    for (const [name, keys] of this.bones) {
      const b = boneNameIndex[name]
      if (b !== undefined) tracks.push([b, keys])
    }
    let K = 0
    for (const t of tracks) K += t[1].length
    const out = {
      trackBone: new Int32Array(tracks.length), keyOff: new Uint32Array(tracks.length + 1), keyFrame: new Float32Array(K),
      keyRot: new Float32Array(K * 4), keyPos: new Float32Array(K * 3), keyInterp: new Uint8Array(K * 16),
    }
    let k = 0
    tracks.forEach(([b, keys], t) => {
      out.trackBone[t] = b
      out.keyOff[t] = k
      for (const key of keys) {
        out.keyFrame[k] = key.frame
        out.keyRot.set([key.rotation.x, key.rotation.y, key.rotation.z, key.rotation.w], k * 4)
        out.keyPos.set([key.position.x, key.position.y, key.position.z], k * 3)
        if (key.interpolation) for (let i = 0; i < 16; i++) out.keyInterp[k * 16 + i] = key.interpolation[i]
        else out.keyInterp.set([20, 20, 20, 20, 20, 20, 20, 20, 107, 107, 107, 107, 107, 107, 107, 107], k * 16) // identity curves
        k++
      }
    })
    out.keyOff[tracks.length] = k
    if (!morphs || morphs.names.length === 0) return out
    const M = morphs.names.length
    const trackOf = new Int32Array(M).fill(-1)
    const mtracks = []
    morphs.names.forEach((name, i) => {
      if (this.morphs.has(name) && trackOf[i] < 0) { trackOf[i] = mtracks.length; mtracks.push(this.morphs.get(name)) }
    })
    let Km = 0
    for (const keys of mtracks) Km += keys.length
    out.mkeyOff = new Uint32Array(mtracks.length + 1); out.mkeyFrame = new Float32Array(Km); out.mkeyWeight = new Float32Array(Km)
    k = 0
    mtracks.forEach((keys, t) => {
      out.mkeyOff[t] = k
      for (const key of keys) { out.mkeyFrame[k] = key.frame; out.mkeyWeight[k] = key.weight; k++ }
    })
    out.mkeyOff[mtracks.length] = k
    const feeds = []
    out.feedOff = new Uint32Array(M + 1)
    for (let i = 0; i < M; i++) {
      out.feedOff[i] = feeds.length
      if (morphs.types[i] !== 1 && morphs.types[i] !== 2) continue // vertex morphs carry deltas, bone morphs move bones; the rest is not on the GPU path
      if (trackOf[i] >= 0) feeds.push([trackOf[i], 1])
      for (let g = 0; g < M; g++) {
        if (morphs.types[g] !== 0 || trackOf[g] < 0 || !morphs.groups[g]) continue
        for (const [child, ratio] of morphs.groups[g]) if (child === i) feeds.push([trackOf[g], ratio])
      }
    }
    out.feedOff[M] = feeds.length
    out.feedTrack = Int32Array.from(feeds.map((f) => f[0])); out.feedRatio = Float32Array.from(feeds.map((f) => f[1]))
    return out
  }

  boneNames(): string[] { return Array.from(this.bones.keys()) }
  morphNames(): string[] { return Array.from(this.morphs.keys()) }
}

export { VMDSampler, bezier }
