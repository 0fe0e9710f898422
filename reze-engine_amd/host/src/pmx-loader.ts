'use strict'
/*
 * Source: host/src/pmx-loader.ts (TypeScript). host/pmx-loader.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * pmx-loader.js — PMX 2.x parser producing the typed arrays the deformation path consumes.
 *
 * Mirrors the reference's PmxLoader (engine/src/pmx-loader.ts): same static entry point
 * `PmxLoader.load(path)` returning a Model, same field order (header :51-96, vertices :98-189,
 * indices :193-202, textures :204-220, materials :222-309, bones :311-448), the same skinning
 * encode (BDEF1/2/4, SDEF-as-BDEF2, QDEF-as-BDEF4 -> 4 x u16 joints + 4 x u8 weights summing to
 * 255, :136-179 and the clamp/renormalise pass of :855-951) and the same translation-only inverse
 * bind (:791-824).
 *
 * Differences, all additions: the reference SKIPS the morph section (:450-553); this loader
 * parses it (type 0 group: morphIndex + ratio, :479-482; type 1 vertex: vertexIndex + vec3,
 * :483-488) into the sparse morph-major arrays rz_upload_morphs_sparse() takes, and records the
 * other types' names so VMD morph keys can be resolved. `fetch` is replaced by fs (Node) with a
 * `loadFromBuffer` entry for tests. Rigid bodies / joints are parsed for API completeness but the
 * physics that consumes them is out of scope here.
 */
import * as fs from 'fs'
import { TextDecoder } from 'util'
import { Model } from './model'
import { Mat4, Vec3 } from './math'

import type { Bone, Material, MorphSet, Texture, Triple } from './types'
interface PmxHeader { version: number; encoding: number; extraVec4: number; vertexIndexSize: number; textureIndexSize: number; materialIndexSize: number; boneIndexSize: number; morphIndexSize: number; rigidBodyIndexSize: number }
interface Geometry { count: number; vertexData: Float32Array; joints: Uint16Array; weights: Uint8Array }
class Cursor {
  view: DataView
  bytes: Uint8Array
  pos: number
  end: number
  constructor(arrayBuffer: ArrayBuffer, byteOffset: number, byteLength: number) {
    this.view = new DataView(arrayBuffer, byteOffset || 0, byteLength)
    this.bytes = new Uint8Array(arrayBuffer, byteOffset || 0, byteLength)
    this.pos = 0
    this.end = this.view.byteLength
  }
  need(n: number): void {
    if (this.pos + n > this.end) throw new RangeError('Offset ' + this.pos + ' + ' + n + ' exceeds buffer bounds ' + this.end)
  }
  u8(): number { this.need(1); return this.view.getUint8(this.pos++) }
  i8(): number { this.need(1); return this.view.getInt8(this.pos++) }
  u16(): number { this.need(2); const v = this.view.getUint16(this.pos, true); this.pos += 2; return v }
  i16(): number { this.need(2); const v = this.view.getInt16(this.pos, true); this.pos += 2; return v }
  i32(): number { this.need(4); const v = this.view.getInt32(this.pos, true); this.pos += 4; return v }
  f32(): number { this.need(4); const v = this.view.getFloat32(this.pos, true); this.pos += 4; return v }
  skip(n: number): void { this.need(n); this.pos += n }
  // vertex index: 1 -> uint8, 2 -> uint16, 4 -> int32   (pmx-loader.ts:981-990)
  vertexIndex(size: number): number { return size === 1 ? this.u8() : size === 2 ? this.u16() : this.i32() }
  // every other index is signed: 1 -> int8, 2 -> int16, 4 -> int32   (:992-1005)
  index(size: number): number { return size === 1 ? this.i8() : size === 2 ? this.i16() : this.i32() }
  vec3(): Triple { return [this.f32(), this.f32(), this.f32()] }
}

function clamp(v: number, lo: number, hi: number): number { return Math.max(lo, Math.min(hi, v)) }

class PmxLoader {
  cur: Cursor
  h: PmxHeader | null
  decoder: TextDecoder
  modelName: string
  constructor(buffer: ArrayBuffer | Uint8Array) {
    // accept ArrayBuffer or Node Buffer / typed array views
    if (buffer instanceof ArrayBuffer) this.cur = new Cursor(buffer, 0, buffer.byteLength)
    else this.cur = new Cursor(buffer.buffer, buffer.byteOffset, buffer.byteLength)
    this.h = null
  }

  /** Reference signature: PmxLoader.load(url) -> Promise<Model> (pmx-loader.ts:30-33). */
  static async load(path: string): Promise<Model> { return PmxLoader.loadFromBuffer(fs.readFileSync(path)) }
  static loadFromBuffer(buffer: ArrayBuffer | Uint8Array): Model { return new PmxLoader(buffer).parse() }

  text(): string {
    const c = this.cur
    const len = c.i32()
    if (len <= 0) return ''
    if (len > 1000) throw new RangeError('Suspicious string length: ' + len + ' at offset ' + (c.pos - 4))
    c.need(len)
    const s = this.decoder.decode(c.bytes.subarray(c.pos, c.pos + len))
    c.pos += len
    return s
  }

  parse(): Model {
    this.header()
    const geo = this.vertices()
    const indices = this.indices()
    const textures = this.guard('textures', () => this.textures(), [])
    const materials = this.guard('materials', () => this.materials(), [])
    const bones = this.guard('bones', () => this.bones(), [])
    const morphs = this.guard('morphs', () => this.morphs(geo.count, bones.length), null)
    let rigidbodies = [], joints = []
    if (morphs !== null && this.guard('display frames', () => this.displayFrames(), false)) {
      rigidbodies = this.guard('rigidbodies', () => this.rigidbodies(), [])
      joints = this.guard('joints', () => this.joints(), [])
    }
    return this.toModel(geo, indices, textures, materials, bones, morphs, rigidbodies, joints)
  }

  guard<T>(what: string, fn: () => T, fallback: T): T {
    try { return fn() } catch (e) { console.warn('Error parsing ' + what + ':', e.message); return fallback }
  }

  header(): void {
    const c = this.cur
    const sig = String.fromCharCode(c.u8(), c.u8(), c.u8())
    if (sig !== 'PMX') throw new Error('Not a PMX file')
    c.u8()
    const version = c.f32()
    if (version < 2.0 || version > 2.2) console.warn('PMX version ' + version + ' may not be fully supported')
    const globals = c.u8()
    if (globals < 8) throw new Error('Invalid globalsCount: ' + globals + ', expected at least 8')
    const g = []
    for (let i = 0; i < globals; i++) g.push(c.u8())
    this.h = {
      version, encoding: g[0], extraVec4: g[1], vertexIndexSize: g[2], textureIndexSize: g[3],
      materialIndexSize: g[4], boneIndexSize: g[5], morphIndexSize: g[6], rigidBodyIndexSize: g[7],
    }
    this.decoder = new TextDecoder(g[0] === 0 ? 'utf-16le' : 'utf-8')
    this.modelName = this.text()
    this.text(); this.text(); this.text()
  }

  vertices(): Geometry {
    const c = this.cur, h = this.h
    const count = c.i32()
    const pos = new Float32Array(count * 3), nrm = new Float32Array(count * 3), uv = new Float32Array(count * 2)
    const joints = new Uint16Array(count * 4), weights = new Uint8Array(count * 4)
    const bone = () => { const j = c.index(h.boneIndexSize); return j >= 0 ? j : 0 }
    for (let v = 0; v < count; v++) {
      pos[v * 3] = c.f32(); pos[v * 3 + 1] = c.f32(); pos[v * 3 + 2] = c.f32()
      nrm[v * 3] = c.f32(); nrm[v * 3 + 1] = c.f32(); nrm[v * 3 + 2] = c.f32()
      uv[v * 2] = c.f32(); uv[v * 2 + 1] = c.f32()
      c.skip(h.extraVec4 * 16)
      const kind = c.u8()
      const o = v * 4
      weights[o] = 255 // default: everything on joint slot 0
      if (kind === 0) { // BDEF1
        joints[o] = bone()
      } else if (kind === 1 || kind === 3) { // BDEF2, SDEF treated as BDEF2
        joints[o] = bone(); joints[o + 1] = bone()
        const w0 = clamp(Math.round(c.f32() * 255), 0, 255)
        weights[o] = w0; weights[o + 1] = clamp(255 - w0, 0, 255)
        if (kind === 3) c.skip(36) // C, R0, R1
      } else if (kind === 2 || kind === 4) { // BDEF4, QDEF treated as BDEF4
        for (let k = 0; k < 4; k++) joints[o + k] = bone()
        const q = [0, 0, 0, 0]
        let sum = 0
        for (let k = 0; k < 4; k++) { q[k] = Math.round(clamp(c.f32(), 0, 1) * 255); sum += q[k] }
        if (sum !== 0) { // rescale the rounded bytes so they add to exactly 255; slot 3 takes the remainder
          const scale = 255 / sum
          let acc = 0
          for (let k = 0; k < 3; k++) { const w = clamp(Math.round(q[k] * scale), 0, 255); weights[o + k] = w; acc += w }
          weights[o + 3] = clamp(255 - acc, 0, 255)
        }
      } else {
        throw new Error('Invalid bone weight type: ' + kind)
      }
      c.skip(4) // edge scale
    }
    return { count, pos, nrm, uv, joints, weights }
  }

  indices(): Uint32Array {
    const c = this.cur
    const n = c.i32()
    const out = new Uint32Array(n)
    for (let i = 0; i < n; i++) out[i] = c.vertexIndex(this.h.vertexIndexSize)
    return out
  }

  textures(): Texture[] {
    const n = this.cur.i32()
    const out = []
    for (let i = 0; i < n; i++) { const p = this.text(); out.push({ path: p, name: p.split('/').pop() || p }) }
    return out
  }

  materials(): Material[] {
    const c = this.cur, h = this.h
    const n = c.i32()
    const out = []
    for (let i = 0; i < n; i++) {
      const name = this.text()
      this.text()
      const diffuse = [c.f32(), c.f32(), c.f32(), c.f32()]
      const specular = [c.f32(), c.f32(), c.f32()]
      const shininess = c.f32()
      const ambient = [c.f32(), c.f32(), c.f32()]
      const edgeFlag = c.u8()
      const edgeColor = [c.f32(), c.f32(), c.f32(), c.f32()]
      const edgeSize = c.f32()
      const diffuseTextureIndex = c.index(h.textureIndexSize)
      const sphereTextureIndex = c.index(h.textureIndexSize)
      const sphereMode = c.u8()
      const sharedToon = c.u8() === 1
      const toonTextureIndex = sharedToon ? c.u8() : c.index(h.textureIndexSize)
      this.text()
      const vertexCount = c.i32()
      const lower = name.toLowerCase()
      const has = (list) => list.some((s) => lower.includes(s))
      out.push({
        name, diffuse, specular, ambient, shininess, diffuseTextureIndex, normalTextureIndex: -1, sphereTextureIndex,
        sphereMode, toonTextureIndex, edgeFlag, edgeColor, edgeSize, vertexCount,
        // render-side classification kept for API compatibility (pmx-loader.ts:282-301); unused by the deform path
        isEye: has(['目', '瞳', 'eye', 'pupil', 'iris', '眼', '睛', '眉']),
        isFace: has(['face', '脸']),
        isHair: has(['hair_f']),
      })
    }
    return out
  }

  bones(): Bone[] {
    const c = this.cur, bs = this.h.boneIndexSize
    const n = c.i32()
    const raw = new Array(n)
    for (let i = 0; i < n; i++) {
      const name = this.text()
      this.text()
      const p = c.vec3()
      const parent = c.index(bs)
      c.i32() // transform order
      const flags = c.u16()
      if (flags & 0x0001) c.index(bs); else c.skip(12) // tail: bone index or offset
      let appendParent, appendRatio
      const appendRotate = (flags & 0x0100) !== 0, appendMove = (flags & 0x0200) !== 0
      if (appendRotate || appendMove) { appendParent = c.index(bs); appendRatio = c.f32() }
      if (flags & 0x0400) c.skip(12) // axis limit
      if (flags & 0x0800) c.skip(24) // local axes
      if (flags & 0x2000) c.i32() // external parent
      if (flags & 0x0020) { // IK block (parsed to keep the cursor aligned; IK solving is out of scope)
        c.index(bs); c.i32(); c.f32()
        const links = c.i32()
        for (let l = 0; l < links; l++) { c.index(bs); if (c.u8() === 1) c.skip(24) }
      }
      raw[i] = { name, parent, p, appendParent, appendRatio, appendRotate, appendMove }
    }
    // absolute positions -> parent-relative bind translations (pmx-loader.ts:416-442)
    return raw.map((b) => {
      const hasParent = b.parent >= 0 && b.parent < n
      const pp = hasParent ? raw[b.parent].p : [0, 0, 0]
      return {
        name: b.name, parentIndex: b.parent, bindTranslation: [b.p[0] - pp[0], b.p[1] - pp[1], b.p[2] - pp[2]],
        children: [], appendParentIndex: b.appendParent, appendRatio: b.appendRatio, appendRotate: b.appendRotate,
        appendMove: b.appendMove,
      }
    })
  }

  // Morph section. Layout as documented by the reference's skipMorphs() (pmx-loader.ts:462-541).
  morphs(vertexCount: number, boneCount: number): MorphSet {
    const c = this.cur, h = this.h
    const n = c.i32()
    if (n < 0 || n > 100000) throw new RangeError('Suspicious morph count: ' + n)
    const names = [], types = new Uint8Array(n), panels = new Uint8Array(n), groups = new Array(n).fill(null)
    const offsets = new Uint32Array(n + 1)
    const vidx = [], dxyz = []
    const bmMorph = [], bmBone = [], bmT = [], bmQ = [] // bone-morph entries (type 2), ascending morph index
    const uvMorph = [], uvVertex = [], uvDelta = [] // UV-morph entries (type 3: the vertex buffer's own uv channel)
    for (let m = 0; m < n; m++) {
      names.push(this.text())
      this.text()
      panels[m] = c.u8()
      const type = c.u8()
      types[m] = type
      const cnt = c.i32()
      offsets[m] = vidx.length
      if (type === 0) {
        const list = []
        for (let k = 0; k < cnt; k++) list.push([c.index(h.morphIndexSize), c.f32()])
        groups[m] = list
      } else if (type === 1) {
        for (let k = 0; k < cnt; k++) {
          const v = c.vertexIndex(h.vertexIndexSize)
          const x = c.f32(), y = c.f32(), z = c.f32()
          if (v >= 0 && v < vertexCount) { vidx.push(v); dxyz.push(x, y, z) }
        }
      } else if (type === 2) {
        // bone index, translation vec3, rotation quaternion x y z w (28 B; the reference's skipper reads 24, pmx-loader.ts:489-497)
        for (let k = 0; k < cnt; k++) {
          const b = c.index(h.boneIndexSize)
          const tx = c.f32(), ty = c.f32(), tz = c.f32(), qx = c.f32(), qy = c.f32(), qz = c.f32(), qw = c.f32()
          if (b >= 0 && b < boneCount) { bmMorph.push(m); bmBone.push(b); bmT.push(tx, ty, tz); bmQ.push(qx, qy, qz, qw) }
        }
      } else if (type >= 3 && type <= 7) {
        // PMX 2.0: UV offsets are vec4 (16 B) and bone-morph offsets vec3 + quaternion (28 B). The reference's
        // skipper reads 8 B (:498-507) and 24 B (:489-497) and so loses sync on models that carry such morphs;
        // this loader follows the file format. Type 3 moves the uv the vertex buffer carries (first two components);
        // types 4-7 move the additional UV channels, which the vertex buffer does not hold (pmx-loader.ts:101-106).
        for (let k = 0; k < cnt; k++) {
          const v = c.vertexIndex(h.vertexIndexSize)
          const du = c.f32(), dv = c.f32()
          c.skip(8)
          if (type === 3 && v >= 0 && v < vertexCount) { uvMorph.push(m); uvVertex.push(v); uvDelta.push(du, dv) }
        }
      } else if (type === 8) {
        for (let k = 0; k < cnt; k++) { c.index(h.materialIndexSize); c.skip(1 + 28 * 4) }
      } else if (type === 9) { // PMX 2.1 flip: morphIndex + ratio
        for (let k = 0; k < cnt; k++) { c.index(h.morphIndexSize); c.f32() }
      } else if (type === 10) { // PMX 2.1 impulse
        for (let k = 0; k < cnt; k++) { c.index(h.rigidBodyIndexSize); c.skip(1 + 24) }
      } else {
        throw new Error('Unknown morph type ' + type)
      }
    }
    offsets[n] = vidx.length
    const boneEntries = {
      morph: Uint32Array.from(bmMorph), bone: Uint32Array.from(bmBone), translation: Float32Array.from(bmT), rotation: Float32Array.from(bmQ),
    }
    const uvEntries = { morph: Uint32Array.from(uvMorph), vertex: Uint32Array.from(uvVertex), delta: Float32Array.from(uvDelta) }
    return { names, types, panels, groups, offsets, vertexIndex: Uint32Array.from(vidx), deltas: Float32Array.from(dxyz), boneEntries, uvEntries }
  }

  displayFrames(): boolean {
    const c = this.cur, h = this.h
    const n = c.i32()
    if (n < 0 || n > 100000) throw new RangeError('Suspicious display frame count: ' + n)
    for (let i = 0; i < n; i++) {
      this.text(); this.text(); c.u8()
      const cnt = c.i32()
      for (let k = 0; k < cnt; k++) { if (c.u8() === 0) c.index(h.boneIndexSize); else c.index(h.morphIndexSize) }
    }
    return true
  }

  rigidbodies(): unknown[] {
    const c = this.cur
    const n = c.i32()
    if (n < 0 || n > 10000) throw new RangeError('Suspicious rigidbody count: ' + n)
    const out = []
    for (let i = 0; i < n; i++) {
      const name = this.text(), englishName = this.text()
      const boneIndex = c.index(this.h.boneIndexSize)
      const group = c.u8(), collisionMask = c.u16(), shape = c.u8()
      const size = c.vec3(), p = c.vec3(), r = c.vec3()
      const mass = c.f32(), linearDamping = c.f32(), angularDamping = c.f32(), restitution = c.f32(), friction = c.f32()
      const type = c.u8()
      out.push({
        name, englishName, boneIndex, group, collisionMask, shape, size: new Vec3(size[0], size[1], size[2]),
        shapePosition: new Vec3(p[0], p[1], p[2]), shapeRotation: new Vec3(r[0], r[1], r[2]), mass, linearDamping,
        angularDamping, restitution, friction, type, bodyOffsetMatrixInverse: Mat4.identity(),
      })
    }
    return out
  }

  joints(): unknown[] {
    const c = this.cur, rs = this.h.rigidBodyIndexSize
    const n = c.i32()
    if (n < 0 || n > 10000) throw new RangeError('Suspicious joint count: ' + n)
    const out = []
    const v3 = () => { const a = c.vec3(); return new Vec3(a[0], a[1], a[2]) }
    for (let i = 0; i < n; i++) {
      const name = this.text(), englishName = this.text()
      const type = c.u8()
      const rigidbodyIndexA = c.index(rs), rigidbodyIndexB = c.index(rs)
      out.push({
        name, englishName, type, rigidbodyIndexA, rigidbodyIndexB, position: v3(), rotation: v3(), positionMin: v3(),
        positionMax: v3(), rotationMin: v3(), rotationMax: v3(), springPosition: v3(), springRotation: v3(),
      })
    }
    return out
  }

  // translation-only inverse bind: IB = T(-sum of parent-relative offsets)   (pmx-loader.ts:791-824)
  static inverseBind(bones: Bone[]): Float32Array {
    const n = bones.length
    const world = new Float32Array(n * 3)
    const done = new Uint8Array(n)
    const f = Math.fround
    const solve = (i) => {
      if (done[i]) return
      const b = bones[i]
      let x = f(b.bindTranslation[0]), y = f(b.bindTranslation[1]), z = f(b.bindTranslation[2])
      if (b.parentIndex >= 0 && b.parentIndex < n) {
        solve(b.parentIndex)
        const p = b.parentIndex * 3
        // Mat4.multiply of two pure translations: f32(parent + local)
        x = f(world[p] + x); y = f(world[p + 1] + y); z = f(world[p + 2] + z)
      }
      world[i * 3] = x; world[i * 3 + 1] = y; world[i * 3 + 2] = z
      done[i] = 1
    }
    const inv = new Float32Array(n * 16)
    for (let i = 0; i < n; i++) {
      solve(i)
      const o = i * 16
      inv[o] = 1; inv[o + 5] = 1; inv[o + 10] = 1; inv[o + 15] = 1
      // identity.translateInPlace(-w): 0 + (-w), so a zero stays +0 as in the reference (pmx-loader.ts:820)
      inv[o + 12] = 0 - world[i * 3]; inv[o + 13] = 0 - world[i * 3 + 1]; inv[o + 14] = 0 - world[i * 3 + 2]
    }
    return inv
  }

  // Joints outside the skeleton lose their weight; the rest is rescaled to exactly 255 (pmx-loader.ts:855-951).
  static sanitizeSkinning(joints: Uint16Array, weights: Uint8Array, boneCount: number): void {
    const ok = (j) => j >= 0 && j < boneCount
    for (let o = 0; o < joints.length; o += 4) {
      let sum = 0, valid = 0
      for (let k = 0; k < 4; k++) {
        if (!ok(joints[o + k])) { weights[o + k] = 0; joints[o + k] = boneCount > 0 ? boneCount - 1 : 0 } else { sum += weights[o + k]; valid++ }
      }
      if (sum === 0 || valid === 0) {
        weights[o] = 255; weights[o + 1] = 0; weights[o + 2] = 0; weights[o + 3] = 0
        joints[o] = 0; joints[o + 1] = 0; joints[o + 2] = 0; joints[o + 3] = 0
        continue
      }
      if (sum === 255) continue
      const scale = 255 / sum
      let acc = 0
      for (let k = 0; k < 3; k++) { const w = clamp(Math.round(weights[o + k] * scale), 0, 255); weights[o + k] = w; acc += w }
      weights[o + 3] = clamp(255 - acc, 0, 255)
      const total = weights[o] + weights[o + 1] + weights[o + 2] + weights[o + 3]
      if (total !== 255) { // put the rounding remainder on the heaviest influence
        let big = 0
        for (let k = 1; k < 4; k++) if (weights[o + k] > weights[o + big]) big = k
        weights[o + big] = clamp(weights[o + big] + (255 - total), 0, 255)
      }
    }
  }

  toModel(geo: Geometry, indices: Uint32Array, textures: Texture[], materials: Material[], bones: Bone[], morphs: MorphSet | null, rigidbodies: unknown[], joints: unknown[]): Model {
    const n = geo.count
    const vertexData = new Float32Array(n * 8)
    for (let v = 0; v < n; v++) {
      const o = v * 8
      vertexData[o] = geo.pos[v * 3]; vertexData[o + 1] = geo.pos[v * 3 + 1]; vertexData[o + 2] = geo.pos[v * 3 + 2]
      vertexData[o + 3] = geo.nrm[v * 3]; vertexData[o + 4] = geo.nrm[v * 3 + 1]; vertexData[o + 5] = geo.nrm[v * 3 + 2]
      vertexData[o + 6] = geo.uv[v * 2]; vertexData[o + 7] = geo.uv[v * 2 + 1]
    }
    PmxLoader.sanitizeSkinning(geo.joints, geo.weights, bones.length)
    const skeleton = { bones, inverseBindMatrices: PmxLoader.inverseBind(bones) }
    const skinning = { joints: geo.joints, weights: geo.weights }
    return new Model(vertexData, indices, textures, materials, skeleton, skinning, rigidbodies, joints, morphs)
  }
}

export { PmxLoader }
