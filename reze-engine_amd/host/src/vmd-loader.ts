'use strict'
/*
 * Source: host/src/vmd-loader.ts (TypeScript). host/vmd-loader.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * vmd-loader.js — Vocaloid Motion Data parser.
 *
 * Mirrors the reference's VMDLoader (engine/src/vmd-loader.ts:14-179): `VMDLoader.load(path)` /
 * `VMDLoader.loadFromBuffer(buf)` return VMDKeyFrame[] = [{ time /* s, frame/30 *\/, boneFrames:
 * [{ boneName, frame, rotation: Quat }] }] grouped by time exactly as :56-97 does (sort by time,
 * new group when |dt| > 0.001).
 *
 * Additions: the reference skips the bone position (12 B) and the 64 interpolation bytes
 * (:129-140) and never reads the morph block; here they are kept (`position`, `interpolation`
 * on each bone frame) and the morph-frame block that follows (u32 count, then 15-byte Shift-JIS
 * name + u32 frame + f32 weight) is returned as `result.morphFrames` — the source of the morph
 * weights the fused kernel consumes (SURVEY §8f rank 2).
 */
import * as fs from 'fs'
import { TextDecoder } from 'util'
import { Quat, Vec3 } from './math'

import type { VMDKeyFrames } from './types'
const FRAME_RATE = 30.0

function makeDecoder(): TextDecoder {
  try { return new TextDecoder('shift-jis') } catch (e) { return new TextDecoder('utf-8') }
}

class VMDLoader {
  view: DataView
  bytes: Uint8Array
  pos: number
  decoder: TextDecoder
  constructor(buffer: ArrayBuffer | Uint8Array) {
    if (buffer instanceof ArrayBuffer) { this.view = new DataView(buffer); this.bytes = new Uint8Array(buffer) }
    else {
      this.view = new DataView(buffer.buffer, buffer.byteOffset, buffer.byteLength)
      this.bytes = new Uint8Array(buffer.buffer, buffer.byteOffset, buffer.byteLength)
    }
    this.pos = 0
    this.decoder = makeDecoder()
  }

  static async load(path: string): Promise<VMDKeyFrames> { return VMDLoader.loadFromBuffer(fs.readFileSync(path)) }
  static loadFromBuffer(buffer: ArrayBuffer | Uint8Array): VMDKeyFrames { return new VMDLoader(buffer).parse() }

  need(n: number): void {
    if (this.pos + n > this.view.byteLength) throw new RangeError('Offset ' + this.pos + ' + ' + n + ' exceeds buffer bounds ' + this.view.byteLength)
  }
  u32(): number { this.need(4); const v = this.view.getUint32(this.pos, true); this.pos += 4; return v }
  f32(): number { this.need(4); const v = this.view.getFloat32(this.pos, true); this.pos += 4; return v }

  // fixed 15-byte, NUL-terminated Shift-JIS name
  name15(): string {
    this.need(15)
    let n = 0
    while (n < 15 && this.bytes[this.pos + n] !== 0) n++
    const raw = this.bytes.subarray(this.pos, this.pos + n)
    this.pos += 15
    try { return this.decoder.decode(raw) } catch (e) { return String.fromCharCode.apply(null, raw) }
  }

  parse(): VMDKeyFrames {
    this.need(30)
    const magic = String.fromCharCode.apply(null, this.bytes.subarray(0, 30))
    if (magic.indexOf('Vocaloid Motion Data') !== 0) throw new Error('Invalid VMD file header')
    this.pos = 30 + 20 // header + model name

    const nBone = this.u32()
    const all = []
    for (let i = 0; i < nBone; i++) {
      const boneName = this.name15()
      const frame = this.u32()
      const position = new Vec3(this.f32(), this.f32(), this.f32())
      const rotation = new Quat(this.f32(), this.f32(), this.f32(), this.f32())
      this.need(64)
      const interpolation = this.bytes.slice(this.pos, this.pos + 64)
      this.pos += 64
      all.push({ time: frame / FRAME_RATE, boneFrame: { boneName, frame, rotation, position, interpolation } })
    }
    // group by time (stable sort keeps file order within a time, like Array.prototype.sort on Node >= 11)
    all.sort((a, b) => a.time - b.time)
    const keyFrames = []
    let t = -1.0, group = []
    for (const e of all) {
      if (Math.abs(e.time - t) > 0.001) {
        if (group.length > 0) keyFrames.push({ time: t, boneFrames: group })
        t = e.time
        group = [e.boneFrame]
      } else {
        group.push(e.boneFrame)
      }
    }
    if (group.length > 0) keyFrames.push({ time: t, boneFrames: group })

    // morph block (absent in truncated files: tolerate)
    const morphFrames = []
    if (this.pos + 4 <= this.view.byteLength) {
      const nMorph = this.u32()
      for (let i = 0; i < nMorph && this.pos + 23 <= this.view.byteLength; i++) {
        const morphName = this.name15()
        const frame = this.u32()
        const weight = this.f32()
        morphFrames.push({ morphName, frame, time: frame / FRAME_RATE, weight })
      }
      morphFrames.sort((a, b) => a.frame - b.frame)
    }
    keyFrames.morphFrames = morphFrames // array-with-extras keeps the reference's return type
    return keyFrames
  }
}

export { VMDLoader, FRAME_RATE }
