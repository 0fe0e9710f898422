'use strict'
/*
 * Source: host/src/math.ts (TypeScript). host/math.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
 * math.js — host-side math primitives with the reference's public surface
 * (reference engine/src/math.ts: easeInOut :2-4, Vec3 :6-54, Quat :56-232, Mat4 :234-546).
 *
 * Exported names, method names, argument meaning and numeric conventions are the reference's:
 * scalars are JS doubles, Mat4 stores a column-major Float32Array(16), so every matrix store
 * rounds to binary32 exactly as the reference does (this is what makes the CPU bone solve of
 * host/model.js reproduce the reference's world matrices bit for bit). The implementation is
 * written for this engine: flat helper functions over typed arrays that Model's forward
 * kinematics call without allocating, wrapped by thin classes for API compatibility.
 */

import type { NumArray, Quad } from './types'
function easeInOut(t: number): number {
  if (t < 0.5) return 2 * t * t
  const u = -2 * t + 2
  return 1 - (u * u) / 2
}

class Vec3 {
  x: number
  y: number
  z: number
  constructor(x: number, y: number, z: number) { this.x = x; this.y = y; this.z = z }
  add(o: Vec3): Vec3 { return new Vec3(this.x + o.x, this.y + o.y, this.z + o.z) }
  subtract(o: Vec3): Vec3 { return new Vec3(this.x - o.x, this.y - o.y, this.z - o.z) }
  scale(k: number): Vec3 { return new Vec3(this.x * k, this.y * k, this.z * k) }
  dot(o: Vec3): number { return this.x * o.x + this.y * o.y + this.z * o.z }
  cross(o: Vec3): Vec3 {
    const ax = this.x, ay = this.y, az = this.z
    return new Vec3(ay * o.z - az * o.y, az * o.x - ax * o.z, ax * o.y - ay * o.x)
  }
  length(): number { return Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z) }
  normalize(): Vec3 {
    const n = this.length()
    return n === 0 ? new Vec3(0, 0, 0) : new Vec3(this.x / n, this.y / n, this.z / n)
  }
  clone(): Vec3 { return new Vec3(this.x, this.y, this.z) }
}

/* ---- quaternion kernels on plain numbers (x, y, z, w order everywhere) ---- */

// Spherical interpolation with the reference's branch structure (math.ts:156-189): shortest arc,
// normalised lerp above cos 0.9995, classic slerp otherwise. Writes into out[0..3].
function slerpInto(out: NumArray, ax: number, ay: number, az: number, aw: number, bx: number, by: number, bz: number, bw: number, t: number): NumArray {
  let c = ax * bx + ay * by + az * bz + aw * bw
  if (c < 0) { c = -c; bx = -bx; by = -by; bz = -bz; bw = -bw }
  if (c > 0.9995) {
    const x = ax + t * (bx - ax), y = ay + t * (by - ay), z = az + t * (bz - az), w = aw + t * (bw - aw)
    const k = 1 / Math.hypot(x, y, z, w)
    out[0] = x * k; out[1] = y * k; out[2] = z * k; out[3] = w * k
    return out
  }
  const th0 = Math.acos(c)
  const s = Math.sin(th0)
  const th = th0 * t
  const ka = Math.sin(th0 - th) / s
  const kb = Math.sin(th) / s
  out[0] = ka * ax + kb * bx; out[1] = ka * ay + kb * by; out[2] = ka * az + kb * bz; out[3] = ka * aw + kb * bw
  return out
}

class Quat {
  x: number
  y: number
  z: number
  w: number
  constructor(x: number, y: number, z: number, w: number) { this.x = x; this.y = y; this.z = z; this.w = w }
  add(o: Quat): Quat { return new Quat(this.x + o.x, this.y + o.y, this.z + o.z, this.w + o.w) }
  clone(): Quat { return new Quat(this.x, this.y, this.z, this.w) }
  conjugate(): Quat { return new Quat(-this.x, -this.y, -this.z, this.w) }
  length(): number { return Math.sqrt(this.x * this.x + this.y * this.y + this.z * this.z + this.w * this.w) }
  normalize(): Quat {
    const n = this.length()
    return n === 0 ? new Quat(0, 0, 0, 1) : new Quat(this.x / n, this.y / n, this.z / n, this.w / n)
  }
  toArray(): Quad { return [this.x, this.y, this.z, this.w] }
  // Hamilton product this * o
  multiply(o: Quat): Quat {
    const x = this.x, y = this.y, z = this.z, w = this.w
    return new Quat(
      w * o.x + x * o.w + y * o.z - z * o.y,
      w * o.y - x * o.z + y * o.w + z * o.x,
      w * o.z + x * o.y - y * o.x + z * o.w,
      w * o.w - x * o.x - y * o.y - z * o.z)
  }
  // q v q^-1 via t = 2 q.xyz x v
  rotateVec(v: Vec3): Vec3 {
    const x = this.x, y = this.y, z = this.z, w = this.w
    const tx = 2 * (y * v.z - z * v.y), ty = 2 * (z * v.x - x * v.z), tz = 2 * (x * v.y - y * v.x)
    return new Vec3(v.x + w * tx + (y * tz - z * ty), v.y + w * ty + (z * tx - x * tz), v.z + w * tz + (x * ty - y * tx))
  }
  rotate(v: Vec3): Vec3 {
    const q = new Vec3(this.x, this.y, this.z)
    const uv = q.cross(v)
    return v.add(uv.scale(2 * this.w)).add(q.cross(uv).scale(2))
  }
  // ZXY order, left-handed (PMX); inverse of fromEuler
  toEuler(): Vec3 {
    const x = this.x, y = this.y, z = this.z, w = this.w
    const rx = Math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    const sp = 2 * (w * y - z * x)
    const ry = Math.abs(sp) >= 1 ? (sp >= 0 ? Math.PI / 2 : -Math.PI / 2) : Math.asin(sp)
    const rz = Math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return new Vec3(rx, ry, rz)
  }
  static slerp(a: Quat, b: Quat, t: number): Quat {
    const o = slerpInto([0, 0, 0, 1], a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, t)
    return new Quat(o[0], o[1], o[2], o[3])
  }
  static fromEuler(rotX: number, rotY: number, rotZ: number): Quat {
    const cx = Math.cos(rotX * 0.5), sx = Math.sin(rotX * 0.5)
    const cy = Math.cos(rotY * 0.5), sy = Math.sin(rotY * 0.5)
    const cz = Math.cos(rotZ * 0.5), sz = Math.sin(rotZ * 0.5)
    return new Quat(
      cy * sx * cz + sy * cx * sz,
      sy * cx * cz - cy * sx * sz,
      cy * cx * sz - sy * sx * cz,
      cy * cx * cz + sy * sx * sz).normalize()
  }
  static fromTo(from: Vec3, to: Vec3): Quat {
    const d = from.dot(to)
    if (d > 0.999999) return new Quat(0, 0, 0, 1)
    if (d < -0.999999) {
      let axis = from.cross(new Vec3(1, 0, 0))
      if (axis.length() < 0.001) axis = from.cross(new Vec3(0, 1, 0))
      return new Quat(axis.x, axis.y, axis.z, 0).normalize()
    }
    const axis = from.cross(to)
    const w = Math.sqrt((1 + d) * 2)
    const k = 1 / w
    return new Quat(axis.x * k, axis.y * k, axis.z * k, w * 0.5).normalize()
  }
}

/* ---- 4x4 kernels on Float32Array segments (column-major; doubles in flight, f32 on store) ---- */

// out[oo..] = a[ao..] * b[bo..]; out must not alias a or b.
function mulInto(out: NumArray, oo: number, a: NumArray, ao: number, b: NumArray, bo: number): void {
  for (let c = 0; c < 16; c += 4) {
    const b0 = b[bo + c], b1 = b[bo + c + 1], b2 = b[bo + c + 2], b3 = b[bo + c + 3]
    out[oo + c] = a[ao] * b0 + a[ao + 4] * b1 + a[ao + 8] * b2 + a[ao + 12] * b3
    out[oo + c + 1] = a[ao + 1] * b0 + a[ao + 5] * b1 + a[ao + 9] * b2 + a[ao + 13] * b3
    out[oo + c + 2] = a[ao + 2] * b0 + a[ao + 6] * b1 + a[ao + 10] * b2 + a[ao + 14] * b3
    out[oo + c + 3] = a[ao + 3] * b0 + a[ao + 7] * b1 + a[ao + 11] * b2 + a[ao + 15] * b3
  }
}

// rotation matrix of a unit quaternion into out[oo..] (math.ts:352-384 term order)
function quatToMatInto(out: NumArray, oo: number, x: number, y: number, z: number, w: number): void {
  const x2 = x + x, y2 = y + y, z2 = z + z
  const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2
  const wx = w * x2, wy = w * y2, wz = w * z2
  out[oo] = 1 - (yy + zz); out[oo + 1] = xy + wz; out[oo + 2] = xz - wy; out[oo + 3] = 0
  out[oo + 4] = xy - wz; out[oo + 5] = 1 - (xx + zz); out[oo + 6] = yz + wx; out[oo + 7] = 0
  out[oo + 8] = xz + wy; out[oo + 9] = yz - wx; out[oo + 10] = 1 - (xx + yy); out[oo + 11] = 0
  out[oo + 12] = 0; out[oo + 13] = 0; out[oo + 14] = 0; out[oo + 15] = 1
}

function identityInto(out: NumArray, oo: number): void {
  out.fill(0, oo, oo + 16)
  out[oo] = 1; out[oo + 5] = 1; out[oo + 10] = 1; out[oo + 15] = 1
}

class Mat4 {
  values: Float32Array
  constructor(values: Float32Array) { this.values = values }
  static identity(): Mat4 { const v = new Float32Array(16); identityInto(v, 0); return new Mat4(v) }
  static fromQuat(x: number, y: number, z: number, w: number): Mat4 { const v = new Float32Array(16); quatToMatInto(v, 0, x, y, z, w); return new Mat4(v) }
  static fromPositionRotation(position: Vec3, rotation: Quat): Mat4 {
    const m = Mat4.fromQuat(rotation.x, rotation.y, rotation.z, rotation.w)
    m.values[12] = position.x; m.values[13] = position.y; m.values[14] = position.z
    return m
  }
  static multiplyArrays(a: Float32Array, aOffset: number, b: Float32Array, bOffset: number, out: Float32Array, outOffset: number): void { mulInto(out, outOffset, a, aOffset, b, bOffset) }
  // left-handed (Z+ forward) projection, depth 0..1
  static perspective(fov: number, aspect: number, near: number, far: number): Mat4 {
    const f = 1.0 / Math.tan(fov / 2), ri = 1.0 / (far - near)
    const v = new Float32Array(16)
    v[0] = f / aspect; v[5] = f; v[10] = (far + near) * ri; v[11] = 1; v[14] = -near * far * ri * 2
    return new Mat4(v)
  }
  static lookAt(eye: Vec3, target: Vec3, up: Vec3): Mat4 {
    const fwd = target.subtract(eye).normalize()
    const right = up.cross(fwd).normalize()
    const u = fwd.cross(right).normalize()
    return new Mat4(new Float32Array([
      right.x, u.x, fwd.x, 0, right.y, u.y, fwd.y, 0, right.z, u.z, fwd.z, 0,
      -right.dot(eye), -u.dot(eye), -fwd.dot(eye), 1]))
  }
  static toQuatFromArray(m: Float32Array, offset: number): Quat {
    const m00 = m[offset], m01 = m[offset + 4], m02 = m[offset + 8]
    const m10 = m[offset + 1], m11 = m[offset + 5], m12 = m[offset + 9]
    const m20 = m[offset + 2], m21 = m[offset + 6], m22 = m[offset + 10]
    const tr = m00 + m11 + m22
    let x, y, z, w
    if (tr > 0) {
      const s = Math.sqrt(tr + 1.0) * 2
      w = 0.25 * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s
    } else if (m00 > m11 && m00 > m22) {
      const s = Math.sqrt(1.0 + m00 - m11 - m22) * 2
      w = (m21 - m12) / s; x = 0.25 * s; y = (m01 + m10) / s; z = (m02 + m20) / s
    } else if (m11 > m22) {
      const s = Math.sqrt(1.0 + m11 - m00 - m22) * 2
      w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25 * s; z = (m12 + m21) / s
    } else {
      const s = Math.sqrt(1.0 + m22 - m00 - m11) * 2
      w = (m10 - m01) / s; x = (m02 + m20) / s; y = (m12 + m21) / s; z = 0.25 * s
    }
    const k = 1 / Math.hypot(x, y, z, w)
    return new Quat(x * k, y * k, z * k, w * k)
  }
  multiply(other: Mat4): Mat4 { const v = new Float32Array(16); mulInto(v, 0, this.values, 0, other.values, 0); return new Mat4(v) }
  clone(): Mat4 { return new Mat4(this.values.slice()) }
  getPosition(): Vec3 { return new Vec3(this.values[12], this.values[13], this.values[14]) }
  toQuat(): Quat { return Mat4.toQuatFromArray(this.values, 0) }
  setIdentity(): this { identityInto(this.values, 0); return this }
  translateInPlace(tx: number, ty: number, tz: number): this { this.values[12] += tx; this.values[13] += ty; this.values[14] += tz; return this }
  // general inverse by cofactors of 2x2 sub-determinants; singular (|det| < 1e-10) -> identity + warning
  inverse(): Mat4 {
    const m = this.values
    const a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3], a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7]
    const a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11], a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15]
    const s0 = a00 * a11 - a01 * a10, s1 = a00 * a12 - a02 * a10, s2 = a00 * a13 - a03 * a10
    const s3 = a01 * a12 - a02 * a11, s4 = a01 * a13 - a03 * a11, s5 = a02 * a13 - a03 * a12
    const c0 = a20 * a31 - a21 * a30, c1 = a20 * a32 - a22 * a30, c2 = a20 * a33 - a23 * a30
    const c3 = a21 * a32 - a22 * a31, c4 = a21 * a33 - a23 * a31, c5 = a22 * a33 - a23 * a32
    let det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0
    if (Math.abs(det) < 1e-10) {
      console.warn('Matrix is not invertible (determinant near zero)')
      return Mat4.identity()
    }
    det = 1.0 / det
    const o = new Float32Array(16)
    o[0] = (a11 * c5 - a12 * c4 + a13 * c3) * det
    o[1] = (a02 * c4 - a01 * c5 - a03 * c3) * det
    o[2] = (a31 * s5 - a32 * s4 + a33 * s3) * det
    o[3] = (a22 * s4 - a21 * s5 - a23 * s3) * det
    o[4] = (a12 * c2 - a10 * c5 - a13 * c1) * det
    o[5] = (a00 * c5 - a02 * c2 + a03 * c1) * det
    o[6] = (a32 * s2 - a30 * s5 - a33 * s1) * det
    o[7] = (a20 * s5 - a22 * s2 + a23 * s1) * det
    o[8] = (a10 * c4 - a11 * c2 + a13 * c0) * det
    o[9] = (a01 * c2 - a00 * c4 - a03 * c0) * det
    o[10] = (a30 * s4 - a31 * s2 + a33 * s0) * det
    o[11] = (a21 * s2 - a20 * s4 - a23 * s0) * det
    o[12] = (a11 * c1 - a10 * c3 - a12 * c0) * det
    o[13] = (a00 * c3 - a01 * c1 + a02 * c0) * det
    o[14] = (a31 * s1 - a30 * s3 - a32 * s0) * det
    o[15] = (a20 * s3 - a21 * s1 + a22 * s0) * det
    return new Mat4(o)
  }
}

const kernels = { slerpInto, mulInto, quatToMatInto, identityInto }

export { easeInOut, Vec3, Quat, Mat4, kernels }
