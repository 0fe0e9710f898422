'use strict'
/* Unit checks of host/math.js against hand-computed values (SURVEY §4 test plan item 1). Prints JSON. */
const path = require('path')
const { Vec3, Quat, Mat4, easeInOut } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host', 'math.js'))
const out = {}
out.ease = [0, 0.25, 0.5, 0.75, 1].map(easeInOut)
const qz = Quat.fromEuler(0, 0, Math.PI / 2)                    // 90 deg about Z
out.fromEulerZ = qz.toArray()
out.rotX = (() => { const v = qz.rotateVec(new Vec3(1, 0, 0)); return [v.x, v.y, v.z] })()
out.rotX2 = (() => { const v = qz.rotate(new Vec3(1, 0, 0)); return [v.x, v.y, v.z] })()
out.mulIdentity = qz.multiply(new Quat(0, 0, 0, 1)).toArray()
out.conj = qz.multiply(qz.conjugate()).toArray()
out.slerpHalf = Quat.slerp(new Quat(0, 0, 0, 1), qz, 0.5).toArray()
out.slerpNeg = Quat.slerp(new Quat(0, 0, 0, 1), new Quat(-qz.x, -qz.y, -qz.z, -qz.w), 0.5).toArray()   // shortest arc
out.slerpNear = Quat.slerp(new Quat(0, 0, 0, 1), new Quat(0, 0, 0.001, 0.9999995), 0.5).toArray()     // nlerp branch
out.euler = (() => { const e = Quat.fromEuler(0.3, 0, 0).toEuler(); const f = Quat.fromEuler(0, -0.2, 0).toEuler(); return [e.x, e.y, e.z, f.x, f.y, f.z] })()
out.fromTo = Quat.fromTo(new Vec3(1, 0, 0), new Vec3(0, 1, 0)).toArray()
const m = Mat4.fromPositionRotation(new Vec3(1, 2, 3), qz)
out.matFromQuat = Array.from(m.values)
out.matInvProduct = Array.from(m.multiply(m.inverse()).values)
out.matToQuat = m.toQuat().toArray()
out.translate = Array.from(Mat4.identity().translateInPlace(4, 5, 6).values)
const a = new Float32Array(32), r = new Float32Array(16)
a.set(m.values, 0); a.set(Mat4.identity().translateInPlace(1, 0, 0).values, 16)
Mat4.multiplyArrays(a, 0, a, 16, r, 0)
out.mulArrays = Array.from(r)
out.singular = Array.from(new Mat4(new Float32Array(16)).inverse().values)                             // -> identity + warning
out.vec = (() => { const v = new Vec3(3, 4, 0); return [v.length(), v.normalize().x, v.cross(new Vec3(0, 0, 1)).x, v.dot(new Vec3(1, 1, 1)), new Vec3(0, 0, 0).normalize().x] })()
const quiet = console.warn; console.warn = () => {}
console.log(JSON.stringify(out)); console.warn = quiet
