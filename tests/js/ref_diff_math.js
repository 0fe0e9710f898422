'use strict'
/* DEV-CONTAINER ONLY: every public method of the reference's math.ts (type-erased copy under argv[2]) against the same
 * method of host/math.js on random inputs; results must be identical bit for bit (as f64 for numbers, as stored for
 * Float32Array). node ref_diff_math.js <erased dir> <seed> */
const path = require('path')
const R = require(path.join(process.argv[2], 'math'))
const M = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host', 'math.js'))
let state = (parseInt(process.argv[3], 10) >>> 0) || 1
const rnd = () => { state |= 0; state = (state + 0x6D2B79F5) | 0; let t = Math.imul(state ^ (state >>> 15), 1 | state); t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296 }
const num = () => (rnd() < 0.1 ? [0, 1, -1, 1e-7, 0.9995, 3][Math.floor(rnd() * 6)] : (rnd() - 0.5) * 4)
const flat = (v) => {
  if (v === null || v === undefined) return [String(v)]
  if (typeof v === 'number') return [v]
  if (v.values) return Array.from(v.values)
  if (Array.isArray(v)) return v.map(Number)
  if (v.w !== undefined) return [v.x, v.y, v.z, v.w]
  if (v.z !== undefined) return [v.x, v.y, v.z]
  return [JSON.stringify(v)]
}
const same = (a, b) => a.length === b.length && a.every((x, i) => Object.is(x, b[i]) || (typeof x === 'number' && typeof b[i] === 'number' && x === b[i]))
const mk = (L) => ({
  vec: (a) => new L.Vec3(a[0], a[1], a[2]), quat: (a, n) => { const q = new L.Quat(a[0], a[1], a[2], a[3]); return n ? q.normalize() : q },
  mat: (a) => new L.Mat4(Float32Array.from(a)),
})
const fr = mk(R), fm = mk(M)
let cases = 0
const refOk = {}
const check = (name, f) => {
  const args = { v1: [num(), num(), num()], v2: [num(), num(), num()], q1: [num(), num(), num(), num()], q2: [num(), num(), num(), num()], t: rnd() * 1.2 - 0.1,
    m1: Array.from({ length: 16 }, num), m2: Array.from({ length: 16 }, num), s: num(), unit: rnd() < 0.7 }
  let a, b
  try { a = flat(f(R, fr, args)); refOk[name] = (refOk[name] || 0) + 1 } catch (e) { a = ['throw']; refOk[name] = refOk[name] || 0 }
  try { b = flat(f(M, fm, args)) } catch (e) { b = ['throw'] }
  if (!same(a, b)) { console.error('DIVERGED ' + name + '\n ref  ' + a + '\n mine ' + b + '\n args ' + JSON.stringify(args)); process.exit(1) }
  cases++
}
const rigid = (L, f, g) => L.Mat4.fromPositionRotation(f.vec(g.v1), f.quat(g.q1, true))
for (let it = 0; it < 400; it++) {
  check('Vec3.add', (L, f, g) => f.vec(g.v1).add(f.vec(g.v2))); check('Vec3.subtract', (L, f, g) => f.vec(g.v1).subtract(f.vec(g.v2)))
  check('Vec3.length', (L, f, g) => f.vec(g.v1).length()); check('Vec3.normalize', (L, f, g) => f.vec(g.v1).normalize())
  check('Vec3.cross', (L, f, g) => f.vec(g.v1).cross(f.vec(g.v2))); check('Vec3.dot', (L, f, g) => f.vec(g.v1).dot(f.vec(g.v2)))
  check('Vec3.scale', (L, f, g) => f.vec(g.v1).scale(g.s))
  check('Quat.add', (L, f, g) => f.quat(g.q1).add(f.quat(g.q2))); check('Quat.multiply', (L, f, g) => f.quat(g.q1, g.unit).multiply(f.quat(g.q2, g.unit)))
  check('Quat.conjugate', (L, f, g) => f.quat(g.q1).conjugate()); check('Quat.length', (L, f, g) => f.quat(g.q1).length())
  check('Quat.normalize', (L, f, g) => f.quat(g.q1).normalize()); check('Quat.rotateVec', (L, f, g) => f.quat(g.q1, g.unit).rotateVec(f.vec(g.v1)))
  check('Quat.rotate', (L, f, g) => f.quat(g.q1, g.unit).rotate(f.vec(g.v1))); check('Quat.fromTo', (L, f, g) => L.Quat.fromTo(f.vec(g.v1).normalize(), f.vec(g.v2).normalize()))
  check('Quat.toArray', (L, f, g) => f.quat(g.q1).toArray()); check('Quat.slerp', (L, f, g) => L.Quat.slerp(f.quat(g.q1, g.unit), f.quat(g.q2, g.unit), g.t))
  check('Quat.fromEuler', (L, f, g) => L.Quat.fromEuler(g.v1[0], g.v1[1], g.v1[2])); check('Quat.toEuler', (L, f, g) => f.quat(g.q1, true).toEuler())
  check('Mat4.identity', (L) => L.Mat4.identity()); check('Mat4.perspective', (L, f, g) => L.Mat4.perspective(0.3 + Math.abs(g.s), 1.5, 0.1, 100))
  check('Mat4.lookAt', (L, f, g) => L.Mat4.lookAt(f.vec(g.v1), f.vec(g.v2), f.vec([0, 1, 0]))); check('Mat4.multiply', (L, f, g) => f.mat(g.m1).multiply(f.mat(g.m2)))
  check('Mat4.multiplyArrays', (L, f, g) => { const o = new Float32Array(16); L.Mat4.multiplyArrays(Float32Array.from(g.m1), 0, Float32Array.from(g.m2), 0, o, 0); return Array.from(o) })
  check('Mat4.fromQuat', (L, f, g) => L.Mat4.fromQuat(g.q1[0], g.q1[1], g.q1[2], g.q1[3])); check('Mat4.fromPositionRotation', (L, f, g) => rigid(L, f, g))
  check('Mat4.getPosition', (L, f, g) => f.mat(g.m1).getPosition()); check('Mat4.toQuat', (L, f, g) => rigid(L, f, g).toQuat())
  check('Mat4.toQuatFromArray', (L, f, g) => L.Mat4.toQuatFromArray(rigid(L, f, g).values, 0)); check('Mat4.setIdentity', (L, f, g) => f.mat(g.m1).setIdentity())
  check('Mat4.translateInPlace', (L, f, g) => f.mat(g.m1).translateInPlace(g.v1[0], g.v1[1], g.v1[2])); check('Mat4.inverse', (L, f, g) => f.mat(g.m1).inverse())
  check('Mat4.inverse(rigid)', (L, f, g) => rigid(L, f, g).inverse()); check('Mat4.clone', (L, f, g) => f.mat(g.m1).clone())
  check('easeInOut', (L, f, g) => L.easeInOut(g.t))
}
const never = Object.keys(refOk).filter((k) => refOk[k] === 0)
if (never.length) { console.error('the reference never answered: ' + never.join(', ')); process.exit(1) }
console.log(JSON.stringify({ cases, methods: Object.keys(refOk).length }))
