'use strict'
/*
 * skin_f32.js — JavaScript twin of oracle/rz_oracle.c. CPU ORACLE / CPU BASELINE: test
 * infrastructure only (tests/, bench.py's cpu_baseline leg). The product never requires it.
 *
 * Same reference lines, same evaluation order, every operation rounded to binary32 with
 * Math.fround (double rounding of + - * / sqrt from 53 to 24 bits is innocuous, so this is exact
 * f32 arithmetic and must agree with the C oracle bit for bit):
 *   palette  /root/reference/engine/src/engine.ts:926-928
 *   skin     /root/reference/engine/src/engine.ts:253-272   (unorm8 fetch :354-355)
 *   morph    no reference implementation (pmx-loader.ts:450-553 skips morphs) — build-defined:
 *            p~ = p + sum_m w_m * delta_m[v], ascending m, zero weights skipped.
 */
const f = Math.fround

function palette(world, invBind, nBones, out) {
  for (let b = 0; b < nBones; b++) {
    const o = b * 16
    for (let c = 0; c < 4; c++) {
      const b0 = invBind[o + c * 4], b1 = invBind[o + c * 4 + 1], b2 = invBind[o + c * 4 + 2], b3 = invBind[o + c * 4 + 3]
      for (let r = 0; r < 4; r++) {
        let t = f(world[o + r] * b0)
        t = f(t + f(world[o + 4 + r] * b1))
        t = f(t + f(world[o + 8 + r] * b2))
        t = f(t + f(world[o + 12 + r] * b3))
        out[o + c * 4 + r] = t
      }
    }
  }
  return out
}

// dense morph + skin for vertices [v0, v1). deltas: [M][V][3] or null. pos/nrm packed [V][3].
function deformRange(v0, v1, nVerts, nMorphs, pos, nrm, joints, weights, skin, deltas, mw, outPos, outNrm) {
  const inv255 = 255
  for (let v = v0; v < v1; v++) {
    let px = pos[v * 3], py = pos[v * 3 + 1], pz = pos[v * 3 + 2]
    if (deltas !== null && nMorphs > 0) {
      let dx = 0, dy = 0, dz = 0
      for (let m = 0; m < nMorphs; m++) {
        const w = mw[m]
        if (w === 0) continue
        const d = (m * nVerts + v) * 3
        dx = f(dx + f(w * deltas[d]))
        dy = f(dy + f(w * deltas[d + 1]))
        dz = f(dz + f(w * deltas[d + 2]))
      }
      px = f(px + dx); py = f(py + dy); pz = f(pz + dz)
    }
    const nx = nrm[v * 3], ny = nrm[v * 3 + 1], nz = nrm[v * 3 + 2]
    let w0 = f(weights[v * 4] / inv255), w1 = f(weights[v * 4 + 1] / inv255)
    let w2 = f(weights[v * 4 + 2] / inv255), w3 = f(weights[v * 4 + 3] / inv255)
    const sum = f(f(f(w0 + w1) + w2) + w3)
    if (sum > f(0.0001)) {
      const inv = f(1 / sum)
      w0 = f(w0 * inv); w1 = f(w1 * inv); w2 = f(w2 * inv); w3 = f(w3 * inv)
    } else {
      w0 = 1; w1 = 0; w2 = 0; w3 = 0
    }
    let sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0
    for (let i = 0; i < 4; i++) {
      const o = joints[v * 4 + i] * 16
      const w = i === 0 ? w0 : i === 1 ? w1 : i === 2 ? w2 : w3
      const ax = f(f(f(f(skin[o] * px) + f(skin[o + 4] * py)) + f(skin[o + 8] * pz)) + skin[o + 12])
      const ay = f(f(f(f(skin[o + 1] * px) + f(skin[o + 5] * py)) + f(skin[o + 9] * pz)) + skin[o + 13])
      const az = f(f(f(f(skin[o + 2] * px) + f(skin[o + 6] * py)) + f(skin[o + 10] * pz)) + skin[o + 14])
      sx = f(sx + f(ax * w)); sy = f(sy + f(ay * w)); sz = f(sz + f(az * w))
      const bx = f(f(f(skin[o] * nx) + f(skin[o + 4] * ny)) + f(skin[o + 8] * nz))
      const by = f(f(f(skin[o + 1] * nx) + f(skin[o + 5] * ny)) + f(skin[o + 9] * nz))
      const bz = f(f(f(skin[o + 2] * nx) + f(skin[o + 6] * ny)) + f(skin[o + 10] * nz))
      tx = f(tx + f(bx * w)); ty = f(ty + f(by * w)); tz = f(tz + f(bz * w))
    }
    outPos[v * 3] = sx; outPos[v * 3 + 1] = sy; outPos[v * 3 + 2] = sz
    const len = f(Math.sqrt(f(f(f(tx * tx) + f(ty * ty)) + f(tz * tz))))
    if (len > 0 && isFinite(len)) {
      outNrm[v * 3] = f(tx / len); outNrm[v * 3 + 1] = f(ty / len); outNrm[v * 3 + 2] = f(tz / len)
    } else {
      outNrm[v * 3] = nx; outNrm[v * 3 + 1] = ny; outNrm[v * 3 + 2] = nz
    }
  }
}

module.exports = { palette, deformRange }
