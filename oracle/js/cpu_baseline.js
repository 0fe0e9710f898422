'use strict'
/*
 * cpu_baseline.js — all-core JavaScript CPU skin of a bounded sample of the bench workload
 * (BASELINE.md §3): worker_threads, one worker per logical core, contiguous vertex ranges over
 * SharedArrayBuffers, f32 arithmetic identical to the oracle. Test/bench infrastructure only.
 *
 * usage: node cpu_baseline.js <dir> <nVerts> <nBones> <nMorphs> <threads> <seconds> [--dump]
 *   <dir> holds pos.f32 nrm.f32 joints.u16 weights.u8 world.f32 invbind.f32 [deltas.f32 mw.f32]
 * prints one JSON line: {verts_per_s, threads, frames, single_thread_verts_per_s}
 */
const fs = require('fs')
const path = require('path')
const { Worker, isMainThread, parentPort, workerData } = require('worker_threads')
const { performance } = require('perf_hooks')
const K = require('./skin_f32.js')

function shared(Type, buf) {
  const sab = new SharedArrayBuffer(buf.byteLength)
  new Uint8Array(sab).set(new Uint8Array(buf.buffer, buf.byteOffset, buf.byteLength))
  return new Type(sab)
}

if (!isMainThread) {
  const d = workerData
  parentPort.on('message', (msg) => {
    if (msg === 'stop') { process.exit(0) }
    K.deformRange(d.v0, d.v1, d.nVerts, d.nMorphs, d.pos, d.nrm, d.joints, d.weights, d.skin,
      d.nMorphs > 0 ? d.deltas : null, d.mw, d.outPos, d.outNrm)
    parentPort.postMessage('done')
  })
} else {
  const [dir, nVertsS, nBonesS, nMorphsS, threadsS, secondsS, flag] = process.argv.slice(2)
  const nVerts = +nVertsS, nBones = +nBonesS, nMorphs = +nMorphsS
  const threads = Math.max(1, +threadsS), seconds = +secondsS
  const rd = (n) => fs.readFileSync(path.join(dir, n))
  const pos = shared(Float32Array, rd('pos.f32')), nrm = shared(Float32Array, rd('nrm.f32'))
  const joints = shared(Uint16Array, rd('joints.u16')), weights = shared(Uint8Array, rd('weights.u8'))
  const world = new Float32Array(new Uint8Array(rd('world.f32')).buffer)
  const invBind = new Float32Array(new Uint8Array(rd('invbind.f32')).buffer)
  const deltas = nMorphs > 0 ? shared(Float32Array, rd('deltas.f32')) : new Float32Array(new SharedArrayBuffer(4))
  const mw = nMorphs > 0 ? shared(Float32Array, rd('mw.f32')) : new Float32Array(new SharedArrayBuffer(4))
  const skin = new Float32Array(new SharedArrayBuffer(nBones * 64))
  const outPos = new Float32Array(new SharedArrayBuffer(nVerts * 12))
  const outNrm = new Float32Array(new SharedArrayBuffer(nVerts * 12))

  // single-thread rate on a slice (same loop, main thread)
  K.palette(world, invBind, nBones, skin)
  const slice = Math.min(nVerts, 20000)
  K.deformRange(0, Math.min(slice, 2000), nVerts, nMorphs, pos, nrm, joints, weights, skin, nMorphs > 0 ? deltas : null, mw, outPos, outNrm)
  let t0 = performance.now()
  K.deformRange(0, slice, nVerts, nMorphs, pos, nrm, joints, weights, skin, nMorphs > 0 ? deltas : null, mw, outPos, outNrm)
  const single = slice / ((performance.now() - t0) / 1000)

  const chunk = Math.ceil(nVerts / threads)
  const workers = []
  for (let t = 0; t < threads; t++) {
    const v0 = Math.min(nVerts, t * chunk), v1 = Math.min(nVerts, v0 + chunk)
    workers.push(new Worker(__filename, { workerData: { v0, v1, nVerts, nMorphs, pos, nrm, joints, weights, skin, deltas, mw, outPos, outNrm } }))
  }
  const frame = () => new Promise((resolve) => {
    K.palette(world, invBind, nBones, skin)      // per-frame palette on the main thread
    let left = workers.length
    for (const w of workers) {
      w.once('message', () => { if (--left === 0) resolve() })
      w.postMessage('go')
    }
  })
  ;(async () => {
    for (let i = 0; i < 3; i++) await frame()
    let frames = 0
    t0 = performance.now()
    do { await frame(); frames++ } while ((performance.now() - t0) / 1000 < seconds && frames < 2000)
    const el = (performance.now() - t0) / 1000
    if (flag === '--dump') {
      fs.writeFileSync(path.join(dir, 'out_pos.f32'), Buffer.from(outPos.buffer))
      fs.writeFileSync(path.join(dir, 'out_nrm.f32'), Buffer.from(outNrm.buffer))
    }
    console.log(JSON.stringify({ verts_per_s: nVerts * frames / el, threads, frames, single_thread_verts_per_s: single }))
    for (const w of workers) w.postMessage('stop')
    setTimeout(() => process.exit(0), 50)
  })()
}
