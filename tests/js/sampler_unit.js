'use strict'
const path = require('path'), fs = require('fs')
const { VMDLoader, VMDSampler, Model, Quat } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const { bezier } = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host', 'vmd-sampler.js'))
const k = VMDLoader.loadFromBuffer(fs.readFileSync(process.argv[2]))
const s = new VMDSampler(k)
const out = { lastFrame: s.lastFrame, bones: s.boneNames(), morphs: s.morphNames() }
out.bez = [0.1, 0.25, 0.5, 0.75, 0.9].map((x) => bezier(x, 0.2, 0.8, 0.6, 0.1))
out.bezIdentity = bezier(0.37, 20 / 127, 20 / 127, 107 / 127, 107 / 127)
out.samples = [0, 7.5, 15, 22.5, 30, 45].map((f) => ({ f, a: s.sampleBone('boneA', f), m: s.sampleMorph('smile', f) }))
// FK with translation: one root bone moved by the sampler
const bones = [{ name: 'boneA', parentIndex: -1, bindTranslation: [0, 1, 0], children: [] }, { name: 'tip', parentIndex: 0, bindTranslation: [0, 2, 0], children: [] }]
const model = new Model(new Float32Array(8), new Uint32Array(3), [], [], { bones, inverseBindMatrices: new Float32Array(32) }, { joints: new Uint16Array(4), weights: new Uint8Array(4) })
model.applySampledFrame(s, 15); model.evaluatePose()
out.world15 = Array.from(model.getBoneWorldMatrices())
console.log(JSON.stringify(out))
