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
// the flattened form the device sampler consumes (rz_upload_animation): bone 'tip' is not keyed, 'boneA' is;
// morph set = [vertex 'smile', group 'grp' -> smile x0.5 twice, vertex 'other' (never keyed)]
const flat = s.flatten({ boneA: 0, tip: 1 }, { names: ['smile', 'grp', 'other'], types: [1, 0, 1], groups: [null, [[0, 0.5], [0, 0.25]], null] })
out.flat = {}
for (const k2 of Object.keys(flat)) out.flat[k2] = Array.from(flat[k2])
console.log(JSON.stringify(out))
