'use strict'
// Source: host/src/index.ts (TypeScript). host/index.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
// Same export list as the reference's engine/src/index.ts:1-2 (+ loaders and Model for host-side use).
const { Engine } = require('./engine')
const { Vec3, Quat, Mat4 } = require('./math')
const { Model } = require('./model')
const { PmxLoader } = require('./pmx-loader')
const { VMDLoader } = require('./vmd-loader')
const { VMDSampler } = require('./vmd-sampler')
module.exports = { Engine, Vec3, Quat, Mat4, Model, PmxLoader, VMDLoader, VMDSampler }
