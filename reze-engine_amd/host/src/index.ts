'use strict'
// Source: host/src/index.ts (TypeScript). host/index.js is that file with its types erased (tools/ts_erase.py; no tsc in the image) - edit the .ts.
// Same export list as the reference's engine/src/index.ts:1-2 (+ loaders and Model for host-side use).
import { Engine } from './engine'
import { Vec3, Quat, Mat4 } from './math'
import { Model } from './model'
import { PmxLoader } from './pmx-loader'
import { VMDLoader } from './vmd-loader'
import { VMDSampler } from './vmd-sampler'
export { Engine, Vec3, Quat, Mat4, Model, PmxLoader, VMDLoader, VMDSampler }
