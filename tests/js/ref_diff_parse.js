'use strict'
/* DEV-CONTAINER ONLY: the reference's PmxLoader / VMDLoader (type-erased copies under argv[2]) and host/pmx-loader.js /
 * host/vmd-loader.js parse the same files; every array the deformation path consumes must be identical bit for bit.
 * node ref_diff_parse.js <erased dir> <file.pmx|file.vmd> ... */
const fs = require('fs'), path = require('path')
global.performance = require('perf_hooks').performance
global.fetch = (p) => Promise.resolve({ arrayBuffer: () => { const b = fs.readFileSync(p); return Promise.resolve(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)) } })
const erased = process.argv[2]
const R = { PmxLoader: require(path.join(erased, 'pmx-loader')).PmxLoader, VMDLoader: require(path.join(erased, 'vmd-loader')).VMDLoader }
const M = require(path.join(__dirname, '..', '..', 'reze-engine_amd', 'host'))
const bytes = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)
const must = (c, what) => { if (!c) { console.error('DIVERGED: ' + what); process.exit(1) } }
;(async () => {
  const quiet = console.warn; console.warn = () => {}
  let files = 0
  for (const f of process.argv.slice(3)) {
    if (f.endsWith('.pmx')) {
      const a = await R.PmxLoader.load(f), b = await M.PmxLoader.load(f)
      must(bytes(a.getVertices()).equals(bytes(b.getVertices())), f + ' vertices')
      must(bytes(a.getIndices()).equals(bytes(b.getIndices())), f + ' indices')
      must(bytes(a.getSkinning().joints).equals(bytes(b.getSkinning().joints)), f + ' joints')
      must(bytes(a.getSkinning().weights).equals(bytes(b.getSkinning().weights)), f + ' weights')
      must(bytes(a.getSkeleton().inverseBindMatrices).equals(bytes(b.getSkeleton().inverseBindMatrices)), f + ' inverse bind')
      const ba = a.getSkeleton().bones, bb = b.getSkeleton().bones
      must(ba.length === bb.length, f + ' bone count')
      ba.forEach((x, i) => {
        const y = bb[i]
        must(x.name === y.name && x.parentIndex === y.parentIndex && JSON.stringify(Array.from(x.bindTranslation)) === JSON.stringify(Array.from(y.bindTranslation)), f + ' bone ' + i)
        must((x.appendParentIndex === undefined ? -1 : x.appendParentIndex) === (y.appendParentIndex === undefined || y.appendParentIndex === null ? -1 : y.appendParentIndex) &&
          !!x.appendRotate === !!y.appendRotate && !!x.appendMove === !!y.appendMove && (x.appendRatio === undefined ? 0 : x.appendRatio) === (y.appendRatio === undefined || y.appendRatio === null ? 0 : y.appendRatio), f + ' bone ' + i + ' append data')
      })
      must(a.getMaterials().length === b.getMaterials().length, f + ' material count')
      a.getMaterials().forEach((x, i) => must(x.vertexCount === b.getMaterials()[i].vertexCount && x.edgeSize === b.getMaterials()[i].edgeSize && x.edgeFlag === b.getMaterials()[i].edgeFlag, f + ' material ' + i))
    } else {
      const a = await R.VMDLoader.load(f), b = await M.VMDLoader.load(f)
      must(a.length === b.length, f + ' key-time count')
      a.forEach((x, i) => {
        must(x.time === b[i].time && x.boneFrames.length === b[i].boneFrames.length, f + ' time ' + i)
        x.boneFrames.forEach((k, j) => { const q = b[i].boneFrames[j]; must(k.boneName === q.boneName && k.frame === q.frame && k.rotation.x === q.rotation.x && k.rotation.y === q.rotation.y && k.rotation.z === q.rotation.z && k.rotation.w === q.rotation.w, f + ' key ' + i + '/' + j) })
      })
    }
    files++
  }
  console.warn = quiet
  console.log(JSON.stringify({ files }))
})().catch((e) => { console.error(e); process.exit(1) })
