#!/usr/bin/env python3
"""DEV-TIME ONLY: pin the bone-morph row (PMX morph type 2) to the reference's own primitives, on the reference's own asset.

The reference never applies a bone morph (its loader only skips the section, engine/src/pmx-loader.ts:489-497), so the
SEMANTICS are this build's — the usual MMD ones: local rotation q' = q * slerp(identity, q_morph, w), local translation
t' = t + w * t_morph. Everything those semantics are made of, though, exists in the reference and can be EXECUTED (types
erased, see tools/ref_erased_run.py): Quat.slerp (math.ts:156-189), Quat.multiply (math.ts:77-85), Model.rotateBones +
Model.evaluatePose (model.ts:246-420) and the Mat4 / Vec3 composition of vs() (engine.ts:255-272). One of the reference's
three PMX files carries a bone morph: web/public/models/塞尔凯特/武器.pmx, morph 0 "变形" turns the two blade bones by
-/+29.5 degrees about z and has no translation — so for this asset the WHOLE frame (morph -> local pose -> hierarchy ->
palette -> skin) can be produced by reference code alone:

  entries        parsed by this build's loader (the reference has no parser for them)
  q'             reference Quat.multiply(q, Quat.slerp(identity, q_morph, w))
  world          reference Model.rotateBones(names, q', 0) + evaluatePose()
  skinned        reference Mat4.multiply / Vec3 composed as vs(), every fourth vertex of the mesh

Output: tests/golden/ref_bone_morph.npz (numbers only). tests/test_oracle.py pins the float64 restatement on it, tests/
test_host_js.py the host loader + Model (container only: needs the asset), tests/test_gpu_round2.py the device path.
Nothing here runs on the GPU box; no reference source is written into the repo.
"""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_erased_run import ASSETS, REF, ROOT, erase  # noqa: E402

SCRATCH = "/tmp/ref_erased_bm"
MODEL = "models/塞尔凯特/武器.pmx"

DRIVER = r"""
const fs = require('fs'), path = require('path')
global.performance = require('perf_hooks').performance
global.fetch = (p) => Promise.resolve({ arrayBuffer: () => { const b = fs.readFileSync(p); return Promise.resolve(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)) } })
const { PmxLoader } = require('./pmx-loader'), { Quat, Vec3, Mat4 } = require('./math')
const [pmx, OUT, specFile] = process.argv.slice(2)
const spec = JSON.parse(fs.readFileSync(specFile, 'utf8'))
const dump = (name, ta) => fs.writeFileSync(path.join(OUT, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength))
;(async () => {
  const silent = console.warn; console.warn = () => {}
  const m = await PmxLoader.load(pmx)
  const bones = m.getSkeleton().bones, B = bones.length
  dump('vertices.f32', m.getVertices()); dump('joints.u16', m.getSkinning().joints); dump('weights.u8', m.getSkinning().weights)
  dump('invbind.f32', m.getSkeleton().inverseBindMatrices)
  fs.writeFileSync(path.join(OUT, 'info.json'), JSON.stringify({ parents: bones.map((b) => b.parentIndex), bind: bones.map((b) => b.bindTranslation),
    names: bones.map((b) => b.name), append: bones.filter((b) => b.appendRotate || b.appendMove).length }))
  const v = m.getVertices(), sk = m.getSkinning(), V = m.getVertexCount()
  const sample = []
  for (let i = 0; i < V; i += 4) sample.push(i)
  spec.weights.forEach((w, k) => {
    // the local pose with the morph folded in, by the reference's own quaternion code
    const q = spec.base.map((a) => new Quat(a[0], a[1], a[2], a[3]))
    for (const e of spec.entries) {
      if (w === 0) continue
      q[e.bone] = q[e.bone].multiply(Quat.slerp(new Quat(0, 0, 0, 1), new Quat(e.q[0], e.q[1], e.q[2], e.q[3]), w))
    }
    m.rotateBones(bones.map((b) => b.name), q, 0)
    m.evaluatePose()
    dump('world_' + k + '.f32', m.getBoneWorldMatrices())
    dump('localrot_' + k + '.f32', m.runtimeSkeleton.localRotations)
    const world = m.getBoneWorldMatrices(), ib = m.getSkeleton().inverseBindMatrices, mats = []
    for (let b = 0; b < B; b++) mats.push(new Mat4(world.slice(b * 16, b * 16 + 16)).multiply(new Mat4(ib.slice(b * 16, b * 16 + 16))))
    const out = new Float64Array(sample.length * 6)
    sample.forEach((vi, s) => {
      const wt = [0, 1, 2, 3].map((i) => sk.weights[vi * 4 + i] / 255)
      const sum = wt[0] + wt[1] + wt[2] + wt[3]
      const nw = sum > 0.0001 ? wt.map((x) => x * (1 / sum)) : [1, 0, 0, 0]                 // engine.ts:255-258
      let P = new Vec3(0, 0, 0), N = new Vec3(0, 0, 0)
      for (let i = 0; i < 4; i++) {
        const S = mats[sk.joints[vi * 4 + i]]
        const col = new Float32Array(16)
        col[0] = 1; col[5] = 1; col[10] = 1
        col[12] = v[vi * 8]; col[13] = v[vi * 8 + 1]; col[14] = v[vi * 8 + 2]; col[15] = 1
        P = P.add(S.multiply(new Mat4(col)).getPosition().scale(nw[i]))
        const ncol = new Float32Array(16)
        ncol[12] = v[vi * 8 + 3]; ncol[13] = v[vi * 8 + 4]; ncol[14] = v[vi * 8 + 5]; ncol[15] = 0
        N = N.add(S.multiply(new Mat4(ncol)).getPosition().scale(nw[i]))
      }
      N = N.normalize()
      out.set([P.x, P.y, P.z, N.x, N.y, N.z], s * 6)
    })
    dump('skinned_' + k + '.f64', out)
  })
  console.warn = silent
})().catch((e) => { console.error(e); process.exit(1) })
"""

# this build's loader supplies what the reference skips: the entries of the bone morph
ENTRIES_JS = r"""
const { PmxLoader } = require(process.argv[2] + '/reze-engine_amd/host/pmx-loader.js')
const m = PmxLoader.loadFromBuffer(require('fs').readFileSync(process.argv[3]))
const mo = m.getMorphs(), be = mo.boneEntries
const entries = []
for (let k = 0; k < be.morph.length; k++) entries.push({ morph: be.morph[k], bone: be.bone[k], t: Array.from(be.translation.subarray(k * 3, k * 3 + 3)), q: Array.from(be.rotation.subarray(k * 4, k * 4 + 4)) })
console.log(JSON.stringify({ names: mo.names, types: Array.from(mo.types), entries }))
"""


def main():
    os.makedirs(SCRATCH, exist_ok=True)
    for f in ("math", "model", "pmx-loader", "vmd-loader"):
        js = erase(open(os.path.join(REF, f + ".ts"), encoding="utf-8").read(), f)
        open(os.path.join(SCRATCH, f + ".js"), "w", encoding="utf-8").write(js)
    open(os.path.join(SCRATCH, "driver.js"), "w").write(DRIVER)
    open(os.path.join(SCRATCH, "entries.js"), "w").write(ENTRIES_JS)
    pmx = os.path.join(ASSETS, MODEL)
    mine = json.loads(subprocess.check_output(["node", os.path.join(SCRATCH, "entries.js"), ROOT, pmx]).decode())
    assert mine["types"] == [2] and len(mine["entries"]) == 2 and all(max(abs(x) for x in e["t"]) == 0 for e in mine["entries"]), mine
    rng = np.random.default_rng(0xB0E)
    base = rng.normal(size=(3, 4)) * [0.25, 0.25, 0.25, 0] + [0, 0, 0, 1]
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float32)
    weights = [0.0, 0.35, 1.0]
    spec = dict(base=base.astype(np.float64).tolist(), entries=mine["entries"], weights=weights)
    out = os.path.join(SCRATCH, "out")
    os.makedirs(out, exist_ok=True)
    json.dump(spec, open(os.path.join(SCRATCH, "spec.json"), "w"))
    subprocess.check_call(["node", os.path.join(SCRATCH, "driver.js"), pmx, out, os.path.join(SCRATCH, "spec.json")])
    info = json.load(open(os.path.join(out, "info.json")))
    assert info["append"] == 0
    rd = lambda n, dt: np.fromfile(os.path.join(out, n), dtype=dt)  # noqa: E731
    v = rd("vertices.f32", np.float32).reshape(-1, 8)
    sample = np.arange(0, len(v), 4)
    e = mine["entries"]
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "ref_bone_morph.npz"),
        sample_index=sample.astype(np.int32), vertices=v[sample], joints=rd("joints.u16", np.uint16).reshape(-1, 4)[sample],
        weights=rd("weights.u8", np.uint8).reshape(-1, 4)[sample], inv_bind=rd("invbind.f32", np.float32).reshape(-1, 16),
        parents=np.array(info["parents"], dtype=np.int32), bind=np.array(info["bind"], dtype=np.float64),
        entry_morph=np.array([x["morph"] for x in e], dtype=np.uint32), entry_bone=np.array([x["bone"] for x in e], dtype=np.uint32),
        entry_translation=np.array([x["t"] for x in e], dtype=np.float32), entry_rotation=np.array([x["q"] for x in e], dtype=np.float32),
        base_rotations=base, morph_weights=np.array(weights, dtype=np.float32),
        world=np.stack([rd("world_%d.f32" % k, np.float32).reshape(-1, 16) for k in range(len(weights))]),
        local_rotations=np.stack([rd("localrot_%d.f32" % k, np.float32).reshape(-1, 4) for k in range(len(weights))]),
        skinned=np.stack([rd("skinned_%d.f64" % k, np.float64).reshape(-1, 6).astype(np.float32) for k in range(len(weights))]))
    print("written tests/golden/ref_bone_morph.npz:", os.path.getsize(os.path.join(ROOT, "tests", "golden", "ref_bone_morph.npz")), "bytes")


if __name__ == "__main__":
    main()
