#!/usr/bin/env python3
"""DEV-TIME ONLY: run the reference's own host code to pin this build's host side.

The reference (TypeScript) cannot run as-is here: the image has no `tsc`, and Node 12 rejects the
`??` the sources use. TypeScript's types are erasable — removing annotations, `interface`/`type`
blocks, `as` casts and access modifiers leaves the program's semantics untouched — so this script
erases them mechanically from four files of /root/reference/engine/src (math.ts, model.ts,
pmx-loader.ts, vmd-loader.ts), writes the result to a scratch directory OUTSIDE the repo, runs it
under Node on the reference's own assets, and stores only small numeric fixtures under
tests/golden/:

  ref_c1_pose0.npz   real demo model (web/public/models/塞尔凯特2/塞尔凯特2.pmx) + pool.vmd frame 0,
                     applied the way Engine.playAnimation does (engine.ts:1474-1505): world matrices
                     [349,16], inverse bind, CRC32s of every parsed array, 256-vertex slices; plus the HOT-PATH
                     pins: the palette world x inverseBind through the reference's Mat4.multiply (math.ts:303-320)
                     and the slice vertices skinned with the reference's Mat4 / Vec3 primitives composed as vs()
                     (engine.ts:255-272), for pose0 and the mid-tween pose.
  ref_models.json    per-asset counts + CRC32s of joints / weights / vertex buffer / inverse bind
                     for the three PMX files and key counts of the two VMD files.

tests/test_host_js.py then checks that host/*.js reproduces these numbers bit for bit. No reference
source or asset is copied into the repo; nothing here runs on the GPU box.
"""
import json
import os
import re
import subprocess
import sys
import zlib

import numpy as np

REF = "/root/reference/engine/src"
ASSETS = "/root/reference/web/public"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = "/tmp/ref_erased"


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def strip_param(p):
    p = p.strip()
    p = re.sub(r"^(private|public|protected|readonly)\s+", "", p)
    m = re.match(r"^(\.\.\.)?(\w+)\??\s*:\s*(.*)$", p, re.S)
    if not m:
        return p
    rest = m.group(3)
    # default value: first " = " that is not part of "=>"
    dm = re.search(r"\s=\s(?!>)", rest)
    default = rest[dm.end():] if dm else None
    name = (m.group(1) or "") + m.group(2)
    return name + (" = " + default if default is not None else "")


def strip_params(s):
    return ", ".join(strip_param(p) for p in split_top(s))


def erase(src, name):
    # multi-line tuple casts:  ] as [\n number, ... \n ]
    src = re.sub(r"\s+as\s+\[[^\]]*\]", "", src)
    src = re.sub(r"\s+as\s+[A-Z]\w*(<[^>]*>)?", "", src)
    src = re.sub(r"new Array<[^>]+>\(", "new Array(", src)
    lines = src.split("\n")
    out = []
    exported = []
    i = 0
    in_params = False
    while i < len(lines):
        ln = lines[i]
        code, cmt = ln, ""
        ci = ln.find("//")
        if ci >= 0 and ln[:ci].count('"') % 2 == 0 and ln[:ci].count("`") % 2 == 0:
            code, cmt = ln[:ci], ln[ci:]
        st = code.strip()
        # imports
        m = re.match(r'^import\s+\{([^}]*)\}\s+from\s+"(\./\w[\w-]*)"', st)
        if m:
            if m.group(2) == "./physics":
                i += 1
                continue
            names = [n.strip() for n in m.group(1).split(",")]
            out.append("const { %s } = require(\"%s\")" % (", ".join(names), m.group(2)))
            i += 1
            continue
        # interface / type blocks
        if re.match(r"^(export\s+)?interface\s+\w+", st) or re.match(r"^(export\s+)?type\s+\w+\s*=\s*\{", st):
            depth = 0
            while True:
                depth += lines[i].count("{") - lines[i].count("}")
                i += 1
                if depth <= 0:
                    break
            continue
        m = re.match(r"^export\s+(class|function)\s+(\w+)", st)
        if m:
            exported.append(m.group(2))
            code = code.replace("export ", "", 1)
            st = code.strip()
        indent = len(code) - len(code.lstrip())
        # class member modifiers
        code = re.sub(r"^(\s*)((private|public|protected|readonly)\s+)+", r"\1", code)
        st = code.strip()
        # class fields (2-space indent, inside a class)
        if indent == 2 and not in_params:
            m = re.match(r"^(\w+)[!?]?\s*:\s*[^=({]+$", st)
            if m:
                i += 1
                continue
            m = re.match(r"^(\w+)[!?]?\s*:\s*[^=]+?\s=\s(.*)$", st)
            if m and "(" not in st.split("=")[0]:
                out.append("  %s = %s%s" % (m.group(1), m.group(2), cmt))
                i += 1
                continue
        # multi-line parameter lists: "name(" ... ") {" / "): T {"
        if in_params:
            if re.match(r"^\)\s*(:\s*[^={]+)?\s*\{$", st):
                out.append(" " * indent + ") {" + cmt)
                in_params = False
            else:
                had_comma = st.endswith(",")
                out.append(" " * indent + strip_param(st.rstrip(",")) + ("," if had_comma else "") + cmt)
            i += 1
            continue
        if re.match(r"^(static\s+)?(async\s+)?\w+\($", st) or re.match(r"^constructor\($", st):
            in_params = True
            out.append(code + cmt)
            i += 1
            continue
        # single-line method / function signatures
        m = re.match(r"^(\s*)((?:static\s+)?(?:async\s+)?(?:function\s+)?\w+)\((.*)\)\s*(?::\s*[^={]+?)?\s*\{$", code)
        if m and not re.match(r"^\s*(if|for|while|switch|catch|return|else)\b", code) and "=>" not in code:
            out.append("%s%s(%s) {%s" % (m.group(1), m.group(2), strip_params(m.group(3)), cmt))
            i += 1
            continue
        # typed arrow functions:  (i: number): void => {
        code = re.sub(r"\(((?:\w+\s*:\s*[\w\[\]<>| ]+,?\s*)+)\)\s*(?::\s*[\w\[\]<>| ]+)?\s*=>",
                      lambda mm: "(" + strip_params(mm.group(1)) + ") =>", code)
        # typed variable declarations
        code = re.sub(r"\b(const|let)\s+(\w+)\s*:\s*[^=]+?\s=\s", r"\1 \2 = ", code)
        code = re.sub(r"\b(let)\s+(\w+)\s*:\s*[\w\[\]<>| ]+$", r"\1 \2", code)
        # nullish coalescing (the two occurrences have simple operands)
        code = re.sub(r"([\w\.\[\]]+)\s\?\?\s(-?\w+)", r"(\1 !== undefined && \1 !== null ? \1 : \2)", code)
        out.append(code + cmt)
        i += 1
    body = "\n".join(out)
    body += "\n" + "\n".join("module.exports.%s = %s" % (e, e) for e in exported) + "\n"
    return body


DRIVER = r"""
const fs = require('fs'), path = require('path'), zlib = require('zlib')
global.performance = require('perf_hooks').performance
global.fetch = (p) => Promise.resolve({ arrayBuffer: () => { const b = fs.readFileSync(p); return Promise.resolve(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)) } })
const { PmxLoader } = require('./pmx-loader'), { VMDLoader } = require('./vmd-loader'), { Quat, Vec3, Mat4 } = require('./math')
const ASSETS = process.argv[2], OUT = process.argv[3]
const crc = (ta) => zlib.crc32 ? zlib.crc32(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)) : null
const dump = (name, ta) => fs.writeFileSync(path.join(OUT, name), Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength))
// Hot-path pins executed by the REFERENCE's own math.ts primitives on the reference's own model + pose:
//   palette   skinMatrices[b] = worldMatrices[b] * inverseBindMatrices[b]  (engine.ts:926-928) through Mat4.multiply
//             (math.ts:303-320) — doubles with an f32 store, the WGSL does the same sum in f32
//   skinned   vs() (engine.ts:255-272) of the 256 slice vertices: every M_i * vec4(p, 1) and M_i * vec4(n, 0) is a
//             Mat4.multiply against a matrix whose last column holds the vector; the weighted sum uses Vec3.scale / add and
//             the normal Vec3.normalize. For a BDEF1 vertex (weights 255,0,0,0) the result is Mat4.multiply alone.
function pinHotPath(m, tag) {
  const world = m.getBoneWorldMatrices(), ib = m.getSkeleton().inverseBindMatrices
  const B = m.getSkeleton().bones.length
  const pal = new Float32Array(B * 16), mats = []
  for (let b = 0; b < B; b++) {
    const S = new Mat4(world.slice(b * 16, b * 16 + 16)).multiply(new Mat4(ib.slice(b * 16, b * 16 + 16)))
    pal.set(S.values, b * 16)
    mats.push(S)
  }
  dump('m2_palette_' + tag + '.f32', pal)
  const v = m.getVertices(), sk = m.getSkinning(), V = m.getVertexCount()
  const idx = []
  for (let i = 0; i < 128; i++) idx.push(i)
  for (let i = 14000; i < 14064; i++) idx.push(i)
  for (let i = V - 64; i < V; i++) idx.push(i)
  const out = new Float64Array(idx.length * 6)
  idx.forEach((vi, k) => {
    const w = [0, 1, 2, 3].map((i) => sk.weights[vi * 4 + i] / 255)
    const sum = w[0] + w[1] + w[2] + w[3]
    const inv = sum > 0.0001 ? 1 / sum : 1                       // engine.ts:255-257
    const nw = sum > 0.0001 ? w.map((x) => x * inv) : [1, 0, 0, 0]
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
    out.set([P.x, P.y, P.z, N.x, N.y, N.z], k * 6)
  })
  dump('m2_skinned_' + tag + '.f64', out)
  // a WIDE sample: every 28th vertex of the model (1 031 vertices over every body part, all three influence types)
  const wide = []
  for (let i = 0; i < V; i += 28) wide.push(i)
  const wout = new Float64Array(wide.length * 6)
  wide.forEach((vi, k) => {
    const w = [0, 1, 2, 3].map((i) => sk.weights[vi * 4 + i] / 255)
    const sum = w[0] + w[1] + w[2] + w[3]
    const inv = sum > 0.0001 ? 1 / sum : 1
    const nw = sum > 0.0001 ? w.map((x) => x * inv) : [1, 0, 0, 0]
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
    wout.set([P.x, P.y, P.z, N.x, N.y, N.z], k * 6)
  })
  dump('m2_skinnedwide_' + tag + '.f64', wout)
}
;(async () => {
  const silent = console.warn; console.warn = () => {}
  const models = { 'models/塞尔凯特2/塞尔凯特2.pmx': 'm2', 'models/塞尔凯特/塞尔凯特.pmx': 'm1', 'models/塞尔凯特/武器.pmx': 'w' }
  const info = {}
  for (const rel of Object.keys(models)) {
    const tag = models[rel]
    const m = await PmxLoader.load(path.join(ASSETS, rel))
    dump(tag + '_vertices.f32', m.getVertices()); dump(tag + '_joints.u16', m.getSkinning().joints)
    dump(tag + '_weights.u8', m.getSkinning().weights); dump(tag + '_invbind.f32', m.getSkeleton().inverseBindMatrices)
    dump(tag + '_indices.u32', m.getIndices())
    const bones = m.getSkeleton().bones
    info[tag] = { verts: m.getVertexCount(), indices: m.getIndices().length, bones: bones.length,
      append: bones.filter((b) => b.appendRotate || b.appendMove).length, materials: m.getMaterials().length,
      rigidbodies: m.getRigidbodies().length, joints: m.getJoints().length,
      boneNames: bones.map((b) => b.name), parents: bones.map((b) => b.parentIndex),
      bind: bones.map((b) => b.bindTranslation), appendParent: bones.map((b) => b.appendParentIndex === undefined ? -1 : b.appendParentIndex),
      appendRatio: bones.map((b) => b.appendRatio === undefined ? 0 : b.appendRatio),
      appendRotate: bones.map((b) => !!b.appendRotate), appendMove: bones.map((b) => !!b.appendMove) }
    if (tag === 'm2') {
      // Engine.playAnimation's frame-0 application (engine.ts:1474-1505) on the reference's own Model
      const frames = await VMDLoader.load(path.join(ASSETS, 'animations/pool.vmd'))
      const byBone = new Map()
      for (const kf of frames) for (const bf of kf.boneFrames) { if (!byBone.has(bf.boneName)) byBone.set(bf.boneName, []); byBone.get(bf.boneName).push({ time: kf.time, rotation: bf.rotation }) }
      const names0 = [], rots0 = [], has0 = new Set()
      for (const [n, ks] of byBone.entries()) { ks.sort((a, b) => a.time - b.time); if (ks[0].time === 0) { names0.push(n); rots0.push(ks[0].rotation); has0.add(n) } }
      m.rotateBones(names0, rots0, 0)
      const reset = bones.map((b) => b.name).filter((n) => !has0.has(n))
      m.rotateBones(reset, reset.map(() => new Quat(0, 0, 0, 1)), 0)
      m.evaluatePose()
      dump('m2_world_pose0.f32', m.getBoneWorldMatrices())
      dump('m2_localrot_pose0.f32', m.runtimeSkeleton.localRotations)
      pinHotPath(m, 'pose0')
      info.pool = { keyTimes: frames.map((f) => [Math.round(f.time * 30), f.boneFrames.length]), bones0: names0 }
      // a second pose exercising tweens: a 400 ms tween sampled at +150 ms through the reference's own clock
      let now = 1000
      global.performance = { now: () => now }
      m.rotateBones(['センター', '上半身', '首'], [new Quat(0.1, 0.2, 0.05, 0.97), new Quat(-0.2, 0.1, 0.0, 0.97), new Quat(0.0, -0.3, 0.1, 0.95)], 400)
      now = 1150
      m.evaluatePose()
      dump('m2_world_tween150.f32', m.getBoneWorldMatrices())
      dump('m2_localrot_tween150.f32', m.runtimeSkeleton.localRotations)
      pinHotPath(m, 'tween150')
      now = 1500
      m.evaluatePose()
      dump('m2_world_tween500.f32', m.getBoneWorldMatrices())
      pinHotPath(m, 'tween500')
      global.performance = require('perf_hooks').performance
    }
  }
  const boom = await VMDLoader.load(path.join(ASSETS, 'animations/boom.vmd'))
  info.boom = { keyTimes: boom.map((f) => [Math.round(f.time * 30), f.boneFrames.length]) }
  fs.writeFileSync(path.join(OUT, 'info.json'), JSON.stringify(info))
  console.warn = silent
})().catch((e) => { console.error(e); process.exit(1) })
"""


def main():
    os.makedirs(SCRATCH, exist_ok=True)
    for f in ("math", "model", "pmx-loader", "vmd-loader"):
        js = erase(open(os.path.join(REF, f + ".ts"), encoding="utf-8").read(), f)
        open(os.path.join(SCRATCH, f + ".js"), "w", encoding="utf-8").write(js)
        subprocess.check_call(["node", "--check", os.path.join(SCRATCH, f + ".js")])
    open(os.path.join(SCRATCH, "driver.js"), "w").write(DRIVER)
    out = os.path.join(SCRATCH, "out")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["node", os.path.join(SCRATCH, "driver.js"), ASSETS, out])
    info = json.load(open(os.path.join(out, "info.json")))
    rd = lambda n, dt: np.fromfile(os.path.join(out, n), dtype=dt)  # noqa: E731
    crc = lambda a: zlib.crc32(a.tobytes()) & 0xFFFFFFFF  # noqa: E731
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    summary = {}
    for tag in ("m2", "m1", "w"):
        v = rd(tag + "_vertices.f32", np.float32)
        j = rd(tag + "_joints.u16", np.uint16)
        w = rd(tag + "_weights.u8", np.uint8)
        ib = rd(tag + "_invbind.f32", np.float32)
        idx = rd(tag + "_indices.u32", np.uint32)
        d = info[tag]
        summary[tag] = dict(verts=d["verts"], indices=d["indices"], bones=d["bones"], append=d["append"],
                            materials=d["materials"], rigidbodies=d["rigidbodies"], joints=d["joints"],
                            crc_vertices=crc(v), crc_joints=crc(j), crc_weights=crc(w), crc_invbind=crc(ib),
                            crc_indices=crc(idx))
    summary["pool"] = dict(keyTimes=info["pool"]["keyTimes"], n_bones0=len(info["pool"]["bones0"]))
    summary["boom"] = info["boom"]
    json.dump(summary, open(os.path.join(gold, "ref_models.json"), "w"), indent=1, sort_keys=True)
    d = info["m2"]
    v = rd("m2_vertices.f32", np.float32).reshape(-1, 8)
    sl = np.r_[0:128, 14000:14064, len(v) - 64:len(v)]       # 256-vertex slices (numbers, not the model)
    wide = np.arange(0, len(v), 28)                          # every 28th vertex (1 031 of 28 842): a thin sample of every body part
    v = v.copy(); v[:, 6:8] = 0.0                           # texture coordinates never pass through the deformation: not stored
    np.savez_compressed(
        os.path.join(gold, "ref_c1_pose0.npz"),
        world_pose0=rd("m2_world_pose0.f32", np.float32).reshape(-1, 16),
        world_tween150=rd("m2_world_tween150.f32", np.float32).reshape(-1, 16),
        world_tween500=rd("m2_world_tween500.f32", np.float32).reshape(-1, 16),
        local_rot_pose0=rd("m2_localrot_pose0.f32", np.float32).reshape(-1, 4),
        local_rot_tween150=rd("m2_localrot_tween150.f32", np.float32).reshape(-1, 4),
        bone_names=np.array(d["boneNames"]),
        inv_bind=rd("m2_invbind.f32", np.float32).reshape(-1, 16),
        parents=np.array(d["parents"], dtype=np.int32), bind=np.array(d["bind"], dtype=np.float64),
        append_parent=np.array(d["appendParent"], dtype=np.int32), append_ratio=np.array(d["appendRatio"], dtype=np.float64),
        append_rotate=np.array(d["appendRotate"]), append_move=np.array(d["appendMove"]),
        palette_pose0=rd("m2_palette_pose0.f32", np.float32).reshape(-1, 16),
        palette_tween150=rd("m2_palette_tween150.f32", np.float32).reshape(-1, 16),
        skinned_pose0=rd("m2_skinned_pose0.f64", np.float64).reshape(-1, 6),
        skinned_tween150=rd("m2_skinned_tween150.f64", np.float64).reshape(-1, 6),
        palette_tween500=rd("m2_palette_tween500.f32", np.float32).reshape(-1, 16),
        skinned_tween500=rd("m2_skinned_tween500.f64", np.float64).reshape(-1, 6),
        wide_index=wide.astype(np.int32), wide_vertices=v[wide],
        wide_joints=rd("m2_joints.u16", np.uint16).reshape(-1, 4)[wide], wide_weights=rd("m2_weights.u8", np.uint8).reshape(-1, 4)[wide],
        skinnedwide_pose0=rd("m2_skinnedwide_pose0.f64", np.float64).reshape(-1, 6).astype(np.float32),
        skinnedwide_tween150=rd("m2_skinnedwide_tween150.f64", np.float64).reshape(-1, 6).astype(np.float32),
        skinnedwide_tween500=rd("m2_skinnedwide_tween500.f64", np.float64).reshape(-1, 6).astype(np.float32),
        slice_index=sl.astype(np.int32), slice_vertices=v[sl],
        slice_joints=rd("m2_joints.u16", np.uint16).reshape(-1, 4)[sl],
        slice_weights=rd("m2_weights.u8", np.uint8).reshape(-1, 4)[sl])
    print("fixtures written:", os.listdir(gold))


if __name__ == "__main__":
    main()
