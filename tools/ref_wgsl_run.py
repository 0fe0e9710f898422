#!/usr/bin/env python3
"""DEV-TIME ONLY: evaluate the reference's own WGSL TEXT for the hot path on real inputs and store the results as fixtures.

  palette   the skin-matrix compute shader's `fn main` (engine/src/engine.ts:919-928) — skinMatrices[b] = worldMat * invBindMat
  skinning  the body of `@vertex fn vs(...)` (engine/src/engine.ts:245-276) up to its `return` — weight renormalisation,
            4-bone LBS of position and normal, normalize; the one statement that needs the camera uniforms
            (`output.position = ...`) is reported as skipped and is not part of the deformation path

Both are read out of /root/reference/engine/src/engine.ts at run time and interpreted by tools/wgsl_eval.py (binary32 per
operation, the association documented there). Inputs: the reference-produced world matrices / inverse bind / vertex, joint and
weight slices already held in tests/golden/ref_c1_pose0.npz (tools/ref_erased_run.py). Output: tests/golden/ref_wgsl.npz —
palettes [349,16] and deformed (position, normal) [N,6] for the 256-vertex slices and for every 28th vertex of the model, under
three poses, plus the SHA-256 of the two shader bodies they came from. No shader text is stored, only numbers.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import wgsl_eval as W  # noqa: E402

ENGINE_TS = "/root/reference/engine/src/engine.ts"
F = np.float32


def main():
    src = open(ENGINE_TS, encoding="utf-8").read()
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_c1_pose0.npz"))
    vs_body, vs_head = W.function_body(src, r"@vertex\s+fn\s+vs\s*\(")
    cs_body, cs_head = W.function_body(src, r"fn\s+main\s*\(\s*@builtin\(global_invocation_id\)")
    ol_body, ol_head = W.function_body(src, r"@vertex\s+fn\s+vs\s*\(", containing="expandedPos")      # the outline pass (engine.ts:431-463)
    assert "skinMats" in vs_body and "normalizedWeights" in vs_body and "skinMatrices[boneIndex]" in cs_body and "expandedPos" not in vs_body
    ol_tokens = W.tokenize(ol_body)
    out = {"vs_sha256": hashlib.sha256(vs_body.encode()).hexdigest(), "cs_sha256": hashlib.sha256(cs_body.encode()).hexdigest()}
    vs_tokens, cs_tokens = W.tokenize(vs_body), W.tokenize(cs_body)
    B = len(g["inv_bind"])
    as_mats = lambda a: [W.Mat(m.reshape(4, 4)) for m in np.asarray(a, dtype=F)]   # noqa: E731  column-major: row of the reshape = column
    skipped_all = set()
    with np.errstate(all="ignore"):
        for pose in ("pose0", "tween150", "tween500"):
            world, ib = as_mats(g["world_" + pose]), as_mats(g["inv_bind"])
            skin = [None] * B
            for b in range(B):                                           # one invocation per bone, like the dispatch
                env = {"globalId": np.array([b, 0, 0], dtype=np.uint32), "boneCount": {"count": B}, "worldMatrices": world,
                       "inverseBindMatrices": ib, "skinMatrices": skin}
                it = W.Interp(cs_tokens, env)
                it.run()
                skipped_all.update(it.skipped)
            # an invocation past the bone count must return before touching the arrays
            it = W.Interp(cs_tokens, {"globalId": np.array([B, 0, 0], dtype=np.uint32), "boneCount": {"count": B}, "worldMatrices": world,
                                      "inverseBindMatrices": ib, "skinMatrices": skin})
            it.run()
            pal = np.stack([np.concatenate(m.cols) for m in skin]).astype(F)
            out["palette_" + pose] = pal
            for tag, vkey, jkey, wkey in (("slice", "slice_vertices", "slice_joints", "slice_weights"), ("wide", "wide_vertices", "wide_joints", "wide_weights")):
                v, joints, weights = g[vkey], g[jkey], g[wkey]
                res = np.zeros((len(v), 6), dtype=F)
                for k in range(len(v)):
                    env = {"position": v[k, 0:3].astype(F), "normal": v[k, 3:6].astype(F), "uv": v[k, 6:8].astype(F),
                           "joints0": joints[k].astype(np.uint32), "weights0": (weights[k].astype(F) / F(255.0)).astype(F),   # unorm8x4 (engine.ts:354-355)
                           "skinMats": skin}
                    it = W.Interp(vs_tokens, env)
                    e = it.run()
                    skipped_all.update(it.skipped)
                    o = e["__return__"]
                    res[k, 0:3] = o["worldPos"]
                    res[k, 3:6] = o["normal"]
                out["%s_%s" % (tag, pose)] = res
            # the outline pass's vs(): the same skinning, then  expandedPos = worldPos + worldNormal * material.edgeSize * 0.01
            v, joints, weights = g["wide_vertices"], g["wide_joints"], g["wide_weights"]
            edge = np.array([0.0, 0.4, 1.0, 1.5], dtype=F)[np.arange(len(v)) % 4]      # materials without / with an outline
            hull = np.zeros((len(v), 3), dtype=F)
            for k in range(len(v)):
                env = {"position": v[k, 0:3].astype(F), "normal": v[k, 3:6].astype(F), "joints0": joints[k].astype(np.uint32),
                       "weights0": (weights[k].astype(F) / F(255.0)).astype(F), "skinMats": skin, "material": {"edgeSize": F(edge[k])}}
                it = W.Interp(ol_tokens, env)
                e = it.run()
                skipped_all.update(it.skipped)
                hull[k] = e["expandedPos"]
            out["hull_wide_" + pose] = hull
            out["hull_edge"] = edge
            print(pose, "done")
    out["ol_sha256"] = hashlib.sha256(ol_body.encode()).hexdigest()
    out["skipped_statements"] = np.array(sorted(skipped_all))
    print("statements skipped for lack of bindings:", sorted(skipped_all))
    assert all(s.startswith("output.position") for s in skipped_all), skipped_all
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_wgsl.npz"), **out)
    print("written tests/golden/ref_wgsl.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "ref_wgsl.npz")), "bytes")


if __name__ == "__main__":
    main()
