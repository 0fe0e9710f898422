"""CPU tests of the oracle itself: known-answer tests that pin the restatement of
engine/src/engine.ts:253-272 / :926-928, and the bit-exact C <-> NumPy cross-check."""
import numpy as np
import pytest

from reze_engine_amd import synth


def rot_z(angle, t=(0, 0, 0)):
    c, s = np.cos(angle), np.sin(angle)
    m = np.zeros(16, dtype=np.float32)
    m[0], m[1], m[4], m[5], m[10], m[15] = c, s, -s, c, 1, 1
    m[12:15] = t
    return m


def ident():
    m = np.zeros(16, dtype=np.float32)
    m[0] = m[5] = m[10] = m[15] = 1
    return m


def test_palette_matches_column_major_product(oracle):
    rng = np.random.default_rng(1)
    W = rng.normal(size=(7, 16)).astype(np.float32)
    IB = rng.normal(size=(7, 16)).astype(np.float32)
    S = oracle.palette(W, IB)
    for b in range(7):
        ref = (W[b].reshape(4, 4).T.astype(np.float64) @ IB[b].reshape(4, 4).T.astype(np.float64)).T.reshape(16)
        np.testing.assert_allclose(S[b], ref, rtol=2e-6, atol=2e-6)
    # bit-exact against the NumPy twin
    assert np.array_equal(S, oracle.np_twin.palette(W, IB))


def test_identity_pose_returns_rest_mesh(oracle):
    """inverse bind is translation-only (pmx-loader.ts:818-822), so identity rotations give
    skin = T(b) * T(-b) = I and the deformed mesh equals the rest mesh."""
    mesh = synth.make_mesh(4096, 64, seed=3)
    quats = np.zeros((64, 4), dtype=np.float32)
    quats[:, 3] = 1
    world = synth.fk_world(mesh["parents"], mesh["bind"], quats)
    S = oracle.palette(world, mesh["inv_bind"])
    eye = np.tile(ident(), (64, 1))
    np.testing.assert_allclose(S, eye, atol=2e-6)
    pos, nrm = oracle.skin(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], eye)
    # weights are u8/255 renormalised by their f32 sum: exact up to a couple of ulp
    np.testing.assert_allclose(pos, mesh["pos"], rtol=4e-7, atol=1e-6)
    np.testing.assert_allclose(nrm, mesh["nrm"], rtol=0, atol=3e-7)


def test_single_bone_rigid_rotation(oracle):
    """BDEF1 vertices under one bone rotating about a pivot: P' = R (p - c) + c exactly (to f32)."""
    rng = np.random.default_rng(5)
    V = 257
    pos = rng.uniform(-5, 5, size=(V, 3)).astype(np.float32)
    nrm = rng.normal(size=(V, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    joints = np.zeros((V, 4), dtype=np.uint16)
    weights = np.zeros((V, 4), dtype=np.uint8)
    weights[:, 0] = 255
    pivot = np.array([1.5, 2.0, -0.5])
    ang = 0.7
    world = rot_z(ang, pivot)[None]                    # W = T(c) R
    ib = ident()[None].copy()
    ib[0, 12:15] = -pivot                              # IB = T(-c)
    S = oracle.palette(world, ib)
    p, n = oracle.skin(pos, nrm, joints, weights, S)
    R = rot_z(ang).reshape(4, 4).T[:3, :3].astype(np.float64)
    ref_p = (pos.astype(np.float64) - pivot) @ R.T + pivot
    ref_n = nrm.astype(np.float64) @ R.T
    np.testing.assert_allclose(p, ref_p, atol=5e-6)
    np.testing.assert_allclose(n, ref_n, atol=5e-7)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=3e-7)


def test_two_bone_blend_hand_computed(oracle):
    """weights (64,191): w = u8/255 then / f32 sum; position is the weighted mean of the two bone transforms."""
    pos = np.array([[1.0, 2.0, 3.0]], dtype=np.float32)
    nrm = np.array([[0.0, 1.0, 0.0]], dtype=np.float32)
    joints = np.array([[1, 0, 0, 0]], dtype=np.uint16)
    weights = np.array([[64, 191, 0, 0]], dtype=np.uint8)
    S = np.stack([ident(), ident()])
    S[0, 12:15] = [10, 0, 0]       # bone 0 translates +10 x
    S[1, 12:15] = [0, -4, 0]       # bone 1 translates -4 y
    p, n = oracle.skin(pos, nrm, joints, weights, S)
    w0, w1 = np.float32(64) / np.float32(255), np.float32(191) / np.float32(255)
    s = np.float32(w0 + w1)
    w0, w1 = w0 * (np.float32(1) / s), w1 * (np.float32(1) / s)
    ref = np.float64(w0) * np.array([1, -2, 3.0]) + np.float64(w1) * np.array([11, 2, 3.0])
    np.testing.assert_allclose(p[0], ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(n[0], [0, 1, 0], atol=1e-7)


def test_zero_weight_sum_and_zero_normal_are_defined(oracle):
    pos = np.array([[1.0, 2.0, 3.0], [1.0, 2.0, 3.0]], dtype=np.float32)
    nrm = np.array([[0.0, 0.0, 0.0], [0.0, 0.6, 0.8]], dtype=np.float32)
    joints = np.array([[1, 0, 0, 0], [1, 0, 0, 0]], dtype=np.uint16)
    weights = np.zeros((2, 4), dtype=np.uint8)         # sum 0 -> select((1,0,0,0)) branch, engine.ts:257
    S = np.stack([ident(), ident()])
    S[1, 12:15] = [5, 5, 5]
    p, n = oracle.skin(pos, nrm, joints, weights, S)
    np.testing.assert_array_equal(p, [[6, 7, 8], [6, 7, 8]])
    np.testing.assert_array_equal(n[0], [0, 0, 0])     # zero-length normal: rest normal returned (build-defined)
    np.testing.assert_allclose(n[1], [0, 0.6, 0.8], atol=1e-7)
    assert np.isfinite(p).all() and np.isfinite(n).all()


@pytest.mark.parametrize("V,B,M", [(1, 1, 0), (1000, 40, 0), (3001, 200, 5), (4096, 256, 64)])
def test_c_and_numpy_twins_agree_bit_for_bit(oracle, V, B, M):
    mesh = synth.make_mesh(V, B, seed=V + B)
    deltas = mw = None
    if M:
        deltas, mw = synth.make_morphs_dense(V, M, seed=M)
        mw[1] = 0.0                                     # a skipped morph
    pc, nc = oracle.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                           mesh["inv_bind"], deltas, mw, threads=3)
    pn, nn = oracle.np_twin.deform(mesh["pos"], mesh["nrm"], mesh["joints"], mesh["weights"], mesh["world"],
                                   mesh["inv_bind"], deltas, mw)
    assert np.array_equal(pc, pn)
    assert np.array_equal(nc, nn)
    # threaded whole-frame driver == the single calls
    S = oracle.palette(mesh["world"], mesh["inv_bind"])
    p0 = mesh["pos"] if not M else oracle.morph_dense(deltas, mw, mesh["pos"])
    p1, n1 = oracle.skin(p0, mesh["nrm"], mesh["joints"], mesh["weights"], S)
    assert np.array_equal(pc, p1) and np.array_equal(nc, n1)


def test_sparse_morph_equals_dense_expansion(oracle):
    V, M = 5000, 24
    off, idx, d3, w = synth.make_morphs_sparse(V, M, density=0.03, seed=9)
    w[3] = 0
    pos = synth.make_mesh(V, 8, seed=2)["pos"]
    dense = synth.sparse_to_dense(V, off, idx, d3)
    a = oracle.morph_sparse(V, off, idx, d3, w, pos)
    b = oracle.morph_dense(dense, w, pos)
    # dense adds w*0 for untouched vertices (exact) so the two forms agree bit for bit
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle.np_twin.morph_sparse(V, off, idx, d3, w, pos))


def test_synthetic_mesh_respects_loader_invariants():
    """pmx-loader.ts:855-951 guarantees weights sum to exactly 255 and joints < boneCount."""
    mesh = synth.make_mesh(30000, 200)
    assert (mesh["weights"].astype(np.int64).sum(axis=1) == 255).all()
    assert mesh["joints"].max() < 200
    assert (mesh["parents"][1:] < np.arange(1, 200)).all()
    np.testing.assert_allclose(np.linalg.norm(mesh["nrm"], axis=1), 1.0, atol=1e-6)
    # world matrices are affine (bottom row 0,0,0,1)
    np.testing.assert_array_equal(mesh["world"][:, [3, 7, 11, 15]], np.tile([0, 0, 0, 1], (200, 1)))


def test_outline_hull_twins_agree(oracle):
    rng = np.random.default_rng(4)
    p = rng.normal(size=(500, 3)).astype(np.float32)
    n = rng.normal(size=(500, 3)).astype(np.float32)
    e = rng.uniform(0, 2, size=500).astype(np.float32)
    a = oracle.hull(p, n, e)
    assert np.array_equal(a, oracle.np_twin.hull(p, n, e))
    np.testing.assert_allclose(a, p + n * e[:, None] * 0.01, rtol=1e-6, atol=1e-6)


# ---- pins to REFERENCE EXECUTION (tests/golden/ref_c1_pose0.npz, written by tools/ref_erased_run.py) ----
def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_c1_pose0.npz"))


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_palette_pinned_to_the_reference_mat4_multiply(oracle, pose):
    """engine.ts:926-928 (skin = world * inverseBind) as computed by the reference's own Mat4.multiply (math.ts:303-320) on
    the real 349-bone model. The reference sums in doubles and stores f32 (<= 0.5 ulp of the result); the oracle (like the
    WGSL) rounds after each of the 4 products-and-adds. Bound per element: 4 * 2^-24 * SUM_k |a_k b_k| (+ one denormal)."""
    g = _golden()
    W, IB, ref = g["world_" + pose], g["inv_bind"], g["palette_" + pose]
    S = oracle.palette(W, IB).reshape(-1, 16)
    A = np.abs(W.astype(np.float64)).reshape(-1, 4, 4)           # [b, k, r]  column k of world
    Bm = np.abs(IB.astype(np.float64)).reshape(-1, 4, 4)         # [b, c, k]  column c of inverse bind
    bound = np.einsum("bkr,bck->bcr", A, Bm).reshape(-1, 16) * 2.0 ** -22 + 1e-30
    diff = np.abs(S.astype(np.float64) - ref.astype(np.float64))
    assert (diff <= bound).all(), "worst excess %.3e" % (diff - bound).max()
    assert (S == ref).mean() > 0.8                               # most elements are bit-identical
    assert np.array_equal(S, oracle.np_twin.palette(W, IB))      # and the twins still agree with each other


@pytest.mark.parametrize("pose", ["pose0", "tween150"])
def test_skin_pinned_to_reference_primitives_on_real_vertices(oracle, pose):
    """vs() (engine.ts:255-272) of 256 real vertices of the demo model, evaluated in the reference run with math.ts'
    Mat4.multiply / Vec3.scale / add / normalize (doubles); 173 of them are BDEF1, where the position is a single
    Mat4.multiply. Bar: 1e-6 relative (f32 rounding of a handful of operations; the GPU tolerance is 1e-4)."""
    g = _golden()
    v = g["slice_vertices"]
    S = oracle.palette(g["world_" + pose], g["inv_bind"])
    p, n = oracle.skin(np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6]), g["slice_joints"], g["slice_weights"], S)
    ref = g["skinned_" + pose]
    ep = np.linalg.norm(p - ref[:, :3], axis=1) / np.maximum(np.linalg.norm(ref[:, :3], axis=1), 1.0)
    en = np.linalg.norm(n - ref[:, 3:], axis=1)
    assert ep.max() <= 1e-6 and en.max() <= 1e-6, (ep.max(), en.max())
    assert (g["slice_weights"][:, 0] == 255).sum() >= 100        # the BDEF1 share the docstring promises
    assert np.abs(p - v[:, 0:3]).max() > 1e-3 or pose == "pose0"  # tween150 really moves the slice


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_skin_pinned_on_a_wide_sample_of_the_real_model(oracle, pose):
    """Every 28th vertex of the demo model (1 031 vertices, 166 of its 349 bones, 413 BDEF1 / 547 BDEF2 / 71 BDEF4) under
    three reference-produced poses, vs() evaluated by the reference run with math.ts primitives (stored as f32). Bar 1e-6."""
    g = _golden()
    v = g["wide_vertices"]
    S = oracle.palette(g["world_" + pose], g["inv_bind"])
    p, n = oracle.skin(np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6]), g["wide_joints"], g["wide_weights"], S)
    ref = g["skinnedwide_" + pose].astype(np.float64)
    ep = np.linalg.norm(p - ref[:, :3], axis=1) / np.maximum(np.linalg.norm(ref[:, :3], axis=1), 1.0)
    en = np.linalg.norm(n - ref[:, 3:], axis=1)
    assert ep.max() <= 1e-6 and en.max() <= 1e-6, (ep.max(), en.max())
    cnt = (g["wide_weights"] > 0).sum(axis=1)
    assert len(v) > 1000 and (cnt == 1).sum() > 400 and (cnt == 2).sum() > 500 and (cnt >= 3).sum() > 60
    assert np.abs(p - v[:, 0:3]).max() > 1.0                  # the poses really move the mesh
    # the numpy twin lands on the same bits as the C oracle here too
    p2, n2 = oracle.np_twin.skin(np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6]), g["wide_joints"], g["wide_weights"], S)
    assert np.array_equal(p, p2) and np.array_equal(n, n2)


# ---- the reference's own WGSL TEXT, interpreted (tests/golden/ref_wgsl.npz, written by tools/ref_wgsl_run.py) ----
def _wgsl():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_wgsl.npz"))


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_oracle_is_bit_identical_to_the_reference_shader_text_interpreted(oracle, pose):
    """tools/wgsl_eval.py parses the bodies of the reference's `@vertex fn vs` (engine.ts:245-276) and of its skin-matrix
    compute shader (:919-928) out of the reference checkout and evaluates them statement by statement (binary32 per
    operation, matrix products summed column by column left to right). The formula is the shader's text, not a
    re-typing of it. On the reference's 349-bone model under three reference-produced poses — palette, the 256-vertex
    slices and every 28th vertex (1 031) — the C oracle, its NumPy twin and that interpretation agree BIT FOR BIT."""
    g, w = _golden(), _wgsl()
    S = oracle.palette(g["world_" + pose], g["inv_bind"])
    assert np.array_equal(S.reshape(-1, 16), w["palette_" + pose])
    for tag in ("slice", "wide"):
        v = g[tag + "_vertices"]
        pos, nrm = np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6])
        p, n = oracle.skin(pos, nrm, g[tag + "_joints"], g[tag + "_weights"], S)
        ref = w["%s_%s" % (tag, pose)]
        assert np.array_equal(p, ref[:, :3]) and np.array_equal(n, ref[:, 3:]), "%s %s" % (tag, pose)
        p2, n2 = oracle.np_twin.skin(pos, nrm, g[tag + "_joints"], g[tag + "_weights"], S)
        assert np.array_equal(p2, ref[:, :3]) and np.array_equal(n2, ref[:, 3:])
    # row f4: the outline pass's vs() (engine.ts:431-463), interpreted the same way: expandedPos = worldPos + worldNormal *
    # material.edgeSize * 0.01 on the wide sample with edge sizes 0 / 0.4 / 1.0 / 1.5 == oracle.hull of the deformed mesh
    ref = w["wide_" + pose]
    h = oracle.hull(np.ascontiguousarray(ref[:, :3]), np.ascontiguousarray(ref[:, 3:]), w["hull_edge"])
    assert np.array_equal(h, w["hull_wide_" + pose]) and np.array_equal(h[::4], ref[::4, :3])
    # the one statement of vs() that was not evaluated is the camera projection, which is not part of the deformation
    assert [str(s) for s in w["skipped_statements"]] == ["output.position  (needs camera)"]
    assert len(str(w["vs_sha256"])) == 64 and len(str(w["cs_sha256"])) == 64


def test_wgsl_interpreter_on_hand_computed_cases():
    """The interpreter itself: a hand-written shader with known answers, f32 rounding per operation, and loud failure on
    syntax outside its subset (it must never skip shader code silently)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wgsl_eval as W
    src = """
    fn f(a: vec3f) -> vec4f {
      var acc = vec4f(0.0, 0.0, 0.0, 0.0);
      let m = mats[1u];
      for (var i = 0u; i < 3u; i++) {
        acc += (m * vec4f(a, 1.0)) * w[i];
      }
      let s = select(2.0, 0.5, acc.x > 100.0);
      out.v = acc * s;
      out.n = normalize(vec3f(3.0, 0.0, 4.0));
      if (k >= 7u) { return out; }
      out.v = vec4f(9.0, 9.0, 9.0, 9.0);
      return out;
    }"""
    ident = W.Mat(np.eye(4, dtype=np.float32))
    shift = W.Mat([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [10, 20, 30, 1]])
    env, skipped, head = W.run_function(src, r"fn\s+f\s*\(", {"a": np.array([1, 2, 3], np.float32), "mats": [ident, shift],
                                                                 "w": np.array([0.5, 0.25, 0.25], np.float32), "k": 7, "out": {}})
    o = env["__return__"]
    assert np.array_equal(o["v"], np.array([22.0, 44.0, 66.0, 2.0], np.float32)) and skipped == [] and "fn f" in head
    assert np.array_equal(o["n"], np.array([0.6, 0.0, 0.8], np.float32))
    env, _, _ = W.run_function(src, r"fn\s+f\s*\(", {"a": np.array([1, 2, 3], np.float32), "mats": [ident, shift],
                                                      "w": np.array([0.5, 0.25, 0.25], np.float32), "k": 6, "out": {}})
    assert np.array_equal(env["__return__"]["v"], np.full(4, 9.0, np.float32))
    # binary32 per operation: 1e8 + 1 - 1e8 is 0 in f32
    env, _, _ = W.run_function("fn g() { let x = 100000000.0 + 1.0 - 100000000.0; return x; }", r"fn\s+g\s*\(", {})
    assert env["__return__"] == np.float32(0.0)
    # a binding the caller did not supply: the statement is reported, not silently dropped
    env, skipped, _ = W.run_function("fn h() { let y = cam.view * 2.0; let z = 1.0; return z; }", r"fn\s+h\s*\(", {})
    assert skipped == ["y  (needs cam)"] and env["__return__"] == np.float32(1.0)
    for bad in ("fn b() { let x = sin(1.0); }", "fn b() { while (true) { } }", "fn b() { x <<= 2; }"):
        with pytest.raises((SyntaxError, AssertionError, KeyError)):
            W.run_function(bad, r"fn\s+b\s*\(", {"x": 1})


# ---- bone morphs (PMX type 2) pinned to reference execution (tests/golden/ref_bone_morph.npz, tools/ref_bone_morph_run.py) ----
@pytest.mark.parametrize("k", [0, 1, 2])
def test_bone_morph_restatement_pinned_to_reference_quaternion_and_fk_code(oracle, k):
    """The reference has no bone morphs; the semantics (q' = q * slerp(I, q_m, w), t' = t + w t_m) are this build's. On the one
    reference asset that carries a bone morph (武器.pmx: two blade bones turned about z, no translation) the whole frame was
    produced by the reference's OWN Quat.slerp / Quat.multiply / Model.rotateBones + evaluatePose / Mat4-Vec3 skin. The
    float64 restatement the GPU tests compare with (helpers.bone_morph_reference + fk_reference) must reproduce those world
    matrices (f32-store tolerance: the reference rounds every matrix product to f32, and rotateBones renormalises q'), and
    the oracle's skin of the fixture's vertices under them the reference-skinned sample."""
    import os
    from helpers import assert_parity, bone_morph_reference, fk_reference
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_bone_morph.npz"))
    w = np.zeros(1)
    w[0] = g["morph_weights"][k]
    q2, t2 = bone_morph_reference(g["base_rotations"], None, g["entry_morph"], g["entry_bone"], g["entry_translation"], g["entry_rotation"], w)
    assert np.abs(q2 - g["local_rotations"][k]).max() < 2e-7                    # the reference's own q * slerp(I, q_m, w), stored as f32
    world = fk_reference(g["parents"], g["bind"], q2, t2)
    assert np.abs(world - g["world"][k]).max() < 5e-6 * max(1.0, np.abs(g["world"][k]).max())
    if k:
        assert np.abs(g["world"][k] - g["world"][0]).max() > 0.05              # the morph really turns the blades
    v = g["vertices"]
    pr, nr = oracle.skin(v[:, 0:3], v[:, 3:6], g["joints"], g["weights"], oracle.palette(g["world"][k], g["inv_bind"]))
    assert_parity(pr, nr, g["skinned"][k][:, 0:3], g["skinned"][k][:, 3:6], "oracle vs reference-skinned weapon, weight %g" % w[0])


@pytest.mark.parametrize("pose", ["pose0", "tween150", "tween500"])
def test_envelope_of_the_evaluations_wgsl_allows(oracle, pose):
    """The reference's vs() (engine.ts:253-272) is WGSL, and WGSL leaves a driver latitude the oracle does not take: it may
    contract a * b + c into an FMA, re-associate the four-term sums of `skinMatrix * position`, and implement normalize()
    through inverseSqrt. Nothing here can execute the shader, but the spread of the LEGAL results can be bounded:
    tests/wgsl_latitude.py evaluates vs() under each model (float64 emulation, rounded to binary32 where the model rounds)
    on the wide sample of the real model (1 031 vertices, 166 bones, three reference-produced poses). With no option set it
    reproduces the oracle BIT FOR BIT (so the emulation is the oracle's arithmetic); every other legal order — and this
    build's own blended-matrix order — stays within 1e-5 of it (measured: 3.1e-7 positions, 1.9e-7 normals), i.e. the
    1e-4 tolerance of the GPU tests is two orders of magnitude wider than anything a conforming driver could produce."""
    import wgsl_latitude as wl
    g = _golden()
    v = g["wide_vertices"]
    pos, nrm = np.ascontiguousarray(v[:, 0:3]), np.ascontiguousarray(v[:, 3:6])
    S = oracle.palette(g["world_" + pose], g["inv_bind"])
    po, no = oracle.skin(pos, nrm, g["wide_joints"], g["wide_weights"], S)
    p0, n0 = wl.vs(pos, nrm, g["wide_joints"], g["wide_weights"], S)
    assert np.array_equal(p0.astype(np.float32), po) and np.array_equal(n0.astype(np.float32), no), "the emulation IS the oracle's arithmetic"
    worst = (0.0, 0.0)
    results = {}
    for name, kw in wl.MODELS.items():
        p, n = wl.vs(pos, nrm, g["wide_joints"], g["wide_weights"], S, **kw)
        ep, en = wl.distances(p, n, po.astype(np.float64), no.astype(np.float64))
        assert ep <= 1e-5 and en <= 1e-5, (name, ep, en)
        worst = (max(worst[0], ep), max(worst[1], en))
        results[name] = (p, n)
    assert worst[0] > 0 and worst[1] > 0          # the models really differ from the oracle (the test is not vacuous)
    assert worst[0] <= 1e-6 and worst[1] <= 1e-6  # what was measured, with margin: 3.1e-7 / 1.9e-7
    # ... and from each other: any two legal evaluations are within 1e-5 of one another
    names = list(results)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            ep, en = wl.distances(results[names[i]][0], results[names[i]][1], results[names[j]][0], results[names[j]][1])
            assert ep <= 1e-5 and en <= 1e-5, (names[i], names[j], ep, en)
